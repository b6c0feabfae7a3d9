// Does a consumer kernel that is allowed to start before its producer has finished (no barrier between the two AQL
// packets: hipExtLaunchKernelGGL with hipExtAnyOrderLaunch) hide its launch and its weight fetch under the producer?
//
//   producer (64 workgroups, like the norm kernel of a decode step): out[row][0..1024) = in[row][..] + tag; then a device-scope
//     release fence and one atomic increment of the pair's counter per workgroup
//   consumer (256 workgroups, like a projection): requests its 64 KiB of "weights" first, then one lane polls the counter
//     (a BOUNDED number of polls: on exhaustion it raises an error flag and goes on, so the probe cannot hang), acquire
//     fence, reads a slice of `out` (every row), writes sum(weights) + sum(slice) per workgroup
//
// A chain of PAIRS such pairs is timed with the consumer launched (a) in order, (b) any-order; eagerly and from a captured
// graph.  Every consumer result is checked on the host: a consumer that read `out` before the producer wrote it shows up as
// a wrong sum (the tag differs per repetition).
//
//   hipcc -O3 --offload-arch=gfx950 tools/anyorder_probe.hip -o tools/_build/anyorder_probe && tools/_build/anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

constexpr int ROWS = 64, D = 1024, CONSUMERS = 256, PAIRS = 30, POLL_BUDGET = 1 << 16;

// stamps[pair][0..4]: producer start, producer end, consumer start, consumer after the wait, consumer end (100 MHz wall clock,
// workgroup 0 of each)
__global__ __launch_bounds__(128) void producer(const float* __restrict__ in, float* __restrict__ out, float tag, int* counter,
                                                long long* stamps) {
    const int row = blockIdx.x, c = threadIdx.x * 8;
    if (blockIdx.x == 0 && threadIdx.x == 0) stamps[0] = wall_clock64();
#pragma unroll
    for (int i = 0; i < 8; ++i) out[row * D + c + i] = in[row * D + c + i] + tag;
    __threadfence();                       // the row is visible device-wide before the workgroup reports
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (blockIdx.x == 0 && threadIdx.x == 0) stamps[1] = wall_clock64();
}

__global__ __launch_bounds__(256) void consumer(const f4* __restrict__ w, const float* out, float* __restrict__ res,
                                                const int* counter, int expect, int* err, long long* stamps) {
    __shared__ float red[4];
    __shared__ int ok;
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && tid == 0) stamps[2] = wall_clock64();
    // 64 KiB of weights per workgroup, requested before anything is waited for
    f4 wv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) wv[i] = __builtin_nontemporal_load(w + ((size_t)blockIdx.x * 16 + i) * 256 + tid);
    if (tid == 0) {
        int polls = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < expect && ++polls < POLL_BUDGET)
            __builtin_amdgcn_s_sleep(2);
        ok = polls < POLL_BUDGET;
        if (polls >= POLL_BUDGET) atomicAdd(err, 1);
        if (blockIdx.x == 0) stamps[3] = wall_clock64();
    }
    __syncthreads();
    __atomic_thread_fence(__ATOMIC_ACQUIRE);   // what the producers released is what the loads below see
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += wv[i].x + wv[i].y + wv[i].z + wv[i].w;
    {   // one 16-byte piece of every row per workgroup (a projection's activations are as small a read as this)
        const f4 v = *reinterpret_cast<const f4*>(out + (size_t)(tid >> 2) * D + ((blockIdx.x * 4 + (tid & 3)) * 4) % D);
        s += v.x + v.y + v.z + v.w;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) res[blockIdx.x] = red[0] + red[1] + red[2] + red[3] + (ok ? 0.f : 1e30f);
    if (blockIdx.x == 0 && tid == 0) stamps[4] = wall_clock64();
}

int main() {
    float *in, *out, *res; f4* w; int *counters, *err;
    const size_t wn = (size_t)CONSUMERS * 16 * 256;
    CK(hipMalloc(&in, sizeof(float) * ROWS * D)); CK(hipMalloc(&out, sizeof(float) * ROWS * D * PAIRS));
    CK(hipMalloc(&res, sizeof(float) * CONSUMERS * PAIRS)); CK(hipMalloc(&w, sizeof(f4) * wn * PAIRS));
    CK(hipMalloc(&counters, sizeof(int) * PAIRS)); CK(hipMalloc(&err, sizeof(int)));
    long long* stamps; CK(hipMalloc(&stamps, sizeof(long long) * 8 * PAIRS));
    CK(hipMemset(in, 0, sizeof(float) * ROWS * D)); CK(hipMemset(w, 0, sizeof(f4) * wn * PAIRS)); CK(hipMemset(err, 0, sizeof(int)));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> h(CONSUMERS * PAIRS);
    int rep_tag = 0;

    auto enqueue = [&](bool any_order, float tag) {
        CK(hipMemsetAsync(counters, 0, sizeof(int) * PAIRS, st));
        for (int p = 0; p < PAIRS; ++p) {
            hipLaunchKernelGGL(producer, dim3(ROWS), dim3(128), 0, st, in, out + (size_t)p * ROWS * D, tag + p, counters + p, stamps + 8 * p);
            hipExtLaunchKernelGGL(consumer, dim3(CONSUMERS), dim3(256), 0, st, nullptr, nullptr, any_order ? hipExtAnyOrderLaunch : 0,
                                  w + wn * p, out + (size_t)p * ROWS * D, res + p * CONSUMERS, counters + p, ROWS, err, stamps + 8 * p);
        }
    };
    auto check = [&](float tag, const char* what) {
        CK(hipMemcpy(h.data(), res, sizeof(float) * h.size(), hipMemcpyDeviceToHost));
        int bad = 0;
        for (int p = 0; p < PAIRS; ++p)
            for (int b = 0; b < CONSUMERS; ++b) bad += h[p * CONSUMERS + b] != (tag + p) * 1024;
        int herr = 0; CK(hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost));
        if (bad || herr) printf("  %s: %d wrong sums, %d poll budgets exhausted\n", what, bad, herr);
        return bad + herr;
    };
    for (int graph = 0; graph < 2; ++graph)
        for (int any = 0; any < 2; ++any) {
            hipGraphExec_t ge = nullptr;
            const int reps = 40;
            float ms = 0.f; int failures = 0;
            // tags are small integers so that every sum is exact in fp32
            if (graph) {
                hipGraph_t g;
                CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                enqueue(any, 1.0f);
                CK(hipStreamEndCapture(st, &g));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                CK(hipGraphDestroy(g));
                for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, st));
                CK(hipStreamSynchronize(st));
                CK(hipEventRecord(e0, st));
                for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                failures += check(1.0f, "graph");
                CK(hipGraphExecDestroy(ge));
            } else {
                for (int r = 0; r < 3; ++r) enqueue(any, 1.0f);
                CK(hipStreamSynchronize(st));
                CK(hipEventRecord(e0, st));
                for (int r = 0; r < reps; ++r) enqueue(any, (float)(2 + ((rep_tag++) & 3)));
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                failures += check((float)(2 + ((rep_tag - 1) & 3)), "eager");
            }
            CK(hipEventElapsedTime(&ms, e0, e1));
            long long hs[8 * PAIRS];
            CK(hipMemcpy(hs, stamps, sizeof(hs), hipMemcpyDeviceToHost));
            const long long* t = hs + 8 * 15;              // pair 15 of the last repetition, times in us from the producer's start
            printf("   pair 15: producer 0 .. %.2f, consumer start %.2f, wait over %.2f, end %.2f; next producer starts %.2f\n",
                   (t[1] - t[0]) / 100.0, (t[2] - t[0]) / 100.0, (t[3] - t[0]) / 100.0, (t[4] - t[0]) / 100.0, (t[8] - t[0]) / 100.0);
            printf("%s, consumer %s: %.2f us per producer+consumer pair (%d pairs x %d reps)%s\n", graph ? "graph" : "eager",
                   any ? "any-order" : "in order ", 1e3 * ms / (reps * PAIRS), PAIRS, reps, failures ? "  ** FAILED **" : "");
            fflush(stdout);
        }
    return 0;
}
