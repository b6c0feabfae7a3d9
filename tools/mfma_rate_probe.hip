// Cycles per v_mfma_f32_32x32x16_bf16 in bare loops (one wave per SIMD, every CU): one accumulator chain, four accumulators
// in turn, and a second product whose B operand is written by vector instructions just before.  hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0)

template <int MODE>
__global__ __launch_bounds__(256) void probe(const u32x4* __restrict__ in, float* __restrict__ out, long long* __restrict__ cyc, int iters) {
    const int lane = threadIdx.x;
    u32x4 a[12], b[12];
    for (int i = 0; i < 12; ++i) { a[i] = in[(i * 256 + lane) % 4096]; b[i] = in[((i + 12) * 256 + lane) % 4096]; }
    f32x16 c0, c1, c2, c3;
    for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; c2[i] = 0.f; c3[i] = 0.f; }
    __builtin_amdgcn_sched_barrier(0);
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {                  // 48 on one accumulator
#pragma unroll
            for (int k = 0; k < 48; ++k) c0 = MF(a[k % 12], b[(k * 5) % 12], c0);
        } else if (MODE == 1) {           // 48 over four accumulators in turn
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                c0 = MF(a[k], b[(k * 5) % 12], c0);
                c1 = MF(a[(k + 1) % 12], b[(k * 5) % 12], c1);
                c2 = MF(a[k], b[(k * 7) % 12], c2);
                c3 = MF(a[(k + 1) % 12], b[(k * 7) % 12], c3);
            }
        } else if (MODE == 3) {           // accumulators in AGPRs, operands in VGPRs
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c0) : "v"(a[k]), "v"(b[(k * 5) % 12]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c1) : "v"(a[(k + 1) % 12]), "v"(b[(k * 5) % 12]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c2) : "v"(a[k]), "v"(b[(k * 7) % 12]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c3) : "v"(a[(k + 1) % 12]), "v"(b[(k * 7) % 12]));
            }
        } else if (MODE == 4) {           // accumulators and the A operand in AGPRs
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c0) : "a"(a[k]), "v"(b[(k * 5) % 12]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c1) : "a"(a[(k + 1) % 12]), "v"(b[(k * 5) % 12]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c2) : "a"(a[k]), "v"(b[(k * 7) % 12]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c3) : "a"(a[(k + 1) % 12]), "v"(b[(k * 7) % 12]));
            }
        } else {                          // chains of 6 on four accumulators
#pragma unroll
            for (int k = 0; k < 2; ++k) {
#pragma unroll
                for (int j = 0; j < 6; ++j) c0 = MF(a[j + 6 * k], b[j], c0);
#pragma unroll
                for (int j = 0; j < 6; ++j) c1 = MF(a[j + 6 * k], b[j + 6], c1);
#pragma unroll
                for (int j = 0; j < 6; ++j) c2 = MF(a[j], b[j + 6 * k], c2);
#pragma unroll
                for (int j = 0; j < 6; ++j) c3 = MF(a[j + 6], b[j + 6 * k], c3);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * 256 + lane] = s;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

// The fp32 matrix instruction (v_mfma_f32_32x32x2_f32, 64 cycles) in the same bare loop: four accumulators in turn, 64 per
// iteration, long enough (milliseconds) for the clock to settle under the load.
__global__ __launch_bounds__(256) void probe_f32(const float* __restrict__ in, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x;
    float a[16], b[16];
    for (int i = 0; i < 16; ++i) { a[i] = in[(i * 256 + lane) % 16384]; b[i] = in[((i + 16) * 256 + lane) % 16384]; }
    f32x16 c0, c1, c2, c3;
    for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; c2[i] = 0.f; c3[i] = 0.f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[(k * 5) % 16], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(k + 1) % 16], b[(k * 5) % 16], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[(k * 7) % 16], c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(k + 1) % 16], b[(k * 7) % 16], c3, 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * 256 + lane] = s;
}

int main() {
    {
        const int grid = 256, iters = 4000;
        std::vector<float> hf(16384);
        float* fin; float* fout;
        hipMalloc(&fin, hf.size() * 4); hipMalloc(&fout, grid * 256 * 4);
        for (int data = 0; data < 2; ++data) {
            for (auto& v : hf) v = data ? (float)(rand() % 2001 - 1000) * 1e-3f : 0.f;
            hipMemcpy(fin, hf.data(), hf.size() * 4, hipMemcpyHostToDevice);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(probe_f32, dim3(grid), dim3(256), 0, 0, fin, fout, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                printf("fp32 32x32x2 data %s rep %d: kernel %.2f ms, %.1f TFLOP/s (64 cycles each at %.2f GHz)\n", data ? "random" : "zeros ", rep, ms,
                       (double)grid * 4 * iters * 64 * 4096.0 / (ms * 1e-3) / 1e12, (double)iters * 64 * 64 / (ms * 1e-3) / 1e9);
            }
        }
        hipFree(fin); hipFree(fout);
    }
    const int iters = 200, grid = 256;
    std::vector<unsigned> h(4096 * 4);
    u32x4* in; float* out; long long* cyc;
    hipMalloc(&in, h.size() * 4); hipMalloc(&out, grid * 256 * 4); hipMalloc(&cyc, grid * 8);
    for (int data = 0; data < 2; ++data) {
        for (auto& v : h) { const unsigned short lo = data ? (unsigned short)(0x3c00 + rand() % 0x600) : 0, hi = data ? (unsigned short)(0xbc00 + rand() % 0x600) : 0; v = lo | ((unsigned)hi << 16); }
        hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 5; ++mode) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(256), 0, 0, in, out, cyc, iters);
                else if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(256), 0, 0, in, out, cyc, iters);
                else if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3(grid), dim3(256), 0, 0, in, out, cyc, iters);
                else if (mode == 4) hipLaunchKernelGGL(probe<4>, dim3(grid), dim3(256), 0, 0, in, out, cyc, iters);
                else hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(256), 0, 0, in, out, cyc, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<long long> c(grid); hipMemcpy(c.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
            double avg = 0; for (auto v : c) avg += v; avg /= grid;
            printf("data %s mode %d: %.1f cycles per MFMA, kernel %.1f us, %.0f TFLOP/s\n", data ? "random" : "zeros ", mode, avg / (iters * 48.0), ms * 1e3,
                   (double)grid * 4 * iters * 48 * 32768.0 / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
