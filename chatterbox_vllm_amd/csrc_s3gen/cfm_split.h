// Flow-matching decoder, split-bf16 forms of the two matrix kernels (cfm_split.hip).
//
// An fp32 value x is held as three bf16 values x0 + x1 + x2 = x (x0 = bf16(x), x1 = bf16(x - x0), x2 = x - x0 - x1: 3 x 8
// significant bits hold the 24 of an fp32 exactly), and a product of two such operands is taken as the six partial products
// x0 y0 + x0 y1 + x1 y0 + x1 y1 + x0 y2 + x2 y0 on the bf16 matrix instruction (each exact in its fp32 accumulator input);
// the three left out are below 2^-24 of |x y|.  Six v_mfma_f32_32x32x16_bf16 cover the reduction depth of eight
// v_mfma_f32_32x32x2f32 in 192 instead of 512 cycles, and the result is as close to the fp64 one as the fp32 instruction's.
#pragma once
#include "hift_common.h"

constexpr int kSplitMaxUtt = 8;

typedef __bf16 s3_bf16x2 __attribute__((ext_vector_type(2)));
// (a, b) -> three dwords, each bf16 of a in the low half and of b in the high half: value, first and second remainder
__device__ __forceinline__ void split_pair(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
    const s3_bf16x2 h0 = {(__bf16)a, (__bf16)b};
    p0 = __builtin_bit_cast(unsigned, h0);
    a -= __uint_as_float(p0 << 16);
    b -= __uint_as_float(p0 & 0xffff0000u);
    const s3_bf16x2 h1 = {(__bf16)a, (__bf16)b};
    p1 = __builtin_bit_cast(unsigned, h1);
    a -= __uint_as_float(p1 << 16);
    b -= __uint_as_float(p1 & 0xffff0000u);
    const s3_bf16x2 h2 = {(__bf16)a, (__bf16)b};
    p2 = __builtin_bit_cast(unsigned, h2);
}


// Attention operands in the order the matrix instruction takes them, one 1 KiB block (64 lanes x 8 bf16) per load
// instruction; element offsets inside the tensor of one (row, head, plane), which holds ld / 32 frame tiles of 2048 elements:
//   Q and K: ((tile * 4 + s) * 64 + lane) * 8 + j   =  X[d = 16 s + 8 (lane / 32) + j][frame = 32 tile + lane % 32]
//                                                      (Q already scaled by log2(e) / sqrt(64))
//   V:       ((tile * 4 + 2 t + db) * 64 + lane) * 8 + j  =  V[frame = 32 tile + 16 t + 8 (j / 4) + 4 (lane / 32) + j % 4]
//                                                             [d = 32 db + lane % 32]
//                                                      (the frame order of a lane's eight values is the order in which a
//                                                       lane holds the scores of a 32 x 32 accumulator)
// Tensor of (row b, head h, plane p) at ((b * 8 + h) * 3 + p) * (ld / 32) * 2048 elements.
struct SplitAttnArgs {
    const unsigned short* qp; const unsigned short* kp; const unsigned short* vp;
    float* o; size_t o_bs;                  // [row][512][ld] fp32, channel-major
    int ld, L, rows;
    int Lk[kSplitMaxUtt];                   // keys of rows 2 u and 2 u + 1
};
int launch_attn_split(const SplitAttnArgs& a, hipStream_t st);

// Element offsets of one value inside the tensor of its (row, head, plane), for the kernel that writes q, k, v straight into
// operand order (lin_kernel's epilogue): frame n, channel d of the head
__device__ __forceinline__ size_t split_qk_offset(int n, int d) {
    return ((size_t)((n >> 5) * 4 + (d >> 4)) * 64 + ((d >> 3) & 1) * 32 + (n & 31)) * 8 + (d & 7);
}
__device__ __forceinline__ size_t split_v_offset(int n, int d) {
    const int kk = n & 31;
    return ((size_t)((n >> 5) * 4 + 2 * (kk >> 4) + (d >> 5)) * 64 + ((kk >> 2) & 1) * 32 + (d & 31)) * 8 + 4 * ((kk >> 3) & 1) + (kk & 3);
}

// fp32 q | k (channel-major [row][1024][ld]) and v (frame-major [row][ld][vt_ld]) -> the three operand tensors above
int launch_split_qkv(const float* qk, size_t qk_bs, const float* vt, size_t vt_bs, int vt_ld, int ld, int rows, unsigned short* qp,
                     unsigned short* kp, unsigned short* vp, hipStream_t st);
