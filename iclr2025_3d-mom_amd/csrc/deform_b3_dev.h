// The exact three-way bf16 split of fp32 operands for v_mfma_f32_32x32x16_bf16 (see deform_field.hip for the scheme): shared by
// the fused forward (pre-split weight fragments in LDS) and the MLP backward (weights split on the fly from their fp32 copy).
#pragma once
#include "deform_mlp_dev.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t hi16(float x) { return __float_as_uint(x) & 0xFFFF0000u; }
// (a, b) -> three dwords, piece p = bf16(a_p) | bf16(b_p) << 16
__device__ __forceinline__ void split_pair(float a, float b, uint32_t (&p)[3])
{
    const uint32_t a1 = hi16(a), b1 = hi16(b);
    const float ra = a - __uint_as_float(a1), rb = b - __uint_as_float(b1);
    const uint32_t a2 = hi16(ra), b2 = hi16(rb);
    const float sa = ra - __uint_as_float(a2), sb = rb - __uint_as_float(b2);
    p[0] = (a1 >> 16) | b1;
    p[1] = (a2 >> 16) | b2;
    p[2] = (__float_as_uint(sa) >> 16) | hi16(sb);
}
// The same split by the hardware's conversion (v_cvt_pk_bf16_f32: round to nearest even, two values per instruction): x1 =
// bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2).  The two residuals are exact in fp32 (|x - x1| <= 2^-9 |x| with at most 16
// significant bits left, and so on), so x1 + x2 + x3 = x up to the rounding of the THIRD piece, 2^-26 |x| -- below the 2^-23 |x||y|
// of the product terms the six-term expansion drops anyway.  Eleven instructions per pair instead of ~13.5 (MOM_SPLIT_RNE: the
// one-kernel MLP backward, where a wave is alone on its SIMD and every vector instruction is on the critical path).
typedef __bf16 mom_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk(float a, float b)
{
    const mom_bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ void split_pair_rne(float a, float b, uint32_t (&p)[3])
{
    p[0] = cvt_pk(a, b);
    const float ra = a - __uint_as_float(p[0] << 16), rb = b - __uint_as_float(p[0] & 0xFFFF0000u);
    p[1] = cvt_pk(ra, rb);
    const float sa = ra - __uint_as_float(p[1] << 16), sb = rb - __uint_as_float(p[1] & 0xFFFF0000u);
    p[2] = cvt_pk(sa, sb);
}
struct Frag3 {
    uint4 p[3];            // the three pieces of eight values: an MFMA operand each
};
__device__ __forceinline__ Frag3 split8(const float (&v)[8])
{
    Frag3 f;
#ifdef B3F_NO_SPLIT        // timing ablation: the operands' raw bits as every piece (wrong numbers, the split's instructions gone)
#pragma unroll
    for (int p = 0; p < 3; p++) f.p[p] = make_uint4(__float_as_uint(v[0]) ^ __float_as_uint(v[4]), __float_as_uint(v[1]) ^ __float_as_uint(v[5]), __float_as_uint(v[2]) ^ __float_as_uint(v[6]), __float_as_uint(v[3]) ^ __float_as_uint(v[7]));
    return f;
#endif
    uint32_t q[4][3];
#pragma unroll
    for (int j = 0; j < 4; j++) {
#ifdef MOM_SPLIT_RNE
        split_pair_rne(v[2 * j], v[2 * j + 1], q[j]);
#else
        split_pair(v[2 * j], v[2 * j + 1], q[j]);
#endif
    }
#pragma unroll
    for (int p = 0; p < 3; p++) f.p[p] = make_uint4(q[0][p], q[1][p], q[2][p], q[3][p]);
    return f;
}
// the B operand of the next layer: accumulator tile -> four K-steps of three pieces (RELU: through the ReLU first)
template <bool RELU>
__device__ __forceinline__ void split_tile(const f32x16 (&t)[2], Frag3 (&B)[4])
{
#pragma unroll
    for (int s = 0; s < 4; s++) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float x = t[s >> 1][8 * (s & 1) + j];
            v[j] = RELU ? fmaxf(x, 0.f) : x;
        }
        B[s] = split8(v);
    }
}
__device__ __forceinline__ f32x16 mfma16(uint4 a, uint4 b, f32x16 c)
{
#ifdef B3F_NO_MFMA         // timing ablation: the operands are kept alive, no matrix instruction is issued
    asm volatile("" :: "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w));
    c[0] += __uint_as_float(a.x);
    return c;
#endif
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// acc[mt] += W x (TRANS: W^T x) with the weights read from their fp32 copy in LDS (Wl[in * kWStride + out]) and split on the
// fly: the A operand of K-step s is the eight weights whose contraction index is 16 s + 4 h + (j & 3) + 8 (j >> 2), j = 0..7 --
// the order in which an accumulator tile holds its features, so that B is a predecessor's accumulator, split in place.
template <bool TRANS>
__device__ __forceinline__ void layer_b3_otf(const float* __restrict__ Wl, const Frag3 (&B)[4], f32x16 (&acc)[2], int col, int h)
{
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int s = 0; s < 4; s++) {
            float v[8];
            const int m = 32 * mt + col;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int kk = 16 * s + 4 * h + (j & 3) + 8 * (j >> 2);
                v[j] = TRANS ? Wl[m * kWStride + kk] : Wl[kk * kWStride + m];
            }
            const Frag3 A = split8(v);
            f32x16 c = acc[mt];
            c = mfma16(A.p[2], B[s].p[0], c);
            c = mfma16(A.p[0], B[s].p[2], c);
            c = mfma16(A.p[1], B[s].p[1], c);
            c = mfma16(A.p[1], B[s].p[0], c);
            c = mfma16(A.p[0], B[s].p[1], c);
            c = mfma16(A.p[0], B[s].p[0], c);
            acc[mt] = c;
            __builtin_amdgcn_sched_barrier(0);
        }
}

}  // namespace
