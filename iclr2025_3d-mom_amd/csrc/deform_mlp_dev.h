// Device helpers of the fused deformation MLP (v_mfma_f32_32x32x2_f32 layers with the weights resident in LDS), shared by
// deform_mlp.hip (MLP on stored features) and deform_field.hip (HexPlane gather fused in front of / behind the MLP).
#pragma once
#include "mom_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kHid = 64;
constexpr int kWStride = 65;                         // LDS row stride of a weight matrix ([in][out])
constexpr int kWFloats = kHid * kWStride;            // one layer
constexpr int kStageStride = 33;                     // staging rows: [feature][32 gaussians + 1 pad]
constexpr int kStageFloats = kHid * kStageStride + 4 * 32;       // sA | dout[32][4]
constexpr int kDxWaves = 8;                          // waves per workgroup of the dx kernel (they share one copy of the weights)

// LDS map (floats)
constexpr int kLW = 0;                               // [4][64][65]
constexpr int kLB = kLW + 4 * kWFloats;              // [4][64]
constexpr int kLW2 = kLB + 4 * kHid;                 // [3][4][64] (rows >= nout are zero)
constexpr int kLB2 = kLW2 + 3 * 4 * kHid;            // [3][4]
constexpr int kLFwdTotal = kLB2 + 16;
constexpr int kLStage = kLFwdTotal;                  // [kDxWaves][kStageFloats]   (backward only)
constexpr int kLBwdTotal = kLStage + kDxWaves * kStageFloats;

__device__ __forceinline__ int fmap(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

struct MlpDev {
    const float *W0, *b0, *W1[3], *b1[3], *W2[3], *b2[3];
    float *dW0, *db0, *dW1[3], *db1[3], *dW2[3], *db2[3];
};

// Cooperative load of all weights into LDS (any workgroup size that is a multiple of 64, at least 256).  The four 64x64
// matrices are fetched as float4 and ALL of a thread's fetches are issued before its first LDS store: written as a plain
// copy loop the prologue paid one memory latency per iteration (16 k cycles of the forward kernel's 150 k).
__device__ __forceinline__ void load_weights(const MlpDev& m, float* __restrict__ lds)
{
    const int nth = (int)blockDim.x;
    const float* Ws[4] = {m.W0, m.W1[0], m.W1[1], m.W1[2]};
    const float* bs[4] = {m.b0, m.b1[0], m.b1[1], m.b1[2]};
    constexpr int kQuads = 4 * kHid * kHid / 4, kMaxPer = kQuads / 256;      // 4096 float4 in all; <= 16 per thread
    float4 v[kMaxPer];
#pragma unroll
    for (int j = 0; j < kMaxPer; j++) {
        const int q = threadIdx.x + j * nth;                       // quad q: layer q >> 10, floats 4 (q & 1023) .. + 3 of it
        if (q < kQuads) v[j] = reinterpret_cast<const float4*>(Ws[q >> 10])[q & 1023];
    }
    float bias = 0.f;                                              // workgroups have at least 4 * kHid = 256 threads
    if (threadIdx.x < 4 * kHid) bias = bs[threadIdx.x >> 6][threadIdx.x & 63];
#pragma unroll
    for (int j = 0; j < kMaxPer; j++) {
        const int q = threadIdx.x + j * nth;
        if (q < kQuads) {
            const int L = q >> 10, i = 4 * (q & 1023), o = i >> 6, k = i & 63;       // W[out][in] row-major -> lds[in][out], stride 65
            float* d = lds + kLW + L * kWFloats + k * kWStride + o;
            d[0] = v[j].x; d[kWStride] = v[j].y; d[2 * kWStride] = v[j].z; d[3 * kWStride] = v[j].w;
        }
    }
    if (threadIdx.x < 4 * kHid) lds[kLB + threadIdx.x] = bias;
    for (int i = threadIdx.x; i < 3 * 4 * kHid; i += nth) {
        const int head = i >> 8, n = (i >> 6) & 3, f = i & 63;
        const int nout = head == 2 ? 4 : 3;
        lds[kLW2 + i] = n < nout ? m.W2[head][n * kHid + f] : 0.f;
    }
    if (threadIdx.x < 12) {
        const int head = threadIdx.x >> 2, n = threadIdx.x & 3;
        const int nout = head == 2 ? 4 : 3;
        lds[kLB2 + threadIdx.x] = n < nout ? m.b2[head][n] : 0.f;
    }
}

// out[mt] += sum_k A(mt,k) in[k].  TRANS=false: A = W (out = W in).  TRANS=true: A = W^T (din = W^T dout).
template <bool TRANS>
__device__ __forceinline__ void layer64(const float* __restrict__ Wl, const f32x16 (&in)[2], f32x16 (&out)[2], int col, int h)
{
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int kt = 0; kt < 2; kt++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int k = 32 * kt + fmap(r, h), mrow = 32 * mt + col;
                const float a = TRANS ? Wl[mrow * kWStride + k] : Wl[k * kWStride + mrow];
                out[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, in[kt][r], out[mt], 0, 0, 0);
                if (r == 15) __builtin_amdgcn_sched_barrier(0);   // bound the scheduler's look-ahead (register budget)
            }
}

// The same layer with the A fragments (weights, from LDS) of the NEXT group of 16 MFMAs requested before the current group is
// issued.  layer64 leaves the order to the compiler, which reads two weights, waits for them and issues two MFMAs: a wave that
// is alone on its SIMD then pays an LDS round trip per pair of MFMAs (29.7 k cycles per tile of 256 MFMAs, where the matrix pipe
// needs 16.4 k: in-kernel stamps of deform_field_fwd_kernel).  With other waves of the same program on the SIMD that wait is
// hidden by their MFMAs, which is why the multi-wave kernels of deform_mlp.hip do not need this form.
template <bool TRANS, int G = 16>
__device__ __forceinline__ void layer64p(const float* __restrict__ Wl, const f32x16 (&in)[2], f32x16 (&out)[2], int col, int h)
{
    // groups of G = 16 or 8 MFMAs, two fragment buffers of G registers
    float a[2][G];
    constexpr int kGroups = 64 / G, kPer = 16 / G;    // groups in all; groups per 16 k-steps of one (mt, kt) block
    auto fetch = [&](int grp, float (&dst)[G]) {
        const int blk = grp / kPer, mt = blk >> 1, kt = blk & 1, r0 = G * (grp % kPer);
#pragma unroll
        for (int r = 0; r < G; r++) {
            const int k = 32 * kt + fmap(r0 + r, h), mrow = 32 * mt + col;
            dst[r] = TRANS ? Wl[mrow * kWStride + k] : Wl[k * kWStride + mrow];
        }
    };
    fetch(0, a[0]);
#pragma unroll
    for (int grp = 0; grp < kGroups; grp++) {
        const int blk = grp / kPer, mt = blk >> 1, kt = blk & 1, r0 = G * (grp % kPer);
        if (grp + 1 < kGroups) fetch(grp + 1, a[(grp + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);            // the next group's reads stay above this group's MFMAs
#pragma unroll
        for (int r = 0; r < G; r++) out[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[grp & 1][r], in[kt][r0 + r], out[mt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// layer64p with the PREVIOUS head's thin output layer (p[n] += sum_f W2[n][f] hp[f], 32 pieces of one 16-byte LDS read + 4 FMAs)
// spread between the MFMAs: a wave that is alone on its SIMD has nothing else to put into the 64 cycles an MFMA occupies the
// pipe, and run after the layer the output layer costs 1.6 k cycles per head (stamps).  The caller adds the cross-half exchange
// and the bias (out_finish).
template <bool TRANS>
__device__ __forceinline__ void layer64p_fill(const float* __restrict__ Wl, const f32x16 (&in)[2], f32x16 (&out)[2], int col, int h,
                                              const float* __restrict__ W2l, const f32x16 (&hp)[2], float (&p)[4])
{
    // A fragments in groups of EIGHT here (two buffers of 8 registers): the two hidden tiles of the pipelined heads leave no room
    // for two buffers of 16 under the 168 registers of a 768-thread workgroup
    float a[2][8];
    auto fetch = [&](int grp, float (&dst)[8]) {      // grp 0..7: (mt, kt, half of the 16 k-steps)
        const int mt = grp >> 2, kt = (grp >> 1) & 1, r0 = 8 * (grp & 1);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int k = 32 * kt + fmap(r0 + r, h), mrow = 32 * mt + col;
            dst[r] = TRANS ? Wl[mrow * kWStride + k] : Wl[k * kWStride + mrow];
        }
    };
    auto piece_w = [&](int s) {      // piece s: output s >> 3, features 32 ((s >> 2) & 1) + 8 (s & 3) + 4 h .. + 3
        return *reinterpret_cast<const float4*>(W2l + (s >> 3) * kHid + 32 * ((s >> 2) & 1) + 8 * (s & 3) + 4 * h);
    };
    fetch(0, a[0]);
    float4 w[2];
    w[0] = piece_w(0);
#pragma unroll
    for (int grp = 0; grp < 8; grp++) {
        const int mt = grp >> 2, kt = (grp >> 1) & 1, r0 = 8 * (grp & 1);
        if (grp + 1 < 8) fetch(grp + 1, a[(grp + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            out[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[grp & 1][r], in[kt][r0 + r], out[mt], 0, 0, 0);
            if ((r & 1) == 0) {
                const int s = grp * 4 + (r >> 1), pm = (s >> 2) & 1, pq = s & 3;
                if (s + 1 < 32) w[(s + 1) & 1] = piece_w(s + 1);
                const float4 ww = w[s & 1];
                p[s >> 3] += ww.x * hp[pm][4 * pq] + ww.y * hp[pm][4 * pq + 1] + ww.z * hp[pm][4 * pq + 2] + ww.w * hp[pm][4 * pq + 3];
            }
        }
        // order inside the group: MFMA, one LDS read, four vector instructions, MFMA, MFMA, ...
#pragma unroll
        for (int r = 0; r < 8; r++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if ((r & 1) == 0) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}
__device__ __forceinline__ void out_finish(const float* __restrict__ b2l, const float (&p)[4], float (&o)[4])
{
#pragma unroll
    for (int n = 0; n < 4; n++) o[n] = p[n] + __shfl_xor(p[n], 32) + b2l[n];
}

__device__ __forceinline__ void init_bias(const float* __restrict__ b, f32x16 (&t)[2], int h)
{
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int r = 0; r < 16; r++) t[mt][r] = b[32 * mt + fmap(r, h)];
}
__device__ __forceinline__ void zero_tile(f32x16 (&t)[2])
{
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int r = 0; r < 16; r++) t[mt][r] = 0.f;
}
__device__ __forceinline__ void relu_tile(f32x16 (&t)[2])
{
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int r = 0; r < 16; r++) t[mt][r] = fmaxf(t[mt][r], 0.f);
}

// feat [P][64] row-major <-> T layout (lane = gaussian column, registers = features)
__device__ __forceinline__ void load_feat(const float* __restrict__ feat, int g, bool ok, int h, f32x16 (&t)[2])
{
    const float* row = feat + (size_t)(ok ? g : 0) * kHid;
#pragma unroll
    for (int kt = 0; kt < 2; kt++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            float4 v = *reinterpret_cast<const float4*>(row + 32 * kt + 8 * q + 4 * h);
            if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
            t[kt][4 * q + 0] = v.x;
            t[kt][4 * q + 1] = v.y;
            t[kt][4 * q + 2] = v.z;
            t[kt][4 * q + 3] = v.w;
        }
}
__device__ __forceinline__ void store_feat(float* __restrict__ feat, int g, bool ok, int h, const f32x16 (&t)[2])
{
    if (!ok) return;
    float* row = feat + (size_t)g * kHid;
#pragma unroll
    for (int kt = 0; kt < 2; kt++)
#pragma unroll
        for (int q = 0; q < 4; q++)
            *reinterpret_cast<float4*>(row + 32 * kt + 8 * q + 4 * h) =
                make_float4(t[kt][4 * q + 0], t[kt][4 * q + 1], t[kt][4 * q + 2], t[kt][4 * q + 3]);
}

// thin output layer on the VALU: o[n] = b2[n] + sum_f W2[n][f] a1[f]; the two lane halves hold complementary feature
// subsets of the same gaussian, so they exchange partial sums
__device__ __forceinline__ void out_layer(const float* __restrict__ W2l, const float* __restrict__ b2l, const f32x16 (&a1)[2], int h,
                                          float (&o)[4])
{
#pragma unroll
    for (int n = 0; n < 4; n++) {
        float p = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float4 w = *reinterpret_cast<const float4*>(W2l + n * kHid + 32 * mt + 8 * q + 4 * h);
                p += w.x * a1[mt][4 * q] + w.y * a1[mt][4 * q + 1] + w.z * a1[mt][4 * q + 2] + w.w * a1[mt][4 * q + 3];
            }
        o[n] = p + __shfl_xor(p, 32) + b2l[n];
    }
}

// Optional activated copies of the forward's outputs (mom_deform_forward_activated): exp(scales), rots / |rots|,
// sigmoid(opacity_raw) -- what mom_activations_forward computes from the stored outputs, without its launch.
struct ActOut {
    float *scales, *rots, *opacity;
    const float* opacity_raw;
};

// stage a T-layout tile as [feature][gaussian] rows
__device__ __forceinline__ void stage_tile(float* __restrict__ s, const f32x16 (&t)[2], int col, int h)
{
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int r = 0; r < 16; r++) s[(32 * mt + fmap(r, h)) * kStageStride + col] = t[mt][r];
}

}  // namespace

static int fill_dev(const MomDeformMLP* w, MlpDev* d)
{
    if (!w || !w->W0 || !w->b0) return MOM_EINVAL;
    d->W0 = w->W0; d->b0 = w->b0;
    d->dW0 = w->dW0; d->db0 = w->db0;
    for (int i = 0; i < 3; i++) {
        if (!w->W1[i] || !w->b1[i] || !w->W2[i] || !w->b2[i]) return MOM_EINVAL;
        d->W1[i] = w->W1[i]; d->b1[i] = w->b1[i]; d->W2[i] = w->W2[i]; d->b2[i] = w->b2[i];
        d->dW1[i] = w->dW1[i]; d->db1[i] = w->db1[i]; d->dW2[i] = w->dW2[i]; d->db2[i] = w->db2[i];
    }
    return MOM_OK;
}

