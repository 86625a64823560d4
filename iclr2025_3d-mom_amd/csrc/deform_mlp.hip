// Fused deformation MLP on the f32 matrix cores, forward and backward, gfx950.
//
// Replaces Deformation.forward_dynamic's 7 nn.Linear + ReLU + residual adds (reference
// scene/deformation.py:53-65,97-135 with the shipped config: W=64, D=0, heads pos/scales/rotations live,
// opacity/shs heads dead) and their autograd backward (14 GEMMs + elementwise):
//     h0 = W0 f + b0                        (feature_out, no activation after it)
//     for head in {pos, scales, rot}:  o = W2 relu(W1 relu(h0) + b1) + b2
//     pts = xyz + o_pos + c * scene_flow ; scales = s + o_scales ; rot = r + o_rot
//
// All 64x64 layers run on v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma chain) in the TRANSPOSED form
// H^T = W X^T: weights are the A operand, activations the B operand with the Gaussian on the lane.  A 32x32
// accumulator tile then has exactly the register layout the next layer needs as ITS B operand (register r of
// lane half h is feature (r&3)+8(r>>2)+4h), so activations never leave registers between layers; the weight
// fragments are pre-swizzled once per step into that K order (mlp_prep_kernel) and stream from L1/L2 as one
// coalesced 256-byte load per MFMA pair.  The 64->{3,3,4} output layers are too thin for a 32-row tile and run on
// the VALU.  One wave owns 64 Gaussians (two 32-column tiles).
//
// Backward recomputes the forward in registers, back-propagates through the same fragments (W^T swizzle) and forms
// every weight gradient dW = dH X^T on the matrix cores too: the two operands are transposed through a per-wave LDS
// staging tile (Gaussian becomes the K index), accumulated per workgroup in LDS across a persistent loop over
// tiles, and flushed with one float atomic per weight per workgroup.
#include "mom_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kHid = 64;
constexpr int kFragFloats = 2 * 2 * 16 * 64;  // one 64x64 layer as MFMA A fragments
constexpr int kLayers = 4;                    // trunk + 3 head hidden layers
constexpr int kStageStride = 33;              // LDS staging rows: [feature][32 gaussians + 1 pad]

__device__ __forceinline__ int fmap(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Wf[L][mt][kt][r][lane] = W_L[32mt + (lane&31)][32kt + fmap(r, lane>>5)]          (forward:  out = W  in)
// Wb[L][mt][kt][r][lane] = W_L[32kt + fmap(r, lane>>5)][32mt + (lane&31)]          (backward: din = W^T dout)
__global__ void __launch_bounds__(256) mlp_prep_kernel(const float* W0, const float* W1a, const float* W1b, const float* W1c,
                                                      float* __restrict__ Wf, float* __restrict__ Wb)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= kLayers * kFragFloats) return;
    const int L = idx / kFragFloats, rem = idx % kFragFloats;
    const int lane = rem & 63, r = (rem >> 6) & 15, kt = (rem >> 10) & 1, mt = (rem >> 11) & 1;
    const float* W = L == 0 ? W0 : (L == 1 ? W1a : (L == 2 ? W1b : W1c));
    const int m = 32 * mt + (lane & 31), k = 32 * kt + fmap(r, lane >> 5);
    Wf[idx] = W[m * kHid + k];
    Wb[idx] = W[k * kHid + m];
}

// out[mt][ct] (+)= sum over k of frag(mt, k) * in[k]  for the wave's NCT 32-Gaussian column tiles
template <int NCT>
__device__ __forceinline__ void layer64(const float* __restrict__ frag, const f32x16 (&in)[2][NCT], f32x16 (&out)[2][NCT], int lane)
{
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int kt = 0; kt < 2; kt++) {
            float a[16];
#pragma unroll
            for (int r = 0; r < 16; r++) a[r] = frag[((mt * 2 + kt) * 16 + r) * 64 + lane];
#pragma unroll
            for (int r = 0; r < 16; r++) {
#pragma unroll
                for (int ct = 0; ct < NCT; ct++)
                    out[mt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], in[kt][ct][r], out[mt][ct], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);  // keep at most one 16-fragment group of loads in flight (register budget)
        }
}

template <int NCT>
__device__ __forceinline__ void init_bias(const float* __restrict__ b, f32x16 (&t)[2][NCT], int h)
{
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float v = b[32 * mt + fmap(r, h)];
#pragma unroll
            for (int ct = 0; ct < NCT; ct++) t[mt][ct][r] = v;
        }
}
template <int NCT>
__device__ __forceinline__ void zero_tile(f32x16 (&t)[2][NCT])
{
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int ct = 0; ct < NCT; ct++)
#pragma unroll
            for (int r = 0; r < 16; r++) t[mt][ct][r] = 0.f;
}
template <int NCT>
__device__ __forceinline__ void relu_tile(const f32x16 (&s)[2][NCT], f32x16 (&d)[2][NCT])
{
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int ct = 0; ct < NCT; ct++)
#pragma unroll
            for (int r = 0; r < 16; r++) d[mt][ct][r] = fmaxf(s[mt][ct][r], 0.f);
}

// feat [P][64] row-major -> T layout (lane = gaussian column, registers = features)
template <int NCT>
__device__ __forceinline__ void load_feat(const float* __restrict__ feat, int g0, int P, int col, int h, f32x16 (&t)[2][NCT])
{
#pragma unroll
    for (int ct = 0; ct < NCT; ct++) {
        const int g = g0 + 32 * ct + col;
        const bool ok = g < P;
        const float* row = feat + (size_t)(ok ? g : 0) * kHid;
#pragma unroll
        for (int kt = 0; kt < 2; kt++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float4 v = *reinterpret_cast<const float4*>(row + 32 * kt + 8 * q + 4 * h);
                if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
                t[kt][ct][4 * q + 0] = v.x;
                t[kt][ct][4 * q + 1] = v.y;
                t[kt][ct][4 * q + 2] = v.z;
                t[kt][ct][4 * q + 3] = v.w;
            }
    }
}
template <int NCT>
__device__ __forceinline__ void store_feat(float* __restrict__ feat, int g0, int P, int col, int h, const f32x16 (&t)[2][NCT])
{
#pragma unroll
    for (int ct = 0; ct < NCT; ct++) {
        const int g = g0 + 32 * ct + col;
        if (g >= P) continue;
        float* row = feat + (size_t)g * kHid;
#pragma unroll
        for (int kt = 0; kt < 2; kt++)
#pragma unroll
            for (int q = 0; q < 4; q++)
                *reinterpret_cast<float4*>(row + 32 * kt + 8 * q + 4 * h) =
                    make_float4(t[kt][ct][4 * q + 0], t[kt][ct][4 * q + 1], t[kt][ct][4 * q + 2], t[kt][ct][4 * q + 3]);
    }
}

__device__ __forceinline__ float other_half(float v) { return __shfl_xor(v, 32); }

// select one of three kernel-argument pointers without indexing the argument struct at run time (which would
// force it into scratch memory)
template <class T>
__device__ __forceinline__ T* pick(T* const (&p)[3], int i) { return i == 0 ? p[0] : (i == 1 ? p[1] : p[2]); }

struct MlpDev {
    const float *b0, *b1[3], *W2[3], *b2[3];
    const float *Wf, *Wb;  // pre-swizzled fragments [4][kFragFloats]
    float *dW0, *db0, *dW1[3], *db1[3], *dW2[3], *db2[3];
};

// thin output layer on the VALU: o[n] = b2[n] + sum_f W2[n][f] a1[f], n < nout <= 4; every lane ends with the full sum
// for its gaussian (the two lane halves hold complementary feature subsets)
template <int NCT>
__device__ __forceinline__ void out_layer(const float* __restrict__ W2, const float* __restrict__ b2, int nout, const f32x16 (&a1)[2][NCT],
                                          int h, float (&o)[NCT][4])
{
#pragma unroll
    for (int n = 0; n < 4; n++) {
        float p[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ct++) p[ct] = 0.f;
        const int nn = n < nout ? n : 0;
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float4 w = *reinterpret_cast<const float4*>(W2 + nn * kHid + 32 * mt + 8 * q + 4 * h);
#pragma unroll
                for (int ct = 0; ct < NCT; ct++)
                    p[ct] += w.x * a1[mt][ct][4 * q] + w.y * a1[mt][ct][4 * q + 1] + w.z * a1[mt][ct][4 * q + 2] + w.w * a1[mt][ct][4 * q + 3];
            }
#pragma unroll
        for (int ct = 0; ct < NCT; ct++) o[ct][n] = p[ct] + other_half(p[ct]) + b2[nn];
    }
}

template <int NCT>
__global__ void __launch_bounds__(256)
deform_fwd_kernel(MlpDev m, int P, int tiles, const float* __restrict__ feat, const float* __restrict__ xyz,
                  const float* __restrict__ scaling, const float* __restrict__ rotation, const float* __restrict__ flow,
                  float flow_coef, float* __restrict__ pts, float* __restrict__ scales, float* __restrict__ rots)
{
    const int lane = threadIdx.x & 63, col = lane & 31, h = lane >> 5;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (gridDim.x * 256) >> 6;
    for (int t = wave; t < tiles; t += nwaves) {
        const int g0 = t * 32 * NCT;
        f32x16 a0[2][NCT];
        {
            f32x16 x[2][NCT];
            load_feat(feat, g0, P, col, h, x);
            init_bias(m.b0, a0, h);
            layer64(m.Wf, x, a0, lane);
        }
        relu_tile(a0, a0);
#pragma nounroll
        for (int head = 0; head < 3; head++) {
            f32x16 h1[2][NCT];
            init_bias(pick(m.b1, head), h1, h);
            layer64(m.Wf + (1 + head) * kFragFloats, a0, h1, lane);
            relu_tile(h1, h1);
            const int nout = head == 2 ? 4 : 3;
            float o[NCT][4];
            out_layer(pick(m.W2, head), pick(m.b2, head), nout, h1, h, o);
            const float* __restrict__ base = head == 0 ? xyz : (head == 1 ? scaling : rotation);
            float* __restrict__ dst = head == 0 ? pts : (head == 1 ? scales : rots);
#pragma unroll
            for (int ct = 0; ct < NCT; ct++) {
                const int g = g0 + 32 * ct + col;
                if (h == 0 && g < P) {
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (k < nout) {
                            float v = o[ct][k];
                            if (head == 0) v += flow_coef * flow[3 * g + k];
                            dst[nout * g + k] = base[nout * g + k] + v;
                        }
                }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------- backward
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v)
{
    const int o = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
    return v + __int_as_float(o);
}
// sum over the 32 lanes of each wave half; the result is valid in lanes 16..31 (half 0) and 48..63 (half 1)
__device__ __forceinline__ float half_sum(float v)
{
    v = dpp_add<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
    v = dpp_add<0x124, 0xF>(v);  // row_ror:4
    v = dpp_add<0x128, 0xF>(v);  // row_ror:8
    v = dpp_add<0x142, 0xA>(v);  // row_bcast:15 -> rows 1 and 3 add the row before them
    return v;
}

// LDS map (floats)
constexpr int kLdsDW = 0;                          // [4][64][64]
constexpr int kLdsDW2 = kLdsDW + 4 * 64 * 64;      // [3][4][64]
constexpr int kLdsDB = kLdsDW2 + 3 * 4 * 64;       // [4][64]
constexpr int kLdsDB2 = kLdsDB + 4 * 64;           // [3][4] (+4 pad)
constexpr int kLdsStage = kLdsDB2 + 16;            // [4 waves][2 operands][64][33]
constexpr int kStageFloats = 2 * 64 * kStageStride;
constexpr int kLdsTotal = kLdsStage + 4 * kStageFloats;

// dW[L] += dH X^T for this wave's 64 gaussians: both operands go through the wave's LDS staging tile so that the
// gaussian becomes the MFMA K index; the 64x64 result is added into the workgroup's LDS accumulator.
template <int NCT>
__device__ __forceinline__ void weight_grad(float* __restrict__ lds, float* __restrict__ stage, int L, const f32x16 (&dH)[2][NCT],
                                            const f32x16 (&X)[2][NCT], int lane)
{
    const int col = lane & 31, h = lane >> 5;
    float* sA = stage;
    float* sB = stage + 64 * kStageStride;
    float* dW = lds + kLdsDW + L * 64 * 64;
#pragma unroll
    for (int ct = 0; ct < NCT; ct++) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int f = 32 * mt + fmap(r, h);
                sA[f * kStageStride + col] = dH[mt][ct][r];
                sB[f * kStageStride + col] = X[mt][ct][r];
            }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
            for (int s2 = 0; s2 < 16; s2++) {
                const int g = 2 * s2 + h;
                const float av = sA[(32 * mt + col) * kStageStride + g];
                const float b0 = sB[col * kStageStride + g], b1 = sB[(32 + col) * kStageStride + g];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc1, 0, 0, 0);
            }
            // acc{kt}: row o = 32mt + fmap(r,h) (output feature), column i = 32kt + col (input feature)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                atomicAdd(&dW[(32 * mt + fmap(r, h)) * 64 + col], acc0[r]);
                atomicAdd(&dW[(32 * mt + fmap(r, h)) * 64 + 32 + col], acc1[r]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// db[L][f] += sum over this wave's gaussians of dH[f][g]
template <int NCT>
__device__ __forceinline__ void bias_grad(float* __restrict__ lds, int L, const f32x16 (&dH)[2][NCT], int lane)
{
    const int h = lane >> 5;
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float sacc = 0.f;
#pragma unroll
            for (int ct = 0; ct < NCT; ct++) sacc += dH[mt][ct][r];
            const float t = half_sum(sacc);
            if ((lane & 31) == 31) atomicAdd(&lds[kLdsDB + L * 64 + 32 * mt + fmap(r, h)], t);
            if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
}

template <int NCT>
__global__ void __launch_bounds__(256)
deform_bwd_kernel(MlpDev m, int P, int tiles, const float* __restrict__ feat, const float* __restrict__ dpts,
                  const float* __restrict__ dscales, const float* __restrict__ drots, float* __restrict__ dfeat)
{
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < kLdsStage; i += 256) lds[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, col = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6;
    float* stage = lds + kLdsStage + wv * kStageFloats;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (gridDim.x * 256) >> 6;
    for (int t = wave; t < tiles; t += nwaves) {
        const int g0 = t * 32 * NCT;
        f32x16 a0[2][NCT], dA0[2][NCT];
        {
            f32x16 x[2][NCT];
            load_feat(feat, g0, P, col, h, x);
            init_bias(m.b0, a0, h);
            layer64(m.Wf, x, a0, lane);
        }
        relu_tile(a0, a0);
        zero_tile(dA0);
#pragma nounroll
        for (int head = 0; head < 3; head++) {
            const int nout = head == 2 ? 4 : 3;
            f32x16 a1[2][NCT];
            init_bias(pick(m.b1, head), a1, h);
            layer64(m.Wf + (1 + head) * kFragFloats, a0, a1, lane);
            relu_tile(a1, a1);
            // gradient of this head's output for my two gaussians
            const float* __restrict__ dsrc = head == 0 ? dpts : (head == 1 ? dscales : drots);
            float dout[NCT][4];
#pragma unroll
            for (int ct = 0; ct < NCT; ct++) {
                const int g = g0 + 32 * ct + col;
#pragma unroll
                for (int k = 0; k < 4; k++) dout[ct][k] = (g < P && k < nout) ? dsrc[nout * g + k] : 0.f;
            }
            // output layer: db2, dW2 (half-wave reductions), then dH1 = relu'(h1) * W2^T dout in place of a1
            const float* __restrict__ W2 = pick(m.W2, head);
#ifndef MOM_DBG_SKIP_DW2
#pragma unroll
            for (int n = 0; n < 4; n++) {
                if (n < nout) {   // wave-uniform
                    float sb = 0.f;
#pragma unroll
                    for (int ct = 0; ct < NCT; ct++) sb += dout[ct][n];
                    const float tb = half_sum(sb);
                    if (lane == 31) atomicAdd(&lds[kLdsDB2 + head * 4 + n], tb);
#pragma unroll
                    for (int mt = 0; mt < 2; mt++)
#pragma unroll
                        for (int r = 0; r < 16; r++) {
                            float sw = 0.f;
#pragma unroll
                            for (int ct = 0; ct < NCT; ct++) sw += dout[ct][n] * a1[mt][ct][r];
                            const float tw = half_sum(sw);
                            if (col == 31) atomicAdd(&lds[kLdsDW2 + (head * 4 + n) * 64 + 32 * mt + fmap(r, h)], tw);
                            if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // bound the scheduler's look-ahead (registers)
                        }
                }
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    float4 w[4];
#pragma unroll
                    for (int n = 0; n < 4; n++)
                        w[n] = n < nout ? *reinterpret_cast<const float4*>(W2 + n * kHid + 32 * mt + 8 * q + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int ct = 0; ct < NCT; ct++) {
                        const float d0 = dout[ct][0], d1 = dout[ct][1], d2 = dout[ct][2], d3 = dout[ct][3];
                        const float v0 = w[0].x * d0 + w[1].x * d1 + w[2].x * d2 + w[3].x * d3;
                        const float v1 = w[0].y * d0 + w[1].y * d1 + w[2].y * d2 + w[3].y * d3;
                        const float v2 = w[0].z * d0 + w[1].z * d1 + w[2].z * d2 + w[3].z * d3;
                        const float v3 = w[0].w * d0 + w[1].w * d1 + w[2].w * d2 + w[3].w * d3;
                        a1[mt][ct][4 * q + 0] = a1[mt][ct][4 * q + 0] > 0.f ? v0 : 0.f;
                        a1[mt][ct][4 * q + 1] = a1[mt][ct][4 * q + 1] > 0.f ? v1 : 0.f;
                        a1[mt][ct][4 * q + 2] = a1[mt][ct][4 * q + 2] > 0.f ? v2 : 0.f;
                        a1[mt][ct][4 * q + 3] = a1[mt][ct][4 * q + 3] > 0.f ? v3 : 0.f;
                    }
                }
            // a1 now holds dH1
            __builtin_amdgcn_sched_barrier(0);
#ifndef MOM_DBG_SKIP_BG
            bias_grad(lds, 1 + head, a1, lane);
#endif
            __builtin_amdgcn_sched_barrier(0);
#ifndef MOM_DBG_SKIP_WG
            weight_grad(lds, stage, 1 + head, a1, a0, lane);
#endif
            __builtin_amdgcn_sched_barrier(0);
            layer64(m.Wb + (1 + head) * kFragFloats, a1, dA0, lane);   // dA0 += W1^T dH1
            __builtin_amdgcn_sched_barrier(0);
        }
        // through the ReLU between trunk and heads: dH0 = relu'(h0) * dA0
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int ct = 0; ct < NCT; ct++)
#pragma unroll
                for (int r = 0; r < 16; r++) dA0[mt][ct][r] = a0[mt][ct][r] > 0.f ? dA0[mt][ct][r] : 0.f;
        bias_grad(lds, 0, dA0, lane);
        {
            f32x16 x[2][NCT];
            load_feat(feat, g0, P, col, h, x);
            weight_grad(lds, stage, 0, dA0, x, lane);
        }
        {
            f32x16 df[2][NCT];
            zero_tile(df);
            layer64(m.Wb, dA0, df, lane);   // dfeat = W0^T dH0
            store_feat(dfeat, g0, P, col, h, df);
        }
    }
    __syncthreads();
    // flush the workgroup's accumulators
    for (int i = threadIdx.x; i < 4 * 64 * 64; i += 256) {
        const float v = lds[kLdsDW + i];
        const int L = i >> 12, e = i & 4095;
        float* dst = L == 0 ? m.dW0 : m.dW1[L - 1];
        if (v != 0.f) atomicAdd(&dst[e], v);
    }
    for (int i = threadIdx.x; i < 3 * 4 * 64; i += 256) {
        const int head = i >> 8, n = (i >> 6) & 3, f = i & 63;
        const int nout = head == 2 ? 4 : 3;
        const float v = lds[kLdsDW2 + i];
        if (n < nout && v != 0.f) atomicAdd(&m.dW2[head][n * 64 + f], v);
    }
    for (int i = threadIdx.x; i < 4 * 64; i += 256) {
        const float v = lds[kLdsDB + i];
        const int L = i >> 6, f = i & 63;
        float* dst = L == 0 ? m.db0 : m.db1[L - 1];
        if (v != 0.f) atomicAdd(&dst[f], v);
    }
    if (threadIdx.x < 12) {
        const int head = threadIdx.x >> 2, n = threadIdx.x & 3;
        const int nout = head == 2 ? 4 : 3;
        const float v = lds[kLdsDB2 + threadIdx.x];
        if (n < nout && v != 0.f) atomicAdd(&m.db2[head][n], v);
    }
}

}  // namespace

static int fill_dev(const MomDeformMLP* w, void* scratch, MlpDev* d)
{
    if (!w || !scratch || !w->W0 || !w->b0) return MOM_EINVAL;
    char* base = mom_align_ptr(scratch);
    d->Wf = (const float*)base;
    d->Wb = (const float*)(base + sizeof(float) * kLayers * kFragFloats);
    d->b0 = w->b0;
    d->dW0 = w->dW0; d->db0 = w->db0;
    for (int i = 0; i < 3; i++) {
        if (!w->W1[i] || !w->b1[i] || !w->W2[i] || !w->b2[i]) return MOM_EINVAL;
        d->b1[i] = w->b1[i]; d->W2[i] = w->W2[i]; d->b2[i] = w->b2[i];
        d->dW1[i] = w->dW1[i]; d->db1[i] = w->db1[i]; d->dW2[i] = w->dW2[i]; d->db2[i] = w->db2[i];
    }
    return MOM_OK;
}

extern "C" size_t mom_deform_scratch_bytes(void) { return 2 * sizeof(float) * kLayers * kFragFloats + 2 * MOM_ALIGN; }

extern "C" int mom_deform_prepare(const MomDeformMLP* w, void* scratch, mom_stream_t stream)
{
    MlpDev d;
    int rc = fill_dev(w, scratch, &d);
    if (rc) return rc;
    hipLaunchKernelGGL(mlp_prep_kernel, dim3((kLayers * kFragFloats + 255) / 256), dim3(256), 0, (hipStream_t)stream, w->W0, w->W1[0],
                       w->W1[1], w->W1[2], (float*)d.Wf, (float*)d.Wb);
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}

extern "C" int mom_deform_forward(const MomDeformMLP* w, int P, const float* feat, const float* xyz, const float* scaling,
                                  const float* rotation, const float* scene_flow, float flow_coef, float* pts, float* scales,
                                  float* rots, void* scratch, mom_stream_t stream)
{
    if (P < 0) return MOM_EINVAL;
    if (P == 0) return MOM_OK;
    if (!feat || !xyz || !scaling || !rotation || !scene_flow || !pts || !scales || !rots) return MOM_EINVAL;
    MlpDev d;
    int rc = fill_dev(w, scratch, &d);
    if (rc) return rc;
    constexpr int NCT = 2;
    const int tiles = (P + 32 * NCT - 1) / (32 * NCT);
    int blocks = (tiles + 3) / 4;
    if (blocks > 1024) blocks = 1024;
    MomProfScope ps(MOM_P_MLP_FWD, (hipStream_t)stream);
    hipLaunchKernelGGL(deform_fwd_kernel<NCT>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d, P, tiles, feat, xyz, scaling, rotation,
                       scene_flow, flow_coef, pts, scales, rots);
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}

extern "C" int mom_deform_backward(const MomDeformMLP* w, int P, const float* feat, const float* dpts, const float* dscales,
                                   const float* drots, float* dfeat, void* scratch, mom_stream_t stream)
{
    if (P < 0) return MOM_EINVAL;
    if (P == 0) return MOM_OK;
    if (!feat || !dpts || !dscales || !drots || !dfeat) return MOM_EINVAL;
    MlpDev d;
    int rc = fill_dev(w, scratch, &d);
    if (rc) return rc;
    if (!d.dW0 || !d.db0) return MOM_EINVAL;
    for (int i = 0; i < 3; i++)
        if (!d.dW1[i] || !d.db1[i] || !d.dW2[i] || !d.db2[i]) return MOM_EINVAL;
    constexpr int NCT = 1;
    const int tiles = (P + 32 * NCT - 1) / (32 * NCT);
    int blocks = (tiles + 3) / 4;
    if (blocks > 256) blocks = 256;   // persistent: one workgroup per CU, LDS-resident gradient accumulators
    static bool attr_set = false;
    const size_t lds_bytes = sizeof(float) * kLdsTotal;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(deform_bwd_kernel<NCT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes) != hipSuccess)
            return MOM_ELAUNCH;
        attr_set = true;
    }
    MomProfScope ps(MOM_P_MLP_BWD, (hipStream_t)stream);
    hipLaunchKernelGGL(deform_bwd_kernel<NCT>, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, d, P, tiles, feat, dpts, dscales, drots,
                       dfeat);
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}
