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
// accumulator tile then has exactly the register layout the next layer needs as ITS B operand (register r of lane
// half h is feature (r&3)+8(r>>2)+4h), so activations never leave registers between layers.
//
// The four 64x64 matrices live in LDS for the whole (persistent) kernel as [in][out] with a row stride of 65
// floats: the forward A fragment (lane = out) reads consecutive banks, the backward A fragment of W^T (lane = in)
// reads banks 65 apart -- both conflict-free from ONE copy.  The 64->{3,3,4} output layers are too thin for a
// 32-row tile and run on the VALU.  One wave owns 32 Gaussians per step of its persistent loop.
//
// Backward is two streaming kernels.  (A) deform_bwd_dx: reloads relu(h0) saved by the forward, recomputes each
// head's hidden layer, forms dH (gradient at every pre-activation), back-propagates through W^T down to d(features)
// and writes the four dH matrices [P][64]; the thin output layers' gradients are row sums over an LDS staging tile,
// kept in registers across the persistent loop.  (B) deform_bwd_dw: dW_L = dH_L^T X_L for the four 64x64 layers on
// the matrix cores with the Gaussian as the MFMA K index -- because dH and X are stored [gaussian][feature], both
// MFMA operands are plain coalesced 256-byte loads (lane = feature), every byte is read exactly once, the four
// 64x64 results live in 256 accumulator registers for the wave's whole range, and the bias gradients fall out of
// the A operands for free.  No atomics until one float atomic per weight per workgroup at the very end.
#include "deform_mlp_dev.h"
#include "deform_b3_dev.h"
#include <stdlib.h>

namespace {

// All threads of a workgroup share ONE copy of the weights in LDS (70 KB).  With 256 threads two workgroups = 8 waves fit per
// CU although the registers allow 16: a workgroup of up to 1024 threads gets up to 16 waves for the same LDS.
__global__ void __launch_bounds__(1024)
deform_fwd_kernel(MlpDev m, int P, int tiles, const float* __restrict__ feat, const float* __restrict__ xyz,
                  const float* __restrict__ scaling, const float* __restrict__ rotation, const float* __restrict__ flow,
                  float flow_coef, float* __restrict__ pts, float* __restrict__ scales, float* __restrict__ rots,
                  float* __restrict__ a0_save, ActOut act)
{
    extern __shared__ float lds[];
    load_weights(m, lds);
    __syncthreads();
    const int lane = threadIdx.x & 63, col = lane & 31, h = lane >> 5;
    // Every workgroup takes a contiguous, equal (+-1) share of the tiles and deals it to its waves.  Dealing tiles to ALL the
    // launch's waves in turn left the remainder with the first workgroups only: at 6250 tiles on 4096 waves 134 CUs worked
    // through 8 tiles per SIMD while 122 had 4 -- the launch lasted 8 where the mean is 6.1.
    const int t_begin = (int)((long long)tiles * blockIdx.x / gridDim.x), t_end = (int)((long long)tiles * (blockIdx.x + 1) / gridDim.x);
    const int t_first = t_begin + (int)(threadIdx.x >> 6), t_step = (int)(blockDim.x >> 6);
    for (int t = t_first; t < t_end; t += t_step) {
        const int g = t * 32 + col;
        const bool ok = g < P;
        f32x16 a0[2];
        {
            f32x16 x[2];
            load_feat(feat, g, ok, h, x);
            init_bias(lds + kLB, a0, h);
            layer64<false>(lds + kLW, x, a0, col, h);
        }
        relu_tile(a0);
        if (a0_save) store_feat(a0_save, g, ok, h, a0);     // relu(h0), reused by the backward kernels
#pragma nounroll
        for (int head = 0; head < 3; head++) {
            f32x16 h1[2];
            init_bias(lds + kLB + (1 + head) * kHid, h1, h);
            layer64<false>(lds + kLW + (1 + head) * kWFloats, a0, h1, col, h);
            relu_tile(h1);
            float o[4];
            out_layer(lds + kLW2 + head * 4 * kHid, lds + kLB2 + head * 4, h1, h, o);
            if (h == 0 && ok) {
                if (head == 0) {
#pragma unroll
                    for (int k = 0; k < 3; k++) pts[3 * g + k] = xyz[3 * g + k] + (o[k] + flow_coef * flow[3 * g + k]);
                } else if (head == 1) {
#pragma unroll
                    for (int k = 0; k < 3; k++) scales[3 * g + k] = scaling[3 * g + k] + o[k];
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++) rots[4 * g + k] = rotation[4 * g + k] + o[k];
                }
            }
        }
    }
    // Activated copies (mom_deform_forward_activated), after the tile loop: inside it their temporaries pushed the kernel over
    // its 128-register budget (100 bytes of scratch per lane, +15 us).  Each lane re-reads what it stored itself.
    if (act.scales || act.rots || act.opacity) {
        for (int t = t_first; t < t_end; t += t_step) {
            const int g = t * 32 + col;
            if (h != 0 || g >= P) continue;
            if (act.scales) {
#pragma unroll
                for (int k = 0; k < 3; k++) act.scales[3 * g + k] = expf(scales[3 * g + k]);
            }
            if (act.rots) {
                const float4 q = *reinterpret_cast<const float4*>(rots + 4 * g);
                const float n = mom_quat_norm(q.x, q.y, q.z, q.w);
                *reinterpret_cast<float4*>(act.rots + 4 * g) = make_float4(q.x / n, q.y / n, q.z / n, q.w / n);
            }
            if (act.opacity) act.opacity[g] = mom_sigmoid(act.opacity_raw[g]);
        }
    }
}

#ifndef MOM_DX_PREFETCH
#define MOM_DX_PREFETCH 1               // measured: mlp_bwd slot 177.4 -> 173.7 us (three runs each)
#endif
// ---------------------------------------------------------------------------------------------- backward
// (A) activations backward: dH for the four layers, d(features), output-layer weight gradients
// 512 threads: two waves per SIMD share one copy of the weights (142 KB of LDS with the staging tiles), so that one wave's vector
// and memory phases -- the thin output layers, the dH stores -- run under the other's MFMAs (4 waves: 288 us for dx + dW, 8: 276)
// B3: the seven 64x64 products of a tile on the bf16 matrix pipe from exact three-way splits (deform_b3_dev.h), the weights split
// on the fly from their fp32 copy in LDS -- pre-split fragments for seven matrices (168 KB) do not fit beside the staging tiles.
// 0.375 of the f32 kernel's matrix cycles, and those co-issue with the other wave's vector work, which the f32 MFMA blocks.
template <bool B3>
__global__ void __launch_bounds__(64 * kDxWaves)
deform_bwd_dx_kernel(MlpDev m, int P, int tiles, const float* __restrict__ a0g, const float* __restrict__ dpts,
                     const float* __restrict__ dscales, const float* __restrict__ drots, float* __restrict__ dfeat,
                     float* __restrict__ dH /* [4][P][64] */)
{
    extern __shared__ float lds[];
    load_weights(m, lds);
    __syncthreads();
    const int lane = threadIdx.x & 63, col = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6;
    float* sA = lds + kLStage + wv * kStageFloats;
    float* sD = sA + kHid * kStageStride;              // dout[32 gaussians][4]
    // Every workgroup takes a contiguous, equal (+-1) share of the tiles and deals it to its waves.  Dealing tiles to ALL the
    // launch's waves in turn left the remainder with the first workgroups only: at 6250 tiles on 4096 waves 134 CUs worked
    // through 8 tiles per SIMD while 122 had 4 -- the launch lasted 8 where the mean is 6.1.
    const int t_begin = (int)((long long)tiles * blockIdx.x / gridDim.x), t_end = (int)((long long)tiles * (blockIdx.x + 1) / gridDim.x);
    const int t_first = t_begin + (int)(threadIdx.x >> 6), t_step = (int)(blockDim.x >> 6);
    const size_t PH = (size_t)P * kHid;
    float dW2[3][4], db2[3];                           // lane = feature; db2: lane n < 4 holds output n
#pragma unroll
    for (int k = 0; k < 3; k++) { db2[k] = 0.f; dW2[k][0] = dW2[k][1] = dW2[k][2] = dW2[k][3] = 0.f; }

#if MOM_DX_PREFETCH
    // the next tile's trunk activations are requested while this tile is worked on (a wave otherwise starts every tile with
    // a memory round trip, and its SIMD partner is not always in a matrix phase to cover it)
    f32x16 a0n[2];
    if (t_first < t_end) load_feat(a0g, t_first * 32 + col, t_first * 32 + col < P, h, a0n);
#endif
    for (int t = t_first; t < t_end; t += t_step) {
        const int g = t * 32 + col;
        const bool ok = g < P;
        f32x16 a0[2], dA0[2];
#if MOM_DX_PREFETCH
        a0[0] = a0n[0];
        a0[1] = a0n[1];
        if (t + t_step < t_end) load_feat(a0g, (t + t_step) * 32 + col, (t + t_step) * 32 + col < P, h, a0n);
#else
        load_feat(a0g, g, ok, h, a0);
#endif
        zero_tile(dA0);
        Frag3 Ba0[4];
        if (B3) split_tile<false>(a0, Ba0);        // the recomputed head layers' B operand, split once per tile
#pragma nounroll
        for (int head = 0; head < 3; head++) {   // rolled on purpose: unrolled, the scheduler interleaves the heads and spills
            const int nout = head == 2 ? 4 : 3;
            f32x16 a1[2];
            init_bias(lds + kLB + (1 + head) * kHid, a1, h);
            if (B3) layer_b3_otf<false>(lds + kLW + (1 + head) * kWFloats, Ba0, a1, col, h);
            else layer64<false>(lds + kLW + (1 + head) * kWFloats, a0, a1, col, h);
            relu_tile(a1);
            const float* __restrict__ dsrc = head == 0 ? dpts : (head == 1 ? dscales : drots);
            float dout[4];
#pragma unroll
            for (int k = 0; k < 4; k++) dout[k] = (ok && k < nout) ? dsrc[nout * g + k] : 0.f;
            __builtin_amdgcn_wave_barrier();
            stage_tile(sA, a1, col, h);
            if (h == 0) *reinterpret_cast<float4*>(sD + 4 * col) = make_float4(dout[0], dout[1], dout[2], dout[3]);
            __builtin_amdgcn_wave_barrier();
            {   // output layer: dW2[n][f] += sum_g dout[n][g] a1[f][g]; db2[n] += sum_g dout[n][g]   (lane = f)
                float w0 = 0.f, w1 = 0.f, w2 = 0.f, w3 = 0.f, bsum = 0.f;
#pragma unroll
                for (int gg = 0; gg < 32; gg++) {
                    const float v = sA[lane * kStageStride + gg];
                    const float4 d = *reinterpret_cast<const float4*>(sD + 4 * gg);
                    w0 += d.x * v; w1 += d.y * v; w2 += d.z * v; w3 += d.w * v;
                    bsum += sD[4 * gg + (lane & 3)];
                }
                if (head == 0) { dW2[0][0] += w0; dW2[0][1] += w1; dW2[0][2] += w2; dW2[0][3] += w3; db2[0] += bsum; }
                else if (head == 1) { dW2[1][0] += w0; dW2[1][1] += w1; dW2[1][2] += w2; dW2[1][3] += w3; db2[1] += bsum; }
                else { dW2[2][0] += w0; dW2[2][1] += w1; dW2[2][2] += w2; dW2[2][3] += w3; db2[2] += bsum; }
            }
            // dH1 = relu'(h1) * W2^T dout, in place of a1
            const float* __restrict__ W2l = lds + kLW2 + head * 4 * kHid;
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float4 wa = *reinterpret_cast<const float4*>(W2l + 0 * kHid + 32 * mt + 8 * q + 4 * h);
                    const float4 wb = *reinterpret_cast<const float4*>(W2l + 1 * kHid + 32 * mt + 8 * q + 4 * h);
                    const float4 wc = *reinterpret_cast<const float4*>(W2l + 2 * kHid + 32 * mt + 8 * q + 4 * h);
                    const float4 wd = *reinterpret_cast<const float4*>(W2l + 3 * kHid + 32 * mt + 8 * q + 4 * h);
                    const float v0 = wa.x * dout[0] + wb.x * dout[1] + wc.x * dout[2] + wd.x * dout[3];
                    const float v1 = wa.y * dout[0] + wb.y * dout[1] + wc.y * dout[2] + wd.y * dout[3];
                    const float v2 = wa.z * dout[0] + wb.z * dout[1] + wc.z * dout[2] + wd.z * dout[3];
                    const float v3 = wa.w * dout[0] + wb.w * dout[1] + wc.w * dout[2] + wd.w * dout[3];
                    a1[mt][4 * q + 0] = a1[mt][4 * q + 0] > 0.f ? v0 : 0.f;
                    a1[mt][4 * q + 1] = a1[mt][4 * q + 1] > 0.f ? v1 : 0.f;
                    a1[mt][4 * q + 2] = a1[mt][4 * q + 2] > 0.f ? v2 : 0.f;
                    a1[mt][4 * q + 3] = a1[mt][4 * q + 3] > 0.f ? v3 : 0.f;
                }
            store_feat(dH + (size_t)(1 + head) * PH, g, ok, h, a1);
            if (B3) {
                Frag3 Bd[4];
                split_tile<false>(a1, Bd);
                layer_b3_otf<true>(lds + kLW + (1 + head) * kWFloats, Bd, dA0, col, h);
            } else {
                layer64<true>(lds + kLW + (1 + head) * kWFloats, a1, dA0, col, h);   // dA0 += W1^T dH1
            }
        }
        // through the ReLU between trunk and heads
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int r = 0; r < 16; r++) dA0[mt][r] = a0[mt][r] > 0.f ? dA0[mt][r] : 0.f;
        store_feat(dH, g, ok, h, dA0);
        {
            f32x16 df[2];
            zero_tile(df);
            if (B3) {
                Frag3 Bd[4];
                split_tile<false>(dA0, Bd);
                layer_b3_otf<true>(lds + kLW, Bd, df, col, h);
            } else {
                layer64<true>(lds + kLW, dA0, df, col, h);  // dfeat = W0^T dH0
            }
            store_feat(dfeat, g, ok, h, df);
        }
    }
    // output-layer gradients: combine the four waves in LDS, one atomic per element per workgroup
    __syncthreads();
    float* R = lds;
    for (int i = threadIdx.x; i < 12 * kHid + 16; i += blockDim.x) R[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int n = 0; n < 4; n++) atomicAdd(&R[(k * 4 + n) * kHid + lane], dW2[k][n]);
        if (lane < 4) atomicAdd(&R[12 * kHid + k * 4 + lane], db2[k]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 12 * kHid; i += blockDim.x) {
        const int k = i >> 8, n = (i >> 6) & 3, f = i & 63;
        const int nout = k == 2 ? 4 : 3;
        const float v = R[i];
        if (n < nout && v != 0.f) atomicAdd(&m.dW2[k][n * kHid + f], v);
    }
    if (threadIdx.x < 12) {
        const int k = threadIdx.x >> 2, n = threadIdx.x & 3;
        const int nout = k == 2 ? 4 : 3;
        const float v = R[12 * kHid + threadIdx.x];
        if (n < nout && v != 0.f) atomicAdd(&m.db2[k][n], v);
    }
}

// (B) weight gradients: dW_L[o][i] = sum_g dH_L[g][o] X_L[g][i], db_L[o] = sum_g dH_L[g][o];  X_0 = feat, X_1..3 = a0
//
// The kernel streams 12 dwords per lane per K-step and is bound by how many cache misses a CU keeps in flight, not by
// the matrix pipe (SQ counters: MFMA busy 30 % of the time, the rest issue stalls on VMEM).  With all four layers per
// wave the 256 accumulators allow one wave per SIMD; NL layers per wave (blockIdx.y picks which: layers [NL*y, NL*y+NL))
// need 64*NL accumulators, so 4/NL waves fit per SIMD and the CU has that many times more loads in flight.  dH is still read
// exactly once overall; a0 is read once per workgroup row that holds one of the layers 1..3.
template <int NL>
__global__ void __launch_bounds__(256)
deform_bwd_dw_kernel(MlpDev m, int P, int chunk, const float* __restrict__ feat, const float* __restrict__ a0g,
                     const float* __restrict__ dH, int ny)
{
    extern __shared__ float lds[];                     // [64][64] reduction scratch
    const int lane = threadIdx.x & 63, col = lane & 31, h = lane >> 5;
    // One-dimensional grid of (ranges of Gaussians) x (ny layer groups), numbered so that the ny workgroups of one range are 8
    // ids apart inside a run of 8 ny ids: workgroup ids go to the eight XCDs in turn, so the layer groups of a range land on the
    // SAME XCD at about the same time, and the rows of a0 the three head layers all read come out of that XCD's L2 twice out of
    // three times (a two-dimensional grid ran all ranges of layer 0, then all of layer 1, ...: three trips to memory).
    const int lin = blockIdx.x, run = 8 * ny;
    const int bx = (lin / run) * 8 + (lin % run) % 8, by = (lin % run) / 8;
    const int wave = (bx * 256 + threadIdx.x) >> 6;
    const int L0 = NL * by;                            // first layer of this workgroup
    const size_t PH = (size_t)P * kHid;
    const int g_begin = wave * chunk, g_end = min(P, g_begin + chunk);   // chunk is even

    f32x16 dW[NL][2][2];
    float db[NL][2];
#pragma unroll
    for (int l = 0; l < NL; l++) {
        zero_tile(dW[l][0]);
        zero_tile(dW[l][1]);
        db[l][0] = db[l][1] = 0.f;
    }
    // Operand loads are double buffered by hand.  A trip is UNR K-steps (2 gaussians each); the loads of trip t+1 are issued
    // before the MFMAs of trip t.  They are issued UNCONDITIONALLY (rows past the end are clamped and masked by `ok`): vmcnt
    // retires in order, and a prefetch behind a branch would make the compiler wait for it too.
    constexpr int UNR = 4;
    constexpr bool kNeedFeat = true, kNeedA0 = NL > 1;   // which X a row needs is uniform per workgroup: see x_of()
    struct Operands {
        float x[UNR][2], da[UNR][NL][2];
        float x2[UNR][2];                               // second X (a0) when the row spans layer 0 and later layers
        bool ok[UNR];
    };
    // layer L reads X = feat (L == 0) or a0 (L >= 1).  Rows with NL > 1 that start at layer 0 need both.
    const bool first_is_feat = L0 == 0;
    const float* __restrict__ xa = first_is_feat ? feat : a0g;
    auto load = [&](Operands& o, int g0) {
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const int g = g0 + 2 * u + h;               // K slot of this lane half
            const bool ok = g < g_end;
            o.ok[u] = ok;
            const size_t row = (size_t)(ok ? g : g_begin) * kHid;
#pragma unroll
            for (int kt = 0; kt < 2; kt++) {
                o.x[u][kt] = xa[row + 32 * kt + col];
                if (kNeedA0) o.x2[u][kt] = a0g[row + 32 * kt + col];
            }
#pragma unroll
            for (int l = 0; l < NL; l++) {
                const float* __restrict__ d = dH + (size_t)(L0 + l) * PH + row;
                o.da[u][l][0] = d[col];
                o.da[u][l][1] = d[32 + col];
            }
        }
    };
    auto compute = [&](const Operands& o) {
#pragma unroll
        for (int u = 0; u < UNR; u++) {
#pragma unroll
            for (int l = 0; l < NL; l++) {
                const float a_lo = o.ok[u] ? o.da[u][l][0] : 0.f, a_hi = o.ok[u] ? o.da[u][l][1] : 0.f;
                db[l][0] += a_lo;
                db[l][1] += a_hi;
                // l == 0 uses the row's first X; later layers of the row always read a0
                const float x0 = (l == 0 || !kNeedA0) ? o.x[u][0] : o.x2[u][0];
                const float x1 = (l == 0 || !kNeedA0) ? o.x[u][1] : o.x2[u][1];
                dW[l][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_lo, x0, dW[l][0][0], 0, 0, 0);
                dW[l][0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_lo, x1, dW[l][0][1], 0, 0, 0);
                dW[l][1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_hi, x0, dW[l][1][0], 0, 0, 0);
                dW[l][1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_hi, x1, dW[l][1][1], 0, 0, 0);
            }
        }
    };
    (void)kNeedFeat;
    if (g_begin < g_end) {
        // single rotation (compute(cur); cur = nxt): the two-phase form (A/B alternating in one body) made the register
        // allocator spill a whole operand buffer to scratch
        Operands cur, nxt;
        load(cur, g_begin);
        for (int g0 = g_begin; g0 < g_end; g0 += 2 * UNR) {
            load(nxt, g0 + 2 * UNR);
            __builtin_amdgcn_sched_barrier(0);
            compute(cur);
            __builtin_amdgcn_sched_barrier(0);
            cur = nxt;
        }
    }
    // Combine the four waves in LDS, then one float atomic per weight per workgroup.  The waves take TURNS on the shared
    // tile with plain read-add-write instead of LDS float atomics: ds_add_f32 runs at roughly 170 cycles per wave
    // instruction here, and 16 waves per CU each issuing 66 of them kept the CU's LDS busy for ~80 us -- more than the whole
    // MFMA loop (ablation: loop only 77 us, reduction only 88 us with every global atomic skipped, together 145 us).
    float* R = lds;
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int l = 0; l < NL; l++) {
        const int L = L0 + l;
        // bias: the two lane halves hold different K slots of the same output feature
        const float b_lo = db[l][0] + __shfl_xor(db[l][0], 32), b_hi = db[l][1] + __shfl_xor(db[l][1], 32);
#pragma unroll 1
        for (int turn = 0; turn < 4; turn++) {
            __syncthreads();
            if (wv == turn) {
#pragma unroll
                for (int mt = 0; mt < 2; mt++)
#pragma unroll
                    for (int kt = 0; kt < 2; kt++)
#pragma unroll
                        for (int r = 0; r < 16; r++) {     // tile row = out feature 32mt+fmap(r,h), column = in feature 32kt+col
                            float* a = &R[(32 * mt + fmap(r, h)) * kHid + 32 * kt + col];
                            *a = (turn == 0 ? 0.f : *a) + dW[l][mt][kt][r];
                        }
                if (h == 0) {
                    float* a = &R[kHid * kHid + col];
                    a[0] = (turn == 0 ? 0.f : a[0]) + b_lo;
                    a[32] = (turn == 0 ? 0.f : a[32]) + b_hi;
                }
            }
        }
        __syncthreads();
        float* dst = L == 0 ? m.dW0 : m.dW1[L - 1];
        float* dbs = L == 0 ? m.db0 : m.db1[L - 1];
        for (int i = threadIdx.x; i < kHid * kHid; i += 256) {
            const float v = R[i];
            if (v != 0.f) atomicAdd(&dst[i], v);
        }
        if (threadIdx.x < kHid) {
            const float v = R[kHid * kHid + threadIdx.x];
            if (v != 0.f) atomicAdd(&dbs[threadIdx.x], v);
        }
    }
}

// (C) dx and dW in ONE kernel: the four dH matrices never leave the CU.
//
// The two-kernel backward writes dH [4][P][64] (205 MB at 200 k Gaussians) and reads it back with a0 three times and feat once
// (378 MB): two thirds of the MLP's traffic for an operand that exists in registers when it is needed.  Here a wave keeps the
// four 64x64 weight-gradient tiles in 256 accumulator registers for its whole share of the Gaussians (one wave per SIMD, up to
// 512 registers) and, per tile of 32 Gaussians, adds dW_L += dH_L^T X_L right after dH_L is formed: dH_L and X_L (a0 for the
// three head layers, feat for the trunk) are transposed through two LDS staging tiles -- the MFMA's K index is the Gaussian
// here, so both operands want lane = feature -- and the bias gradients fall out of the A operands.  704 MFMAs per tile either
// way; what is gone is 480 MB of traffic (only the trunk's dH still goes out, see kNR).  (v_mfma_f32_32x32x2_f32 occupies the
// vector ALU, so a second wave per SIMD would not overlap its vector work with this wave's matrix work anyway:
// tools/probe/mfma_valu_coissue.hip.)
//
// MEASURED (round 3, 200 k Gaussians).  First version: 333 us against 153 us for dx with the weight-gradient kernel hidden on the
// second stream.  Second version (a0 kept in registers as the recomputed layers' B operand instead of two LDS reads per MFMA; the
// 64 MFMAs of dA0 += W1^T dH1 placed between the staging writes and their read-back; no spills once the SLP vectoriser was off):
// 266 us against 183 (dx with Adam's early launch beside it), the HexPlane backward beside it 222 us instead of 258 -- it has the
// chip and 480 MB less traffic to share -- and the step 870 against 932 steps/s.  The arithmetic decides it: the one-kernel form
// puts all 640 f32 MFMAs per tile on the critical path (119 us at the pipe's full rate, which these kernels reach to 55 %),
// the two-kernel form 448, with the other 256 hidden on the second stream; the bytes saved buy back 36 us of the 83.
// The kernel is correct (tests/test_ops_gpu.py compares it with the two-kernel form) and stays opt-in: MOM_MLP_BWD=fused.
constexpr int kFusedStage = 2 * kHid * kStageStride + 4 * 32;          // sA | sX | dout[32][4]
constexpr int kLFusedStage = kLFwdTotal;
constexpr int kLFusedTotal = kLFusedStage + 4 * kFusedStage;
#ifndef MOM_FUSED_NR
#define MOM_FUSED_NR 3
#endif
constexpr int kNR = MOM_FUSED_NR;                       // layers whose weight gradient lives in registers (3, or 4 = trunk too)

__device__ __forceinline__ void dw_accumulate(const float* __restrict__ sA, const float* __restrict__ sX, f32x16 (&dW)[2][2], float (&db)[2],
                                              int col, int h)
{
    // K-step j: Gaussians 2 j + h of the tile; A = dH^T (row = out feature), B = X (column = in feature).  Operands are fetched
    // four K-steps ahead of their MFMAs.
    constexpr int kAhead = 4;
    float a_lo[2][kAhead], a_hi[2][kAhead], x0[2][kAhead], x1[2][kAhead];
    auto fetch = [&](int j0, int b) {
#pragma unroll
        for (int u = 0; u < kAhead; u++) {
            const int j = j0 + u;
            a_lo[b][u] = sA[col * kStageStride + 2 * j + h];
            a_hi[b][u] = sA[(32 + col) * kStageStride + 2 * j + h];
            x0[b][u] = sX[col * kStageStride + 2 * j + h];
            x1[b][u] = sX[(32 + col) * kStageStride + 2 * j + h];
        }
    };
    fetch(0, 0);
#pragma unroll
    for (int c = 0; c < 16 / kAhead; c++) {
        const int b = c & 1;
        if (c + 1 < 16 / kAhead) fetch((c + 1) * kAhead, b ^ 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < kAhead; u++) {
            db[0] += a_lo[b][u];
            db[1] += a_hi[b][u];
            dW[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_lo[b][u], x0[b][u], dW[0][0], 0, 0, 0);
            dW[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_lo[b][u], x1[b][u], dW[0][1], 0, 0, 0);
            dW[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_hi[b][u], x0[b][u], dW[1][0], 0, 0, 0);
            dW[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_hi[b][u], x1[b][u], dW[1][1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// h1 = W a0 + b with the B operand (a0, staged as [feature][gaussian]) read from LDS: a0 then needs no registers of its own
__device__ __forceinline__ void layer64_ldsB(const float* __restrict__ Wl, const float* __restrict__ sB, f32x16 (&out)[2], int col, int h)
{
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int kt = 0; kt < 2; kt++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int k = 32 * kt + fmap(r, h), mrow = 32 * mt + col;
                out[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(Wl[k * kWStride + mrow], sB[k * kStageStride + col], out[mt], 0, 0, 0);
                if (r == 15) __builtin_amdgcn_sched_barrier(0);
            }
}

__global__ void __launch_bounds__(256, 1)
deform_bwd_fused_kernel(MlpDev m, int P, int tiles, const float* __restrict__ feat, const float* __restrict__ a0g,
                        const float* __restrict__ dpts, const float* __restrict__ dscales, const float* __restrict__ drots,
                        float* __restrict__ dfeat, float* __restrict__ dH0)
{
    extern __shared__ float lds[];
    load_weights(m, lds);
    __syncthreads();
    const int lane = threadIdx.x & 63, col = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6;
    // The staging area sits above 64 KB of LDS, beyond the reach of a DS instruction's 16-bit offset field: left to constant
    // folding, every distinct (region base + row offset) became a loop-invariant address register of its own -- hundreds, spilled.
    // One opaque per-wave base keeps every staging access at `base + small immediate`.
    // One opaque per-wave base, renewed INSIDE the tile loop (a loop-invariant one is hoisted all the same), keeps every access
    // at `base + small immediate`; the weights get the same treatment.
    const unsigned stage_off0 = (unsigned)(kLFusedStage + wv * kFusedStage) * 4u;
    const int t_begin = (int)((long long)tiles * blockIdx.x / gridDim.x), t_end = (int)((long long)tiles * (blockIdx.x + 1) / gridDim.x);
    // Three of the four layers: at one wave per SIMD the compiler puts EVERY MFMA result into the 256 accumulation registers,
    // the layer tiles of the moment (64) included, which leaves 192 = three 64x64 gradients.  The trunk layer's dH is the one
    // matrix that still goes out (51 MB instead of 205), to a quarter-size run of the weight-gradient kernel.
    f32x16 dW[kNR][2][2];
    float db[kNR][2];
#pragma unroll
    for (int l = 0; l < kNR; l++) {
        zero_tile(dW[l][0]);
        zero_tile(dW[l][1]);
        db[l][0] = db[l][1] = 0.f;
    }
    float dW2[3][4], db2[3];                           // lane = feature; db2: lane n < 4 holds output n
#pragma unroll
    for (int k = 0; k < 3; k++) { db2[k] = 0.f; dW2[k][0] = dW2[k][1] = dW2[k][2] = dW2[k][3] = 0.f; }

    for (int t = t_begin + wv; t < t_end; t += 4) {
        const int g = t * 32 + col;
        const bool ok = g < P;
        unsigned stage_off = stage_off0, lds_off = 0;
        asm volatile("" : "+v"(stage_off), "+v"(lds_off));
        float* sA = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + stage_off);
        float* sX = sA + kHid * kStageStride;
        float* sD = sX + kHid * kStageStride;           // dout[32 gaussians][4]
        const float* L = reinterpret_cast<const float*>(reinterpret_cast<const char*>(lds) + lds_off);
        f32x16 dA0[2], a0[2];
        load_feat(a0g, g, ok, h, a0);                  // stays in registers: the B operand of the three recomputed head layers
        __builtin_amdgcn_wave_barrier();
        stage_tile(sX, a0, col, h);                    // and, transposed, the X operand of their weight gradients
        __builtin_amdgcn_wave_barrier();
        zero_tile(dA0);
#pragma unroll
        for (int head = 0; head < 3; head++) {
            const int nout = head == 2 ? 4 : 3;
            f32x16 a1[2];
            init_bias(L + kLB + (1 + head) * kHid, a1, h);
            layer64<false>(L + kLW + (1 + head) * kWFloats, a0, a1, col, h);
            relu_tile(a1);
            const float* __restrict__ dsrc = head == 0 ? dpts : (head == 1 ? dscales : drots);
            float dout[4];
#pragma unroll
            for (int k = 0; k < 4; k++) dout[k] = (ok && k < nout) ? dsrc[nout * g + k] : 0.f;
            __builtin_amdgcn_wave_barrier();
            stage_tile(sA, a1, col, h);
            if (h == 0) *reinterpret_cast<float4*>(sD + 4 * col) = make_float4(dout[0], dout[1], dout[2], dout[3]);
            __builtin_amdgcn_wave_barrier();
            {   // output layer: dW2[n][f] += sum_g dout[n][g] a1[f][g]; db2[n] += sum_g dout[n][g]   (lane = f)
                float w0 = 0.f, w1 = 0.f, w2 = 0.f, w3 = 0.f, bsum = 0.f;
#pragma unroll 1
                for (int g0 = 0; g0 < 32; g0 += 8)     // rolled: unrolled whole, its 160 operand registers are loaded up front
#pragma unroll
                    for (int gg = g0; gg < g0 + 8; gg++) {
                        const float v = sA[lane * kStageStride + gg];
                        const float4 d = *reinterpret_cast<const float4*>(sD + 4 * gg);
                        w0 += d.x * v; w1 += d.y * v; w2 += d.z * v; w3 += d.w * v;
                        bsum += sD[4 * gg + (lane & 3)];
                    }
                dW2[head][0] += w0; dW2[head][1] += w1; dW2[head][2] += w2; dW2[head][3] += w3; db2[head] += bsum;
            }
            // dH1 = relu'(h1) * W2^T dout, in place of a1
            const float* __restrict__ W2l = L + kLW2 + head * 4 * kHid;
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float4 wa = *reinterpret_cast<const float4*>(W2l + 0 * kHid + 32 * mt + 8 * q + 4 * h);
                    const float4 wb = *reinterpret_cast<const float4*>(W2l + 1 * kHid + 32 * mt + 8 * q + 4 * h);
                    const float4 wc = *reinterpret_cast<const float4*>(W2l + 2 * kHid + 32 * mt + 8 * q + 4 * h);
                    const float4 wd = *reinterpret_cast<const float4*>(W2l + 3 * kHid + 32 * mt + 8 * q + 4 * h);
                    const float v0 = wa.x * dout[0] + wb.x * dout[1] + wc.x * dout[2] + wd.x * dout[3];
                    const float v1 = wa.y * dout[0] + wb.y * dout[1] + wc.y * dout[2] + wd.y * dout[3];
                    const float v2 = wa.z * dout[0] + wb.z * dout[1] + wc.z * dout[2] + wd.z * dout[3];
                    const float v3 = wa.w * dout[0] + wb.w * dout[1] + wc.w * dout[2] + wd.w * dout[3];
                    a1[mt][4 * q + 0] = a1[mt][4 * q + 0] > 0.f ? v0 : 0.f;
                    a1[mt][4 * q + 1] = a1[mt][4 * q + 1] > 0.f ? v1 : 0.f;
                    a1[mt][4 * q + 2] = a1[mt][4 * q + 2] > 0.f ? v2 : 0.f;
                    a1[mt][4 * q + 3] = a1[mt][4 * q + 3] > 0.f ? v3 : 0.f;
                }
            __builtin_amdgcn_wave_barrier();
            stage_tile(sA, a1, col, h);                // dH1^T for the weight gradient (the staged a1 has been consumed)
            __builtin_amdgcn_wave_barrier();
            layer64<true>(L + kLW + (1 + head) * kWFloats, a1, dA0, col, h);     // dA0 += W1^T dH1: 64 MFMAs between the staging
            __builtin_amdgcn_sched_barrier(0);                                   // writes above and their read-back below
            dw_accumulate(sA, sX, dW[kNR - 3 + head], db[kNR - 3 + head], col, h);
            __builtin_amdgcn_sched_barrier(0);          // the heads stay apart: interleaved by the scheduler they spill
        }
        // through the ReLU between trunk and heads (a0 = relu(h0) is still staged in sX)
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                dA0[mt][r] = sX[(32 * mt + fmap(r, h)) * kStageStride + col] > 0.f ? dA0[mt][r] : 0.f;
        if constexpr (kNR == 3) {
            store_feat(dH0, g, ok, h, dA0);           // the trunk layer's weight gradient: deform_bwd_dw_kernel, layer 0 only
        } else {
            f32x16 x[2];
            load_feat(feat, g, ok, h, x);              // X of the trunk layer
            __builtin_amdgcn_wave_barrier();
            stage_tile(sA, dA0, col, h);
            stage_tile(sX, x, col, h);
            __builtin_amdgcn_wave_barrier();
            dw_accumulate(sA, sX, dW[0], db[0], col, h);
        }
        {
            f32x16 df[2];
            zero_tile(df);
            layer64<true>(L + kLW, dA0, df, col, h);    // dfeat = W0^T dH0
            store_feat(dfeat, g, ok, h, df);
        }
    }
    // ---- reductions over the workgroup's four waves, then one float atomic per weight per workgroup
    __syncthreads();
    float* R = lds;                                     // the weights are no longer needed
    for (int i = threadIdx.x; i < 12 * kHid + 16; i += blockDim.x) R[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int n = 0; n < 4; n++) atomicAdd(&R[(k * 4 + n) * kHid + lane], dW2[k][n]);
        if (lane < 4) atomicAdd(&R[12 * kHid + k * 4 + lane], db2[k]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 12 * kHid; i += blockDim.x) {
        const int k = i >> 8, n = (i >> 6) & 3, f = i & 63;
        const int nout = k == 2 ? 4 : 3;
        const float v = R[i];
        if (n < nout && v != 0.f) atomicAdd(&m.dW2[k][n * kHid + f], v);
    }
    if (threadIdx.x < 12) {
        const int k = threadIdx.x >> 2, n = threadIdx.x & 3;
        const int nout = k == 2 ? 4 : 3;
        const float v = R[12 * kHid + threadIdx.x];
        if (n < nout && v != 0.f) atomicAdd(&m.db2[k][n], v);
    }
    __syncthreads();
    // the four 64x64 layers: the waves take turns on one shared tile with plain read-add-write (LDS float atomics cost ~190
    // cycles per wave instruction: tools/probe/lds_atomic_probe.hip)
#pragma unroll
    for (int l = 0; l < kNR; l++) {
        const float b_lo = db[l][0] + __shfl_xor(db[l][0], 32), b_hi = db[l][1] + __shfl_xor(db[l][1], 32);
#pragma unroll 1
        for (int turn = 0; turn < 4; turn++) {
            __syncthreads();
            if (wv == turn) {
#pragma unroll
                for (int mt = 0; mt < 2; mt++)
#pragma unroll
                    for (int kt = 0; kt < 2; kt++)
#pragma unroll
                        for (int r = 0; r < 16; r++) {     // tile row = out feature 32mt+fmap(r,h), column = in feature 32kt+col
                            float* a = &R[(32 * mt + fmap(r, h)) * kHid + 32 * kt + col];
                            *a = (turn == 0 ? 0.f : *a) + dW[l][mt][kt][r];
                        }
                if (h == 0) {
                    float* a = &R[kHid * kHid + col];
                    a[0] = (turn == 0 ? 0.f : a[0]) + b_lo;
                    a[32] = (turn == 0 ? 0.f : a[32]) + b_hi;
                }
            }
        }
        __syncthreads();
        float* dst = (kNR == 4 && l == 0) ? m.dW0 : m.dW1[l - (kNR - 3)];
        float* dbs = (kNR == 4 && l == 0) ? m.db0 : m.db1[l - (kNR - 3)];
        for (int i = threadIdx.x; i < kHid * kHid; i += 256) {
            const float v = R[i];
            if (v != 0.f) atomicAdd(&dst[i], v);
        }
        if (threadIdx.x < kHid) {
            const float v = R[kHid * kHid + threadIdx.x];
            if (v != 0.f) atomicAdd(&dbs[threadIdx.x], v);
        }
    }
}

}  // namespace

extern "C" int mom_deform_forward(const MomDeformMLP* w, int P, const float* feat, const float* xyz, const float* scaling,
                                  const float* rotation, const float* scene_flow, float flow_coef, float* pts, float* scales,
                                  float* rots, float* a0_save, mom_stream_t stream)
{
    return mom_deform_forward_activated(w, P, feat, xyz, scaling, rotation, scene_flow, flow_coef, pts, scales, rots, a0_save, nullptr,
                                        nullptr, nullptr, nullptr, stream);
}

extern "C" int mom_deform_forward_activated(const MomDeformMLP* w, int P, const float* feat, const float* xyz, const float* scaling,
                                            const float* rotation, const float* scene_flow, float flow_coef, float* pts, float* scales,
                                            float* rots, float* a0_save, const float* opacity_raw, float* scales_act, float* rots_act,
                                            float* opacity_act, mom_stream_t stream)
{
    if (P < 0) return MOM_EINVAL;
    if (P == 0) return MOM_OK;
    if (!feat || !xyz || !scaling || !rotation || !scene_flow || !pts || !scales || !rots) return MOM_EINVAL;
    MlpDev d;
    int rc = fill_dev(w, &d);
    if (rc) return rc;
    const int tiles = (P + 31) / 32;
    // one workgroup per CU, any multiple of 64 threads from 256 to 1024 (MOM_MLP_FWD_BLOCK); the weights stay in LDS (70 KB)
    static int block = 0;
    if (!block) {
        const char* e = getenv("MOM_MLP_FWD_BLOCK");
        block = e ? atoi(e) : 1024;
        if (block < 256 || block > 1024 || (block & 63)) block = 1024;      // load_weights needs at least 256 threads
    }
    // as many workgroups as CUs can hold at once, but never more than tiles: a small problem is spread over the CUs (one tile per
    // workgroup, 39 -> 21 us at 5 k Gaussians) instead of filling sixteen waves of a few workgroups
    const int per_cu = block <= 256 ? 2 : 1;
    int blocks = tiles < 256 * per_cu ? tiles : 256 * per_cu;
    static bool attr_set = false;
    const size_t lds_bytes = sizeof(float) * kLFwdTotal;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(deform_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes) != hipSuccess)
            return MOM_ELAUNCH;
        attr_set = true;
    }
    if ((opacity_act != nullptr) != (opacity_raw != nullptr)) return MOM_EINVAL;
    const ActOut act = {scales_act, rots_act, opacity_act, opacity_raw};
    MomProfScope ps(MOM_P_MLP_FWD, (hipStream_t)stream);
    hipLaunchKernelGGL(deform_fwd_kernel, dim3(blocks), dim3(block), lds_bytes, (hipStream_t)stream, d, P, tiles, feat, xyz, scaling,
                       rotation, scene_flow, flow_coef, pts, scales, rots, a0_save, act);
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}

int mom_launch_deform_bwd_b3f(const MomDeformMLP* w, int P, const float* feat, const float* a0, const float* dpts, const float* dscales,
                              const float* drots, float* dfeat, void* scratch, hipStream_t s, hipStream_t dw_stream);      // deform_bwd_b3.hip
size_t mom_deform_bwd_b3f_scratch_bytes(void);

// dx and the head layers' dW in one kernel; only the trunk's dH (the first [P,64] of `scratch`) goes through memory
static int deform_backward_fused(const MomDeformMLP* w, int P, const float* feat, const float* a0, const float* dpts,
                                 const float* dscales, const float* drots, float* dfeat, void* scratch, mom_stream_t stream,
                                 mom_stream_t dw_stream)
{
    if (P < 0) return MOM_EINVAL;
    if (P == 0) return MOM_OK;
    if (!feat || !a0 || !dpts || !dscales || !drots || !dfeat || !scratch) return MOM_EINVAL;
    MlpDev d;
    int rc = fill_dev(w, &d);
    if (rc) return rc;
    if (!d.dW0 || !d.db0) return MOM_EINVAL;
    for (int i = 0; i < 3; i++)
        if (!d.dW1[i] || !d.db1[i] || !d.dW2[i] || !d.db2[i]) return MOM_EINVAL;
    float* dH0 = (float*)scratch;
    const int tiles = (P + 31) / 32;
    const int blocks = tiles < 256 * 4 ? (tiles + 3) / 4 : 256;      // persistent: one workgroup of four waves per CU
    static bool attr_set = false;
    const size_t lds_bytes = sizeof(float) * kLFusedTotal;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(deform_bwd_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes) != hipSuccess)
            return MOM_ELAUNCH;
        attr_set = true;
    }
    MomProfScope ps(MOM_P_MLP_BWD, (hipStream_t)stream);
    hipLaunchKernelGGL(deform_bwd_fused_kernel, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, d, P, tiles, feat, a0, dpts,
                       dscales, drots, dfeat, dH0);
    if (hipGetLastError() != hipSuccess) return MOM_ELAUNCH;
    if (kNR == 4) return MOM_OK;
    hipStream_t ws = (hipStream_t)dw_stream;
    if (ws != (hipStream_t)stream) {
        static hipEvent_t dx_done = nullptr;
        if (!dx_done && hipEventCreateWithFlags(&dx_done, mom_order_event_flags()) != hipSuccess) return MOM_ELAUNCH;
        if (hipEventRecord(dx_done, (hipStream_t)stream) != hipSuccess) return MOM_ELAUNCH;
        if (hipStreamWaitEvent(ws, dx_done, 0) != hipSuccess) return MOM_ELAUNCH;
    }
    // the trunk layer's weight gradient: the weight-gradient kernel on layer 0 only (dH laid out as its first matrix)
    const int waves = 1024;
    int chunk = (P + waves - 1) / waves;
    chunk += chunk & 1;
    hipLaunchKernelGGL(deform_bwd_dw_kernel<1>, dim3(waves / 4), dim3(256), sizeof(float) * (kHid * kHid + kHid), ws, d, P, chunk, feat,
                       a0, dH0, 1);
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}

// the larger of what the forms need: the two-kernel form's four [P,64] matrices, the one-kernel form's per-workgroup partial sums
extern "C" size_t mom_deform_backward_scratch_bytes(int P)
{
    const size_t a = (size_t)4 * (size_t)(P > 0 ? P : 1) * kHid * sizeof(float), b = mom_deform_bwd_b3f_scratch_bytes();
    return a > b ? a : b;
}

extern "C" int mom_deform_backward(const MomDeformMLP* w, int P, const float* feat, const float* a0, const float* dpts,
                                   const float* dscales, const float* drots, float* dfeat, void* scratch, mom_stream_t stream)
{
    return mom_deform_backward_split(w, P, feat, a0, dpts, dscales, drots, dfeat, scratch, stream, stream);
}

extern "C" int mom_deform_backward_split(const MomDeformMLP* w, int P, const float* feat, const float* a0, const float* dpts,
                                         const float* dscales, const float* drots, float* dfeat, void* scratch, mom_stream_t stream,
                                         mom_stream_t dw_stream)
{
    if (P < 0) return MOM_EINVAL;
    if (P == 0) return MOM_OK;
    if (!feat || !a0 || !dpts || !dscales || !drots || !dfeat || !scratch) return MOM_EINVAL;
    // MOM_MLP_BWD (read per call so that tests can compare the forms): unset or "b3" -- the one-kernel backward on the bf16 pipe
    // with role-specialised waves (deform_bwd_b3.hip; dfeat is complete on `stream`, the small reduction that completes the weight
    // gradients runs on dw_stream); "split" -- the two
    // f32 kernels below (dx on `stream`, dW on `dw_stream`); "fused" -- round 3's one-kernel f32 form (measured slower).
    const char* e = getenv("MOM_MLP_BWD");
    if (e && e[0] == 'f') return deform_backward_fused(w, P, feat, a0, dpts, dscales, drots, dfeat, scratch, stream, dw_stream);
    if (!e || e[0] == 'b')
        return mom_launch_deform_bwd_b3f(w, P, feat, a0, dpts, dscales, drots, dfeat, scratch, (hipStream_t)stream, (hipStream_t)dw_stream);
    MlpDev d;
    int rc = fill_dev(w, &d);
    if (rc) return rc;
    if (!d.dW0 || !d.db0) return MOM_EINVAL;
    for (int i = 0; i < 3; i++)
        if (!d.dW1[i] || !d.db1[i] || !d.dW2[i] || !d.db2[i]) return MOM_EINVAL;
    float* dH = (float*)scratch;
    const int tiles = (P + 31) / 32;
    int blocks = tiles < 256 ? tiles : 256;             // persistent: one workgroup per CU; a small problem still uses every CU
    static bool attr_set = false;
    const size_t lds_a = sizeof(float) * kLBwdTotal;
    const size_t lds_b = sizeof(float) * (kHid * kHid + kHid);
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(deform_bwd_dx_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_a) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(deform_bwd_dx_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_a) != hipSuccess)
            return MOM_ELAUNCH;
        attr_set = true;
    }
    MomProfScope ps(MOM_P_MLP_BWD, (hipStream_t)stream);
    // MOM_DX_MODE=b3: the products on the bf16 pipe (read per call: tests compare the two).  Measured at 200 k Gaussians: the kernel
    // alone 116 us against 144 (35.9 M vector instructions instead of 13.8 M, matrix pipe busy 67 M cycles instead of 179 M) -- and
    // the training step NOT faster (930 against 943 steps/s): the stretch from dx to the HexPlane scatter moves 1.8 GB and is
    // bound by memory bandwidth, so the 45 us this kernel gives up are taken by its neighbours (Adam's early launch no longer
    // fits beside it and runs into the gather: gather 153 -> 173 us, scatter 104 -> 131).  Opt-in for that reason.
    const char* e_dx = getenv("MOM_DX_MODE");
    if (e_dx && e_dx[0] == 'b')
        hipLaunchKernelGGL(deform_bwd_dx_kernel<true>, dim3(blocks), dim3(64 * kDxWaves), lds_a, (hipStream_t)stream, d, P, tiles, a0, dpts,
                           dscales, drots, dfeat, dH);
    else
        hipLaunchKernelGGL(deform_bwd_dx_kernel<false>, dim3(blocks), dim3(64 * kDxWaves), lds_a, (hipStream_t)stream, d, P, tiles, a0, dpts,
                           dscales, drots, dfeat, dH);
    if (hipGetLastError() != hipSuccess) return MOM_ELAUNCH;
    hipStream_t ws = (hipStream_t)dw_stream;
    if (ws != (hipStream_t)stream) {
        // the weight-gradient kernel reads what dx wrote (dH) and nothing behind it on `stream` depends on it: it goes to the
        // caller's second stream, behind an event, and overlaps whatever the caller enqueues on `stream` next
        static hipEvent_t dx_done = nullptr;
        if (!dx_done && hipEventCreateWithFlags(&dx_done, mom_order_event_flags()) != hipSuccess) return MOM_ELAUNCH;
        if (hipEventRecord(dx_done, (hipStream_t)stream) != hipSuccess) return MOM_ELAUNCH;
        if (hipStreamWaitEvent(ws, dx_done, 0) != hipSuccess) return MOM_ELAUNCH;
    }
    // weight gradients: 1024 waves, each a contiguous (even-sized) range of gaussians (MOM_DW_WAVES: a multiple of 32, measurement)
    static int waves = 0;
    if (!waves) { const char* e = getenv("MOM_DW_WAVES"); waves = e ? atoi(e) : 1024; if (waves < 32 || waves % 32) waves = 1024; }
    int chunk = (P + waves - 1) / waves;
    chunk += chunk & 1;
    // layers per wave (MOM_DW_NL = 4, 2 or 1): fewer layers -> fewer accumulators -> more waves and more loads in flight per CU
    static int nl = 0;
    if (!nl) {
        const char* e = getenv("MOM_DW_NL");
        nl = e ? atoi(e) : 1;                 // measured: dx + dW 344 / 339 / 326 us for 4 / 2 / 1 layers per wave
        if (nl != 1 && nl != 2 && nl != 4) nl = 1;
    }
    const int ny = 4 / nl;
    const dim3 grid((waves / 4) * ny);                 // one-dimensional: the kernel deals (range, layer group) to the XCDs itself
    if (nl == 4)
        hipLaunchKernelGGL(deform_bwd_dw_kernel<4>, grid, dim3(256), lds_b, ws, d, P, chunk, feat, a0, dH, ny);
    else if (nl == 2)
        hipLaunchKernelGGL(deform_bwd_dw_kernel<2>, grid, dim3(256), lds_b, ws, d, P, chunk, feat, a0, dH, ny);
    else
        hipLaunchKernelGGL(deform_bwd_dw_kernel<1>, grid, dim3(256), lds_b, ws, d, P, chunk, feat, a0, dH, ny);
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}

