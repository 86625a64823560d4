// The deformation MLP's backward in ONE kernel on the bf16 matrix pipe: d(features), every weight and bias gradient; the four
// [P,64] pre-activation gradients (dH) never leave the CU.  gfx950.
//
// Replaces autograd through Deformation.forward_dynamic's trunk and pos / scales / rotations heads (reference
// scene/deformation.py:53-65,97-153).  What it computes is what deform_bwd_dx_kernel + deform_bwd_dw_kernel (deform_mlp.hip)
// compute; what changes is where the intermediates live:
//
//   two kernels (rounds 1-3):  dx writes dH [4][P][64] and dfeat (256 MB at 200 k Gaussians), dW reads dH, a0 three times and feat
//                              (309 MB) on a second stream BESIDE the HexPlane backward, which it slows from 83 + 81 us to
//                              116 + 139 us: that stretch of the step is bound by memory bandwidth;
//   round 3's one-kernel f32 form: no dH traffic, but all 640 v_mfma_f32_32x32x2_f32 of a tile on one wave's critical path,
//                              and that instruction blocks the SIMD's vector ALU: 266 us against 183;
//   here:  the fp32-exact three-way bf16 split of deform_field.hip's forward (six v_mfma_f32_32x32x16_bf16 per product block,
//          0.375 of the f32 matrix cycles, vector ALU free meanwhile) and FOUR ROLES, each cut in two by deform_bwd_b3g_kernel
//          further down (the roles as such:
//            waves 0-2, "head k":  a1 = relu(W1_k a0 + b1_k) recomputed, dW2_k / db2_k, dH1_k = relu'(.) W2_k^T dout_k,
//                                  dA0_k = W1_k^T dH1_k handed to the trunk wave through LDS, dW1_k += dH1_k^T a0, db1_k;
//            wave 3, "trunk":      dH0 = relu'(a0) (dA0_0 + dA0_1 + dA0_2), dfeat = W0^T dH0, dW0 += dH0^T feat, db0.)
//          A role needs TWO weight-fragment sets, not seven: a head wave keeps W1_k's forward fragments (96 registers) for its
//          whole share of the Gaussians and reads W1_k^T's from LDS (3 x 24 KB, pre-split once per workgroup); the trunk wave
//          keeps W0^T's in registers.  Nothing is split per tile except activations.  (The per-tile, on-the-fly split of the
//          weights is what held deform_bwd_dx_kernel<B3> to 116 us: 3360 of its ~4500 vector instructions per tile.)
//          A wave holds ONE 64x64 weight-gradient tile in accumulators (64 registers) -- the one-wave-does-everything form
//          needed four (256) and could not keep them.
//
// Per tile of 32 Gaussians: heads 144 MFMAs each, trunk 96; vector work is the splits of the activations (16 split8 per head)
// and the thin output layer.  HBM traffic: a0 (read by all four roles of a CU at about the same time: once from HBM), feat, the
// ten output gradients, dfeat: ~155 MB at 200 k Gaussians against ~640 MB.
//
// Numerics: every product is the six-term bf16 expansion of deform_b3_dev.h (terms below 2^-23 |x||y| dropped), accumulated in
// fp32 by the MFMA: the same arithmetic as the forward kernel and as deform_bwd_dx_kernel<B3>; against the f32-MFMA kernels the
// results differ by summation order (tests/test_ops_gpu.py: dfeat to 2e-5 of scale, weight gradients to 2e-5 of scale).
// (-DMOM_SPLIT_RNE: the operand split by v_cvt_pk_bf16_f32 instead of masks, deform_b3_dev.h -- 11 instead of ~13.5 vector
// instructions per pair of values.  Measured 185 against 182 us: no gain, so the masks stay, the same arithmetic as the forward.)
#include "deform_b3_dev.h"
#include <stdlib.h>
#include <mutex>

namespace {

constexpr int kXS = 36;                                    // staging rows [feature][32 gaussians + 4]: 16-byte aligned
constexpr int kFragU4 = 3 * 2 * 4 * 64;                    // one matrix orientation, [piece][mt][s][lane] uint4 = 24 KB
// LDS map (bytes)

// A tile's products, smallest terms first (as layer_b3 in deform_field.hip)
__device__ __forceinline__ f32x16 mfma6(const Frag3& A, const Frag3& B, f32x16 c)
{
    c = mfma16(A.p[2], B.p[0], c);
    c = mfma16(A.p[0], B.p[2], c);
    c = mfma16(A.p[1], B.p[1], c);
    c = mfma16(A.p[1], B.p[0], c);
    c = mfma16(A.p[0], B.p[1], c);
    c = mfma16(A.p[0], B.p[0], c);
    return c;
}
// acc[mt] += W x with the A fragments in registers
__device__ __forceinline__ void layer_regs(const Frag3 (&wf)[2][4], const Frag3 (&B)[4], f32x16 (&acc)[2])
{
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int s = 0; s < 4; s++) acc[mt] = mfma6(wf[mt][s], B[s], acc[mt]);
}
// ... with the A fragments pre-split in LDS ([piece][mt][s][lane])
__device__ __forceinline__ void layer_lds(const uint4* __restrict__ wf, const Frag3 (&B)[4], f32x16 (&acc)[2], int lane)
{
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int s = 0; s < 4; s++) {
            Frag3 A;
#pragma unroll
            for (int p = 0; p < 3; p++) A.p[p] = wf[((p * 2 + mt) * 4 + s) * 64 + lane];
            acc[mt] = mfma6(A, B[s], acc[mt]);
        }
}
// stage a T-layout tile (lane = gaussian, registers = features) as [feature][gaussian] rows of stride kXS
__device__ __forceinline__ void stage36(float* __restrict__ s, const f32x16 (&t)[2], int col, int h)
{
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int r = 0; r < 16; r++) s[(32 * mt + fmap(r, h)) * kXS + col] = t[mt][r];
}
// The weight gradient's operands, lane = feature: K-step ks covers Gaussians 16 ks + 8 h + j of the tile, j = 0..7.
//   A (from the staged dH^T): rows = out features 32 mt + col;  also returns the sum of the eight values (the bias gradient)
__device__ __forceinline__ Frag3 dw_a_frag(const float* __restrict__ s, int mt, int ks, int col, int h, float& bias_acc)
{
    const float4 lo = *reinterpret_cast<const float4*>(s + (32 * mt + col) * kXS + 16 * ks + 8 * h);
    const float4 hi = *reinterpret_cast<const float4*>(s + (32 * mt + col) * kXS + 16 * ks + 8 * h + 4);
    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    bias_acc += ((lo.x + lo.y) + (lo.z + lo.w)) + ((hi.x + hi.y) + (hi.z + hi.w));
    return split8(v);
}
//   B (X = a0 or feat, straight from memory: the rows were just read in the other layout, they come out of the L1 / L2):
//   columns = in features 32 nt + col.  Requested early (x_request), split late (x_frag).
struct XRows {
    float v[2][2][8];                                      // [nt][ks][j]
};
__device__ __forceinline__ void x_request(const float* __restrict__ X, int tile, int P, int col, int h, XRows& x)
{
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int g = tile * 32 + 16 * ks + 8 * h + j;
            const float* row = X + (size_t)(g < P ? g : P - 1) * kHid;      // rows past the end: any finite value (their dH is zero)
            x.v[0][ks][j] = row[col];
            x.v[1][ks][j] = row[32 + col];
        }
}

// A workgroup's partial sums: [4 layers (W0, W1_0..2)][64*64 weights + 64 biases] floats, then [3 heads][4*64 + 4] for the thin
// output layers.  The workgroups leave them in scratch with plain stores and deform_bwd_reduce_kernel adds them up: 256 workgroups
// adding 16 640 floats each with float atomics onto the same 16 640 addresses cost 63 of the kernel's 193 us (ablation).
constexpr int kPartLayer = kHid * kHid + kHid;
constexpr int kPartThin = 4 * kHid + 4;
constexpr int kPartFloats = 4 * kPartLayer + 3 * kPartThin;
__device__ __forceinline__ void flush_dw(float* __restrict__ part /* this layer's slice of the workgroup's partial */, const f32x16 (&dW)[2][2],
                                          const float (&db)[2], int col, int h)
{
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) part[(32 * mt + fmap(r, h)) * kHid + 32 * nt + col] = dW[mt][nt][r];
    // bias: lane (col, h) summed the Gaussians 8 h .. 8 h + 7 (+16) of every tile for out feature 32 mt + col
#pragma unroll
    for (int mt = 0; mt < 2; mt++) {
        const float b = db[mt] + __shfl_xor(db[mt], 32);
        if (h == 0) part[kHid * kHid + 32 * mt + col] = b;
    }
}
// dst[i] += sum over the workgroups' partials (the gradients are ACCUMULATED into, mom4d.h).  A block takes 64 consecutive
// elements; its 1024 threads are 16 groups of 64, group q sums the partials q, q + 16, ... (every load instruction of a wave reads
// 256 contiguous bytes), the groups meet in LDS, and one thread per element adds the total.  No atomics: one thread per element
// walking all 256 partials took 30 us (a chain of dependent round trips), sixteen or sixty-four atomics per element 28 / 43 us.
constexpr int kReduceGroups = 16;
__global__ void __launch_bounds__(64 * kReduceGroups) deform_bwd_reduce_kernel(MlpDev m, const float* __restrict__ parts, int nparts)
{
    __shared__ float s_sum[kReduceGroups][64];
    const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + e;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < kPartFloats) {
        int w = q;
        for (; w + 3 * kReduceGroups < nparts; w += 4 * kReduceGroups) {
            s0 += parts[(size_t)w * kPartFloats + i];
            s1 += parts[(size_t)(w + kReduceGroups) * kPartFloats + i];
            s2 += parts[(size_t)(w + 2 * kReduceGroups) * kPartFloats + i];
            s3 += parts[(size_t)(w + 3 * kReduceGroups) * kPartFloats + i];
        }
        for (; w < nparts; w += kReduceGroups) s0 += parts[(size_t)w * kPartFloats + i];
    }
    s_sum[q][e] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (q != 0 || i >= kPartFloats) return;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < kReduceGroups; k++) s += s_sum[k][e];
    if (s == 0.f) return;
    float* dst;
    if (i < 4 * kPartLayer) {
        const int L = i / kPartLayer, j = i - L * kPartLayer;
        float* W = L == 0 ? m.dW0 : m.dW1[L - 1];
        float* b = L == 0 ? m.db0 : m.db1[L - 1];
        dst = j < kHid * kHid ? W + j : b + (j - kHid * kHid);
    } else {
        const int k = (i - 4 * kPartLayer) / kPartThin, j = (i - 4 * kPartLayer) - k * kPartThin;
        const int nout = k == 2 ? 4 : 3;
        if (j < 4 * kHid) {
            if ((j >> 6) >= nout) return;
            dst = m.dW2[k] + j;
        } else {
            if (j - 4 * kHid >= nout) return;
            dst = m.db2[k] + (j - 4 * kHid);
        }
    }
    *dst += s;
}

__device__ __forceinline__ int ld_acquire(const int* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void st_release(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void wait_ge(const int* p, int v)
{
    // (the polling interval -- s_sleep 0 .. 4 -- does not matter: measured with -DB3G_STAMPS)
    while (__builtin_amdgcn_readfirstlane(ld_acquire(p)) < v) __builtin_amdgcn_s_sleep(2);
}

// ===========================================================================================================================
// The kernel: EIGHT role waves per workgroup, two per SIMD.  Its predecessor -- four role waves, one per SIMD, one whole role each:
// rounds 4-5, in the git history up to round 5 and described in docs/DESIGN_round_5.md 3.2 -- exposed nearly all of its MFMA time
// (ablation: 41 of 43 us) and every LDS / memory round trip, because a wave cannot issue its vector instructions under its own
// dependent MFMAs (182-190 us alone against 151 for this one).  Here every role is cut in two along its data flow, and the two halves -- waves w
// and w + 4, which the hardware places on the same SIMD -- work on consecutive tiles at the same time:
//   head k, wave A (k):      a1 = relu(W1_k a0 + b1_k) [W1_k fragments in registers], dW2_k / db2_k, dH1_k; stages dH1_k^T for B
//   head k, wave B (k + 4):  dA0_k = W1_k^T dH1_k [fragments from LDS] -> added into the trunk's slot; dW1_k += dH1_k^T a0, db1_k
//   trunk,  wave A (3):      dH0 = relu'(a0) (sum of the heads' dA0); stages dH0^T for B; dfeat = W0^T dH0 [fragments in registers]
//   trunk,  wave B (7):      dW0 += dH0^T feat, db0
// A role half fits 256 registers (no wave holds more than one fragment set AND one 64 x 64 weight-gradient tile), and LDS holds one
// staging area per hand-over instead of one per wave: 148 KB.  Hand-overs are single-buffered: the producer works a tile ahead in
// registers and waits for the consumer's release only before it stages.  The heads' dA0 meet in one slot in a fixed order (below).
// Where the cut goes was measured (-DB3G_STAMPS: cycles alive / polling per role wave): as above, the head's wave A is active 78 % of
// a tile and its wave B 88 %, the trunk's waves 40 %: 150-158 us.  dW2_k formed by the trunk's wave B instead: wave A drops to 53 %,
// the kernel stays at 158 (wave B's chain is the bound).  dA0_k formed by wave A from the registers dH1 is in (no scalar re-read by B):
// wave A 80 %, wave B 49 %, 170 us -- the longer chain is what counts, not the SIMD's total.
constexpr int kG_OffFrag = 0;                                   // W1_k^T fragments, k = 0..2
constexpr int kG_OffSA = kG_OffFrag + 3 * kFragU4 * 16;         // [3 heads][64][kXS] a1^T: private to the head's A wave
constexpr int kG_OffSH = kG_OffSA + 3 * 64 * kXS * 4;           // [3 heads][64][kXS] dH1^T: A -> B
constexpr int kG_OffST = kG_OffSH + 3 * 64 * kXS * 4;           // [64][kXS] dH0^T: trunk A -> trunk B
constexpr int kG_OffXch = kG_OffST + 64 * kXS * 4;              // [32 values][64 lanes] floats: the sum of the heads' dA0
constexpr int kG_OffDout = kG_OffXch + 32 * 64 * 4;             // [3 heads][32][4] floats
constexpr int kG_OffW2 = kG_OffDout + 3 * 32 * 4 * 4;           // [3][4][64] floats (rows >= nout zero)
constexpr int kG_OffB1 = kG_OffW2 + 3 * 4 * 64 * 4;             // [3][64] floats
constexpr int kG_OffFlag = kG_OffB1 + 3 * 64 * 4;               // pubA[3] relB[3] pubB[3] relT pubTA relTB
constexpr int kG_LdsBytes = kG_OffFlag + 64;
static_assert(kG_LdsBytes <= 160 * 1024, "the eight-wave MLP backward must fit a CU's LDS");
enum { kF_PubA = 0, kF_RelB = 3, kF_PubB = 6, kF_RelT = 9, kF_PubTA = 10, kF_RelTB = 11 };

// dW += dH^T X for one tile: A from the staged dH^T, B from the rows requested at the top of the tile; db += row sums of dH
__device__ __forceinline__ void dw_phase(const float* __restrict__ sH, const XRows& xr, f32x16 (&dW)[2][2], float (&db)[2], int col, int h)
{
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
        Frag3 A[2], B[2];
#pragma unroll
        for (int mt = 0; mt < 2; mt++) A[mt] = dw_a_frag(sH, mt, ks, col, h, db[mt]);
#pragma unroll
        for (int nt = 0; nt < 2; nt++) B[nt] = split8(xr.v[nt][ks]);
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int nt = 0; nt < 2; nt++) dW[mt][nt] = mfma6(A[mt], B[nt], dW[mt][nt]);
    }
}

#ifdef B3G_STAMPS      // per role wave: cycles alive and cycles spent polling a hand-over flag (tools/probe/b3f_stamps.py)
#define G_WAIT(p, v) do { const unsigned long long w0_ = __builtin_amdgcn_s_memtime(); wait_ge(p, v); waited_ += __builtin_amdgcn_s_memtime() - w0_; } while (0)
#else
#define G_WAIT(p, v) wait_ge(p, v)
#endif
__global__ void __launch_bounds__(512, 1)
deform_bwd_b3g_kernel(MlpDev m, int P, int tiles, const float* __restrict__ feat, const float* __restrict__ a0g,
                      const float* __restrict__ dpts, const float* __restrict__ dscales, const float* __restrict__ drots,
                      float* __restrict__ dfeat, float* __restrict__ parts)
{
    float* __restrict__ part = parts + (size_t)blockIdx.x * kPartFloats;
    extern __shared__ char lds_raw[];
    uint4* fragT = reinterpret_cast<uint4*>(lds_raw + kG_OffFrag);
    float* xch = reinterpret_cast<float*>(lds_raw + kG_OffXch);
    float* sW2 = reinterpret_cast<float*>(lds_raw + kG_OffW2);
    float* sB1 = reinterpret_cast<float*>(lds_raw + kG_OffB1);
    int* flags = reinterpret_cast<int*>(lds_raw + kG_OffFlag);
    const int lane = threadIdx.x & 63, col = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6;
    const int pair = wv & 3, half = wv >> 2;               // waves w and w + 4 share a SIMD

    // ---- prologue, all eight waves
    for (int slot = threadIdx.x; slot < 3 * 2 * 4 * 64; slot += 512) {
        const int k = slot >> 9, mt = (slot >> 8) & 1, s = (slot >> 6) & 3, ln = slot & 63;
        const int mrow = 32 * mt + (ln & 31), k0 = 16 * s + 4 * (ln >> 5);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = m.W1[k][(k0 + (j & 3) + 8 * (j >> 2)) * kHid + mrow];
        const Frag3 f = split8(v);
#pragma unroll
        for (int p = 0; p < 3; p++) fragT[k * kFragU4 + ((p * 2 + mt) * 4 + s) * 64 + ln] = f.p[p];
    }
    for (int i = threadIdx.x; i < 3 * 4 * kHid; i += 512) {
        const int head = i >> 8, n = (i >> 6) & 3;
        const int nout = head == 2 ? 4 : 3;
        sW2[i] = n < nout ? m.W2[head][n * kHid + (i & 63)] : 0.f;
    }
    if (threadIdx.x < 3 * kHid) sB1[threadIdx.x] = m.b1[threadIdx.x >> 6][threadIdx.x & 63];
    if (threadIdx.x < 16) flags[threadIdx.x] = 0;
    __syncthreads();

    const int t_begin = (int)((long long)tiles * blockIdx.x / gridDim.x), t_end = (int)((long long)tiles * (blockIdx.x + 1) / gridDim.x);
#ifdef B3G_STAMPS
    unsigned long long waited_ = 0;
    const unsigned long long start_ = __builtin_amdgcn_s_memtime();
#endif

    if (pair < 3 && half == 0) {
        // =================================================================================== head k, wave A
        const int k = pair, nout = k == 2 ? 4 : 3;
        const float* __restrict__ dsrc = k == 0 ? dpts : (k == 1 ? dscales : drots);
        float* sA = reinterpret_cast<float*>(lds_raw + kG_OffSA) + k * 64 * kXS;
        float* sH = reinterpret_cast<float*>(lds_raw + kG_OffSH) + k * 64 * kXS;
        float* sD = reinterpret_cast<float*>(lds_raw + kG_OffDout) + k * 32 * 4;
        const float* __restrict__ W2l = sW2 + k * 4 * kHid;
        Frag3 wf[2][4];
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const float* row = m.W1[k] + (32 * mt + col) * kHid + 16 * s + 4 * h;
                const float4 lo = *reinterpret_cast<const float4*>(row), hi = *reinterpret_cast<const float4*>(row + 8);
                const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                wf[mt][s] = split8(v);
            }
        float dW2[4] = {0.f, 0.f, 0.f, 0.f}, db2[4] = {0.f, 0.f, 0.f, 0.f};
        f32x16 a0n[2];
        if (t_begin < t_end) load_feat(a0g, t_begin * 32 + col, t_begin * 32 + col < P, h, a0n);
        for (int t = t_begin; t < t_end; t++) {
            const int g = t * 32 + col;
            const bool ok = g < P;
            float dout[4];
#pragma unroll
            for (int q = 0; q < 4; q++) dout[q] = (ok && q < nout) ? dsrc[nout * g + q] : 0.f;
            f32x16 a1[2];
            {
                Frag3 Ba0[4];
                split_tile<false>(a0n, Ba0);
                init_bias(sB1 + k * kHid, a1, h);
                layer_regs(wf, Ba0, a1);
            }
            if (t + 1 < t_end) load_feat(a0g, (t + 1) * 32 + col, (t + 1) * 32 + col < P, h, a0n);
            relu_tile(a1);
            __builtin_amdgcn_wave_barrier();
            stage36(sA, a1, col, h);                         // a1^T: read back by this wave only (dW2 below)
            if (h == 0) *reinterpret_cast<float4*>(sD + 4 * col) = make_float4(dout[0], dout[1], dout[2], dout[3]);
            __builtin_amdgcn_wave_barrier();
            // dH1 = relu'(h1) * W2^T dout, in place of a1
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
            G_WAIT(flags + kF_RelB + k, t - t_begin);       // wave B has finished with the previous tile's dH1^T
            stage36(sH, a1, col, h);
            st_release(flags + kF_PubA + k, t - t_begin + 1);
            {   // output layer: dW2[n][f] += sum_g dout[n][g] a1[f][g]   (lane = f);  db2[n] += dout[n][this lane's Gaussian]
#pragma unroll
                for (int q = 0; q < 4; q++) db2[q] += dout[q];
                float w0 = 0.f, w1 = 0.f, w2 = 0.f, w3 = 0.f;
#pragma unroll
                for (int g4 = 0; g4 < 32; g4 += 4) {
                    const float4 v4 = *reinterpret_cast<const float4*>(sA + lane * kXS + g4);
                    const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const float4 d = *reinterpret_cast<const float4*>(sD + 4 * (g4 + u));
                        w0 += d.x * vv[u]; w1 += d.y * vv[u]; w2 += d.z * vv[u]; w3 += d.w * vv[u];
                    }
                }
                dW2[0] += w0; dW2[1] += w1; dW2[2] += w2; dW2[3] += w3;
            }
        }
        float* thin = part + 4 * kPartLayer + k * kPartThin;
#pragma unroll
        for (int n = 0; n < 4; n++) thin[n * kHid + lane] = dW2[n];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            float b = db2[q];
#pragma unroll
            for (int d = 16; d >= 1; d >>= 1) b += __shfl_xor(b, d);
            if (lane == 0) thin[4 * kHid + q] = b;
        }
    } else if (pair < 3) {
        // =================================================================================== head k, wave B
        const int k = pair;
        const float* sH = reinterpret_cast<const float*>(lds_raw + kG_OffSH) + k * 64 * kXS;
        const uint4* __restrict__ wT = fragT + k * kFragU4;
        f32x16 dW[2][2];
        float db[2] = {0.f, 0.f};
        zero_tile(dW[0]);
        zero_tile(dW[1]);
        for (int t = t_begin; t < t_end; t++) {
            XRows xr;
            x_request(a0g, t, P, col, h, xr);               // a0 in the weight gradient's layout, used at the end of the tile
            G_WAIT(flags + kF_PubA + k, t - t_begin + 1);
            f32x16 dA0[2];
            zero_tile(dA0);
            {
                // the B operand of W1_k^T dH1: K-step s holds the features 16 s + 4 h + (j & 3) + 8 (j >> 2), j = 0..7 -- read from the
                // staged transposed copy, eight dwords per K-step
                Frag3 Bd[4];
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) v[j] = sH[(16 * s + 4 * h + (j & 3) + 8 * (j >> 2)) * kXS + col];
                    Bd[s] = split8(v);
                }
                layer_lds(wT, Bd, dA0, lane);
            }
            // The three heads' dA0 meet in ONE slot, in a fixed order: head 0 writes (once the trunk has read the previous tile's sum),
            // head 1 adds to it, head 2 adds to that -- (d0 + d1) + d2, the one-wave kernel's order, bit for bit.  (Three slots do not
            // fit beside the staging areas; ds_add_f32 from the three waves at once cost 160 us per launch: LDS float atomics.)
            if (k == 0) {
                G_WAIT(flags + kF_RelT, t - t_begin);
#pragma unroll
                for (int mt = 0; mt < 2; mt++)
#pragma unroll
                    for (int r = 0; r < 16; r++) xch[(mt * 16 + r) * 64 + lane] = dA0[mt][r];
            } else {
                G_WAIT(flags + kF_PubB + k - 1, t - t_begin + 1);
#pragma unroll
                for (int mt = 0; mt < 2; mt++)
#pragma unroll
                    for (int r = 0; r < 16; r++) xch[(mt * 16 + r) * 64 + lane] += dA0[mt][r];
            }
            st_release(flags + kF_PubB + k, t - t_begin + 1);
            dw_phase(sH, xr, dW, db, col, h);
            st_release(flags + kF_RelB + k, t - t_begin + 1);
        }
        flush_dw(part + (1 + k) * kPartLayer, dW, db, col, h);
    } else if (half == 0) {
        // =================================================================================== trunk, wave A
        float* sT = reinterpret_cast<float*>(lds_raw + kG_OffST);
        Frag3 wf[2][4];
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int s = 0; s < 4; s++) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = m.W0[(16 * s + 4 * h + (j & 3) + 8 * (j >> 2)) * kHid + 32 * mt + col];
                wf[mt][s] = split8(v);
            }
        f32x16 a0n[2];
        if (t_begin < t_end) load_feat(a0g, t_begin * 32 + col, t_begin * 32 + col < P, h, a0n);
        for (int t = t_begin; t < t_end; t++) {
            const int g = t * 32 + col;
            const bool ok = g < P;
            G_WAIT(flags + kF_PubB + 2, t - t_begin + 1);   // head 2 adds last
            f32x16 dH0[2];
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int r = 0; r < 16; r++) dH0[mt][r] = a0n[mt][r] > 0.f ? xch[(mt * 16 + r) * 64 + lane] : 0.f;
            st_release(flags + kF_RelT, t - t_begin + 1);    // head 0 may write the next tile's
            if (t + 1 < t_end) load_feat(a0g, (t + 1) * 32 + col, (t + 1) * 32 + col < P, h, a0n);
            G_WAIT(flags + kF_RelTB, t - t_begin);          // wave B has finished with the previous tile's dH0^T
            stage36(sT, dH0, col, h);
            st_release(flags + kF_PubTA, t - t_begin + 1);
            {
                Frag3 Bd[4];
                split_tile<false>(dH0, Bd);
                f32x16 df[2];
                zero_tile(df);
                layer_regs(wf, Bd, df);                     // dfeat = W0^T dH0
                store_feat(dfeat, g, ok, h, df);
            }
        }
    } else {
        // =================================================================================== trunk, wave B
        const float* sT = reinterpret_cast<const float*>(lds_raw + kG_OffST);
        f32x16 dW[2][2];
        float db[2] = {0.f, 0.f};
        zero_tile(dW[0]);
        zero_tile(dW[1]);
        for (int t = t_begin; t < t_end; t++) {
            XRows xr;
            x_request(feat, t, P, col, h, xr);
            G_WAIT(flags + kF_PubTA, t - t_begin + 1);
            dw_phase(sT, xr, dW, db, col, h);
            st_release(flags + kF_RelTB, t - t_begin + 1);
        }
        flush_dw(part, dW, db, col, h);
    }
#ifdef B3G_STAMPS
    if (lane == 0) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(parts + (size_t)256 * kPartFloats) + ((size_t)blockIdx.x * 8 + wv) * 2;
        dbg[0] = __builtin_amdgcn_s_memtime() - start_;
        dbg[1] = waited_;
    }
#endif
}

constexpr int kMaxDevices = 64;
struct PerDevice { bool attr_set = false; };
PerDevice per_device[kMaxDevices];
std::mutex per_device_mutex;

}  // namespace

// The entry point deform_mlp.hip's mom_deform_backward_split dispatches to by default (MOM_MLP_BWD unset or "b3").  (It takes the
// public descriptor: MlpDev lives in an unnamed namespace, a different type in every translation unit.)
size_t mom_deform_bwd_b3f_scratch_bytes(void) { return (size_t)256 * kPartFloats * sizeof(float); }

int mom_launch_deform_bwd_b3f(const MomDeformMLP* w, int P, const float* feat, const float* a0, const float* dpts, const float* dscales,
                              const float* drots, float* dfeat, void* scratch, hipStream_t s, hipStream_t dw_stream)
{
    MlpDev d;
    int rc = fill_dev(w, &d);
    if (rc) return rc;
    if (!d.dW0 || !d.db0) return MOM_EINVAL;
    for (int i = 0; i < 3; i++)
        if (!d.dW1[i] || !d.db1[i] || !d.dW2[i] || !d.db2[i]) return MOM_EINVAL;
    const int tiles = (P + 31) / 32;
    // Persistent: one workgroup of four role waves per CU, and each wave takes its SIMD's whole register file -- nothing co-runs on
    // a CU this kernel holds.  A caller with a second stream (the training step: the appearance parameters' Adam launch is waiting
    // there) gets 224 workgroups, 28 per XCD: the 32 CUs left over let that HBM-bound launch run UNDER this kernel (2.5 TB/s is all
    // it needs) instead of beside the HexPlane backward afterwards, whose gather it slowed from 83 to 116 us.  Measured, steps/s at
    // 256 / 240 / 232 / 224 / 216 / 208 workgroups: config 2 1047 / 1040 / 1035 / 1061 / 1057 / 1054, config 3 285 / 282 / 280 /
    // 291 / 289 / 287, config 5 91.1 / - / - / 93.1 (a workgroup's tile count steps from 25 to 27 at 240 and to 28 at 224: 240 pays
    // for too few free CUs).  MOM_B3F_BLOCKS overrides.
    static int forced_blocks = -1;
    if (forced_blocks < 0) { const char* e = getenv("MOM_B3F_BLOCKS"); forced_blocks = (e && atoi(e) > 0 && atoi(e) <= 256) ? atoi(e) : 0; }
    const int max_blocks = forced_blocks ? forced_blocks : (dw_stream != s ? 224 : 256);
    const int blocks = tiles < max_blocks ? tiles : max_blocks;
    int dev_id = 0;
    if (hipGetDevice(&dev_id) != hipSuccess || dev_id < 0 || dev_id >= kMaxDevices) return MOM_ELAUNCH;
    PerDevice& pd = per_device[dev_id];
    {
        // per-device lazy state (the dynamic-LDS attribute is per device; the hand-over event belongs to the device it was created
        // on): a process may drive several devices, from several threads.  A failed attribute call is retried by the next launch.
        std::lock_guard<std::mutex> lock(per_device_mutex);
        if (!pd.attr_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(deform_bwd_b3g_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    kG_LdsBytes) != hipSuccess)
                return MOM_ELAUNCH;
            pd.attr_set = true;
        }
    }
    MomProfScope ps(MOM_P_MLP_BWD, s);
    hipLaunchKernelGGL(deform_bwd_b3g_kernel, dim3(blocks), dim3(512), kG_LdsBytes, s, d, P, tiles, feat, a0, dpts, dscales, drots, dfeat,
                       (float*)scratch);
    if (hipGetLastError() != hipSuccess) return MOM_ELAUNCH;
    // The sum over the workgroups' partials is all that is left of "the weight gradients are complete on dw_stream": it goes to the
    // caller's second stream, behind an event.  On `stream` it sat between the MLP backward and the HexPlane backward and, in the
    // training step, shared HBM with the early Adam launch: 46 us on the critical path for a kernel that takes 6 alone.
    static int reduce_on_main = -1;      // MOM_B3F_REDUCE_MAIN=1: keep the reduction on `stream` (measurement)
    if (reduce_on_main < 0) { const char* e = getenv("MOM_B3F_REDUCE_MAIN"); reduce_on_main = (e && e[0] == '1') ? 1 : 0; }
    if (reduce_on_main) dw_stream = s;
    if (dw_stream != s) {
        // one event per call: record + wait capture the state at the record, and hipEventDestroy of a recorded event is deferred
        // until it has completed, so nothing is shared between the calls of different threads or streams
        hipEvent_t main_done = nullptr;
        if (hipEventCreateWithFlags(&main_done, mom_order_event_flags()) != hipSuccess) return MOM_ELAUNCH;
        const bool ok = hipEventRecord(main_done, s) == hipSuccess && hipStreamWaitEvent(dw_stream, main_done, 0) == hipSuccess;
        (void)hipEventDestroy(main_done);
        if (!ok) return MOM_ELAUNCH;
    }
    hipLaunchKernelGGL(deform_bwd_reduce_kernel, dim3((kPartFloats + 63) / 64), dim3(64 * kReduceGroups), 0, dw_stream, d, (const float*)scratch,
                       blocks);
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}
