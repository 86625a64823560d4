// HexPlane feature field, forward and backward, gfx950.
//
// Replaces HexPlaneField.forward -> interpolate_ms_features -> 6 x F.grid_sample(bilinear,
// align_corners=True, padding_mode='border') per level + product over planes + concat over
// levels (reference scene/hexplane.py:19-46,73-106,160-183) and the 12 scatter-add backward
// kernels autograd runs for them.
//
// Layout: every plane is stored CHANNEL-LAST, [H][W][32] floats, so one texel is one 128-byte
// line.  32 lanes (half a wave64) own one (Gaussian, level): lane c owns channel c, so every
// texel fetch of a half-wave is one coalesced 128-B line and every gradient scatter is one
// 128-B row of float atomics (the shape that runs at the full atomic rate).  The six plane
// samples and their product stay in registers; features leave as one 128-B row per half-wave.
//
// Arithmetic follows ATen's grid_sampler_2d (unnormalise with align_corners, clip to the
// border, nw/ne/sw/se weights, accumulation order nw,ne,sw,se) so that results agree with the
// reference's torch ops to rounding.
#include "mom_common.h"
#include <stdlib.h>

namespace {

struct PlaneSample {
    int i00, i01, i10, i11;      // texel indices (row-major over [H][W]), -1 when out of bounds
    float w00, w01, w10, w11;    // nw, ne, sw, se
    float gx_mul, gy_mul;        // d(ix)/d(coord) incl. border-clip mask
    float ix, iy;
    int ixn, iyn;
};

__device__ __forceinline__ float unnorm_clip(float c, int size, float& gmul)
{
    // align_corners=True: ((c+1)/2)*(size-1); border: clip to [0, size-1] with zero gradient when clipped
    float v = ((c + 1.f) / 2.f) * (float)(size - 1);
    gmul = (float)(size - 1) / 2.f;
    if (v <= 0.f) {
        v = 0.f;
        gmul = 0.f;
    } else {
        const float mx = (float)(size - 1);
        if (v >= mx) {
            v = mx;
            gmul = 0.f;
        }
    }
    return v;
}

__device__ __forceinline__ PlaneSample make_sample(float cx, float cy, int Wd, int Hd)
{
    PlaneSample s;
    s.ix = unnorm_clip(cx, Wd, s.gx_mul);
    s.iy = unnorm_clip(cy, Hd, s.gy_mul);
    const float fx = floorf(s.ix), fy = floorf(s.iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    s.ixn = x0;
    s.iyn = y0;
    s.w00 = ((float)x1 - s.ix) * ((float)y1 - s.iy);
    s.w01 = (s.ix - (float)x0) * ((float)y1 - s.iy);
    s.w10 = ((float)x1 - s.ix) * (s.iy - (float)y0);
    s.w11 = (s.ix - (float)x0) * (s.iy - (float)y0);
    const bool x0in = x0 >= 0 && x0 < Wd, x1in = x1 >= 0 && x1 < Wd, y0in = y0 >= 0 && y0 < Hd, y1in = y1 >= 0 && y1 < Hd;
    s.i00 = (x0in && y0in) ? y0 * Wd + x0 : -1;
    s.i01 = (x1in && y0in) ? y0 * Wd + x1 : -1;
    s.i10 = (x0in && y1in) ? y1 * Wd + x0 : -1;
    s.i11 = (x1in && y1in) ? y1 * Wd + x1 : -1;
    return s;
}

struct HexArgs {
    int P, levels;
    int res[4][4];
    const float* planes[4][6];
    float* grads[4][6];
    float a0[3], a1[3];  // aabb rows exactly as the reference stores them (row 0 = xyz_max, row 1 = xyz_min)
    float time;
    const float* times;  // optional per-point timestamps [P]; null -> `time` for every point
    const uint32_t* order;  // optional processing order (a permutation of 0..P-1, e.g. Morton order); null -> identity
};

__constant__ int kCombA[6] = {0, 0, 0, 1, 1, 2};
__constant__ int kCombB[6] = {1, 2, 3, 2, 3, 3};

__device__ __forceinline__ void norm_coords(const HexArgs& a, const float* __restrict__ xyz, int g, float c[4])
{
#pragma unroll
    for (int k = 0; k < 3; k++) c[k] = (xyz[3 * g + k] - a.a0[k]) * (2.0f / (a.a1[k] - a.a0[k])) - 1.0f;
    c[3] = a.times ? a.times[g] : a.time;
}

// grid: one half-wave per (Gaussian, level); blockDim 256 = 8 half-waves
__global__ void __launch_bounds__(256) hexplane_fwd_kernel(HexArgs a, const float* __restrict__ xyz, float* __restrict__ feat)
{
    const int ch = threadIdx.x & 31;
    const long long unit = (long long)blockIdx.x * 8 + (threadIdx.x >> 5);
    const int gi = (int)(unit / a.levels), lvl = (int)(unit % a.levels);
    if (gi >= a.P) return;
    const int g = a.order ? (int)a.order[gi] : gi;   // spatially sorted processing order (speed only)
    float c[4];
    norm_coords(a, xyz, g, c);
    float prod = 1.f;
#pragma unroll
    for (int p = 0; p < 6; p++) {
        const int ca = kCombA[p], cb = kCombB[p];
        const int Wd = a.res[lvl][ca], Hd = a.res[lvl][cb];
        const PlaneSample s = make_sample(c[ca], c[cb], Wd, Hd);
        const float* __restrict__ pl = a.planes[lvl][p];
        float v = 0.f;
        if (s.i00 >= 0) v += pl[(size_t)s.i00 * 32 + ch] * s.w00;
        if (s.i01 >= 0) v += pl[(size_t)s.i01 * 32 + ch] * s.w01;
        if (s.i10 >= 0) v += pl[(size_t)s.i10 * 32 + ch] * s.w10;
        if (s.i11 >= 0) v += pl[(size_t)s.i11 * 32 + ch] * s.w11;
        prod = prod * v;
    }
    feat[(size_t)g * (a.levels * 32) + lvl * 32 + ch] = prod;
}

__device__ __forceinline__ float half_wave_sum(float v)
{
    // sum over the 32 lanes of this half-wave (xor butterflies never cross bit 5)
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

__global__ void __launch_bounds__(256)
hexplane_bwd_kernel(HexArgs a, const float* __restrict__ xyz, const float* __restrict__ dfeat, float* __restrict__ dxyz)
{
    const int ch = threadIdx.x & 31;
    const long long unit = (long long)blockIdx.x * 8 + (threadIdx.x >> 5);
    const int g = (int)(unit / a.levels), lvl = (int)(unit % a.levels);
    if (g >= a.P) return;
    float c[4];
    norm_coords(a, xyz, g, c);
    PlaneSample s[6];
    float v[6], t00[6], t01[6], t10[6], t11[6];
#pragma unroll
    for (int p = 0; p < 6; p++) {
        const int ca = kCombA[p], cb = kCombB[p];
        s[p] = make_sample(c[ca], c[cb], a.res[lvl][ca], a.res[lvl][cb]);
        const float* __restrict__ pl = a.planes[lvl][p];
        t00[p] = s[p].i00 >= 0 ? pl[(size_t)s[p].i00 * 32 + ch] : 0.f;
        t01[p] = s[p].i01 >= 0 ? pl[(size_t)s[p].i01 * 32 + ch] : 0.f;
        t10[p] = s[p].i10 >= 0 ? pl[(size_t)s[p].i10 * 32 + ch] : 0.f;
        t11[p] = s[p].i11 >= 0 ? pl[(size_t)s[p].i11 * 32 + ch] : 0.f;
        float acc = 0.f;
        acc += t00[p] * s[p].w00;
        acc += t01[p] * s[p].w01;
        acc += t10[p] * s[p].w10;
        acc += t11[p] * s[p].w11;
        v[p] = acc;
    }
    const float go = dfeat[(size_t)g * (a.levels * 32) + lvl * 32 + ch];
    // prefix / suffix products -> product excluding plane p
    float pre[7], suf[7];
    pre[0] = 1.f;
#pragma unroll
    for (int p = 0; p < 6; p++) pre[p + 1] = pre[p] * v[p];
    suf[6] = 1.f;
#pragma unroll
    for (int p = 5; p >= 0; p--) suf[p] = suf[p + 1] * v[p];
    float gc[3] = {0.f, 0.f, 0.f};  // dL/d(normalised x,y,z), this channel's share
#pragma unroll
    for (int p = 0; p < 6; p++) {
        const float gv = go * pre[p] * suf[p + 1];
        float* __restrict__ gp = a.grads[lvl][p];
        if (s[p].i00 >= 0) atomicAdd(&gp[(size_t)s[p].i00 * 32 + ch], gv * s[p].w00);
        if (s[p].i01 >= 0) atomicAdd(&gp[(size_t)s[p].i01 * 32 + ch], gv * s[p].w01);
        if (s[p].i10 >= 0) atomicAdd(&gp[(size_t)s[p].i10 * 32 + ch], gv * s[p].w10);
        if (s[p].i11 >= 0) atomicAdd(&gp[(size_t)s[p].i11 * 32 + ch], gv * s[p].w11);
        // grid gradient (ATen grid_sampler_2d backward): with x1 = x0+1, y1 = y0+1
        const float x0 = (float)s[p].ixn, y0 = (float)s[p].iyn, x1 = x0 + 1.f, y1 = y0 + 1.f;
        float gix = 0.f, giy = 0.f;
        gix -= t00[p] * (y1 - s[p].iy) * gv;
        giy -= t00[p] * (x1 - s[p].ix) * gv;
        gix += t01[p] * (y1 - s[p].iy) * gv;
        giy -= t01[p] * (s[p].ix - x0) * gv;
        gix -= t10[p] * (s[p].iy - y0) * gv;
        giy += t10[p] * (x1 - s[p].ix) * gv;
        gix += t11[p] * (s[p].iy - y0) * gv;
        giy += t11[p] * (s[p].ix - x0) * gv;
        const int ca = kCombA[p], cb = kCombB[p];
        if (ca < 3) gc[ca] += gix * s[p].gx_mul;
        if (cb < 3) gc[cb] += giy * s[p].gy_mul;
    }
    if (dxyz) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float tot = half_wave_sum(gc[k]) * (2.0f / (a.a1[k] - a.a0[k]));
            if (ch == 0) atomicAdd(&dxyz[3 * g + k], tot);  // two levels add into the same slot
        }
    }
}


// ---- aggregated backward -------------------------------------------------------------------------------------
// One timestamp for every point (the render() case).  Two structural facts cut the float-atomic traffic of the
// generic kernel above (1.2 GB per call at 200k Gaussians, and worse, all of it for the space-time planes lands
// on just two texel rows):
//  * space-time planes (x,t) (y,t) (z,t): the t interpolation is the same for every point, so a workgroup sums
//    S[ix][ch] = sum gv * wx in an LDS line per plane and adds wt0*S / wt1*S to the two global rows once at the end;
//  * space planes: a half-wave walks a CONTIGUOUS chunk of the (spatially sorted) processing order and keeps, per
//    plane and corner, a pending texel row in registers; consecutive points that fall on the same texel are summed
//    there and reach memory as ONE 128-byte atomic row.
__device__ __forceinline__ int time_sample(float c, int size, int& i0, int& i1, float& w0, float& w1)
{
    float gm;
    const float v = unnorm_clip(c, size, gm);
    const int x0 = (int)floorf(v), x1 = x0 + 1;
    w0 = (float)x1 - v;
    w1 = v - (float)x0;
    i0 = (x0 >= 0 && x0 < size) ? x0 : -1;
    i1 = (x1 >= 0 && x1 < size) ? x1 : -1;
    return 0;
}

__global__ void __launch_bounds__(256)
hexplane_bwd_agg_kernel(HexArgs a, int chunk, const float* __restrict__ xyz, const float* __restrict__ dfeat, float* __restrict__ dxyz)
{
    extern __shared__ float s_line[];   // [3 space-time planes][Wmax][32]
    const int ch = threadIdx.x & 31;
    const int hw = (blockIdx.x * 256 + threadIdx.x) >> 5;          // half-wave id
    const int begin = hw * chunk, end = min(a.P, begin + chunk);
    for (int lvl = 0; lvl < a.levels; lvl++) {
        const int Wx = a.res[lvl][0], Wy = a.res[lvl][1], Wz = a.res[lvl][2], Wt = a.res[lvl][3];
        const int line_off[3] = {0, Wx * 32, (Wx + Wy) * 32};      // planes 2 (x,t), 4 (y,t), 5 (z,t)
        const int line_total = (Wx + Wy + Wz) * 32;
        __syncthreads();
        for (int i = threadIdx.x; i < line_total; i += 256) s_line[i] = 0.f;
        __syncthreads();
        // the shared t interpolation
        int t0, t1;
        float wt0, wt1;
        time_sample(a.time, Wt, t0, t1, wt0, wt1);
        // pending rows: 3 space planes x 4 corners
        int pid[3][4];
        float pacc[3][4];
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int c = 0; c < 4; c++) { pid[p][c] = -1; pacc[p][c] = 0.f; }
        float* gsp[3] = {a.grads[lvl][0], a.grads[lvl][1], a.grads[lvl][3]};
        for (int gi = begin; gi < end; gi++) {
            const int g = a.order ? (int)a.order[gi] : gi;
            float c[4];
            norm_coords(a, xyz, g, c);
            PlaneSample s[6];
            float v[6], t00[6], t01[6], t10[6], t11[6];
#pragma unroll
            for (int p = 0; p < 6; p++) {
                const int ca = kCombA[p], cb = kCombB[p];
                s[p] = make_sample(c[ca], c[cb], a.res[lvl][ca], a.res[lvl][cb]);
                const float* __restrict__ pl = a.planes[lvl][p];
                t00[p] = s[p].i00 >= 0 ? pl[(size_t)s[p].i00 * 32 + ch] : 0.f;
                t01[p] = s[p].i01 >= 0 ? pl[(size_t)s[p].i01 * 32 + ch] : 0.f;
                t10[p] = s[p].i10 >= 0 ? pl[(size_t)s[p].i10 * 32 + ch] : 0.f;
                t11[p] = s[p].i11 >= 0 ? pl[(size_t)s[p].i11 * 32 + ch] : 0.f;
                float acc = 0.f;
                acc += t00[p] * s[p].w00;
                acc += t01[p] * s[p].w01;
                acc += t10[p] * s[p].w10;
                acc += t11[p] * s[p].w11;
                v[p] = acc;
            }
            const float go = dfeat[(size_t)g * (a.levels * 32) + lvl * 32 + ch];
            float pre[7], suf[7];
            pre[0] = 1.f;
#pragma unroll
            for (int p = 0; p < 6; p++) pre[p + 1] = pre[p] * v[p];
            suf[6] = 1.f;
#pragma unroll
            for (int p = 5; p >= 0; p--) suf[p] = suf[p + 1] * v[p];
            float gc[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int p = 0; p < 6; p++) {
                const float gv = go * pre[p] * suf[p + 1];
                const int ca = kCombA[p], cb = kCombB[p];
                if (cb == 3) {
                    // space-time plane: W axis = space coordinate ca, H axis = t (shared)
                    const int li = p == 2 ? 0 : (p == 4 ? 1 : 2);
                    const float wx0 = (float)(s[p].ixn + 1) - s[p].ix, wx1 = s[p].ix - (float)s[p].ixn;
                    const int Wd = a.res[lvl][ca];
                    if (s[p].ixn >= 0 && s[p].ixn < Wd) atomicAdd(&s_line[line_off[li] + s[p].ixn * 32 + ch], gv * wx0);
                    if (s[p].ixn + 1 >= 0 && s[p].ixn + 1 < Wd) atomicAdd(&s_line[line_off[li] + (s[p].ixn + 1) * 32 + ch], gv * wx1);
                } else {
                    const int si = p == 0 ? 0 : (p == 1 ? 1 : 2);   // planes 0 (x,y), 1 (x,z), 3 (y,z)
                    const int ids[4] = {s[p].i00, s[p].i01, s[p].i10, s[p].i11};
                    const float ws[4] = {s[p].w00, s[p].w01, s[p].w10, s[p].w11};
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if (ids[k] == pid[si][k]) {
                            pacc[si][k] += gv * ws[k];
                        } else {
                            if (pid[si][k] >= 0) atomicAdd(&gsp[si][(size_t)pid[si][k] * 32 + ch], pacc[si][k]);
                            pid[si][k] = ids[k];
                            pacc[si][k] = gv * ws[k];
                        }
                    }
                }
                // grid gradient (same expressions as the generic kernel)
                const float x0 = (float)s[p].ixn, y0 = (float)s[p].iyn, x1 = x0 + 1.f, y1 = y0 + 1.f;
                float gix = 0.f, giy = 0.f;
                gix -= t00[p] * (y1 - s[p].iy) * gv;
                giy -= t00[p] * (x1 - s[p].ix) * gv;
                gix += t01[p] * (y1 - s[p].iy) * gv;
                giy -= t01[p] * (s[p].ix - x0) * gv;
                gix -= t10[p] * (s[p].iy - y0) * gv;
                giy += t10[p] * (x1 - s[p].ix) * gv;
                gix += t11[p] * (s[p].iy - y0) * gv;
                giy += t11[p] * (s[p].ix - x0) * gv;
                if (ca < 3) gc[ca] += gix * s[p].gx_mul;
                if (cb < 3) gc[cb] += giy * s[p].gy_mul;
            }
            if (dxyz) {
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const float tot = half_wave_sum(gc[k]) * (2.0f / (a.a1[k] - a.a0[k]));
                    if (ch == 0) atomicAdd(&dxyz[3 * g + k], tot);
                }
            }
        }
        // drain the pending rows
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int c = 0; c < 4; c++)
                if (pid[p][c] >= 0) atomicAdd(&gsp[p][(size_t)pid[p][c] * 32 + ch], pacc[p][c]);
        __syncthreads();
        // S lines -> the two global rows t0 / t1 of each space-time plane
        float* gtp[3] = {a.grads[lvl][2], a.grads[lvl][4], a.grads[lvl][5]};
        const int Wl[3] = {Wx, Wy, Wz};
#pragma unroll
        for (int li = 0; li < 3; li++) {
            for (int i = threadIdx.x; i < Wl[li] * 32; i += 256) {
                const float sv = s_line[line_off[li] + i];
                if (sv != 0.f) {
                    if (t0 >= 0) atomicAdd(&gtp[li][(size_t)t0 * Wl[li] * 32 + i], sv * wt0);
                    if (t1 >= 0) atomicAdd(&gtp[li][(size_t)t1 * Wl[li] * 32 + i], sv * wt1);
                }
            }
        }
    }
}


// =================================================================================================================
// v4 kernels: sample parameters are computed ONCE per (point, level) -- by one lane, in "phase A" of a 64-point
// chunk -- and parked in LDS as a 16-byte record per plane; "phase B" then walks the chunk with lane = channel and
// reads each point's records as LDS broadcasts.  This removes the 32-fold replication of the coordinate arithmetic
// of the kernels above (every channel lane redoes it there).  Workgroups specialise by level (blockIdx.y).
//
// Record of one (point, plane): {off00 | flags, bx, by, x0} with off00 = texel (y0, x0) in floats (a multiple of 32,
// so the low bits are free): bit0 = x0+1 is inside, bit1 = y0+1 is inside, bit2 / bit3 = the x / y coordinate was
// NOT clipped at the border (its gradient multiplier is (size-1)/2, else 0), bit4 = y0 is odd.  bx = ix - x0 and by = iy - y0 are
// exact; ax = 1 - bx, ay = 1 - by are bit-identical to ATen's (x0+1) - ix (Sterbenz), so the four weights are
// ATen's.  A corner that is outside gets weight exactly 0 and is redirected to the texel next to it.
// =================================================================================================================
#ifdef MOM_DBG_NOGATOM
#define GATOM(p, v) ((void)(v))
#else
#define GATOM(p, v) atomicAdd(p, v)
#endif
#ifdef MOM_DBG_NOLATOM
#define LATOM(p, v) ((void)(v))
#else
#define LATOM(p, v) atomicAdd(p, v)
#endif
#ifdef MOM_DBG_NOTEX
#define TEX(pl, o) (__int_as_float(o) * 1e-30f + 0.5f)
#else
#define TEX(pl, o) (pl)[o]
#endif
#ifdef MOM_DBG_NODFEAT
#define DFEAT(x) (1e-3f * ch)
#else
#define DFEAT(x) (x)
#endif
constexpr int kChunk4 = 64;

__device__ __forceinline__ float4 make_rec4(float cx, float cy, int Wd, int Hd)
{
    float gxm, gym;
    const float ix = unnorm_clip(cx, Wd, gxm), iy = unnorm_clip(cy, Hd, gym);
    const int x0 = (int)floorf(ix), y0 = (int)floorf(iy);
    int off = (y0 * Wd + x0) * 32;
    off |= (x0 + 1 < Wd) ? 1 : 0;
    off |= (y0 + 1 < Hd) ? 2 : 0;
    off |= (gxm != 0.f) ? 4 : 0;
    off |= (gym != 0.f) ? 8 : 0;
    off |= (y0 & 1) ? 16 : 0;          // parity of y0 (that of x0 is in .w): the backward's pending-row slots
    return make_float4(__int_as_float(off), ix - (float)x0, iy - (float)y0, __int_as_float(x0));
}

struct Corner4 {
    int o00, o01, o10, o11;
    float w00, w01, w10, w11, ax, bx, ay, by;
};
__device__ __forceinline__ Corner4 decode4(const float4 r, int Wd)
{
    Corner4 c;
    const int raw = __float_as_int(r.x);
    c.o00 = raw & ~31;
    c.o01 = c.o00 + ((raw & 1) ? 32 : 0);
    c.o10 = c.o00 + ((raw & 2) ? Wd * 32 : 0);
    c.o11 = c.o10 + ((raw & 1) ? 32 : 0);
    c.bx = r.y; c.by = r.z;
    c.ax = 1.f - c.bx; c.ay = 1.f - c.by;
    c.w00 = c.ax * c.ay; c.w01 = c.bx * c.ay; c.w10 = c.ax * c.by; c.w11 = c.bx * c.by;
    return c;
}

__global__ void __launch_bounds__(256)
hexplane_fwd4_kernel(HexArgs a, int nchunks, const float* __restrict__ xyz, float* __restrict__ feat)
{
    __shared__ float4 s_rec[4][kChunk4][6];
    const int lane = threadIdx.x & 63, ch = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6;
    const int lvl = blockIdx.y;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (gridDim.x * 256) >> 6;
    for (int chunk = wave; chunk < nchunks; chunk += nwaves) {
        const int gi = chunk * kChunk4 + lane;
        const int g_mine = gi < a.P ? (a.order ? (int)a.order[gi] : gi) : -1;
        __builtin_amdgcn_wave_barrier();
        if (g_mine >= 0) {
            float c[4];
            norm_coords(a, xyz, g_mine, c);
#pragma unroll
            for (int p = 0; p < 6; p++) s_rec[wv][lane][p] = make_rec4(c[kCombA[p]], c[kCombB[p]], a.res[lvl][kCombA[p]], a.res[lvl][kCombB[p]]);
        }
        __builtin_amdgcn_wave_barrier();
        const int npts = min(kChunk4, a.P - chunk * kChunk4);
        const int n_half = max(0, min(32, npts - 32 * h));     // this half walks points [32h, 32h + n_half)
        for (int i = 0; i < n_half; i++) {
            const int g = __shfl(g_mine, 32 * h + i);
            float prod = 1.f;
#pragma unroll
            for (int p = 0; p < 6; p++) {
                const Corner4 c = decode4(s_rec[wv][32 * h + i][p], a.res[lvl][kCombA[p]]);
                const float* __restrict__ pl = a.planes[lvl][p] + ch;
                float v = 0.f;
                v += pl[c.o00] * c.w00;
                v += pl[c.o01] * c.w01;
                v += pl[c.o10] * c.w10;
                v += pl[c.o11] * c.w11;
                prod = prod * v;
            }
            feat[(size_t)g * (a.levels * 32) + lvl * 32 + ch] = prod;
        }
    }
}

__global__ void __launch_bounds__(256)
hexplane_bwd4_kernel(HexArgs a, int chunks_per_wave, int nchunks, const float* __restrict__ xyz, const float* __restrict__ dfeat,
                     float* __restrict__ dxyz)
{
    extern __shared__ float s_dyn[];                   // [4][64][6] float4 records | [3 planes][W][32] lines of this level
    float4* s_recs = reinterpret_cast<float4*>(s_dyn);
    float* s_line = s_dyn + 4 * kChunk4 * 6 * 4;
    const int lane = threadIdx.x & 63, ch = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6;
    const int lvl = blockIdx.y;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int Wx = a.res[lvl][0], Wy = a.res[lvl][1], Wz = a.res[lvl][2], Wt = a.res[lvl][3];
    const int line_off[3] = {0, Wx * 32, (Wx + Wy) * 32};
    const int line_total = (Wx + Wy + Wz) * 32;
    for (int i = threadIdx.x; i < line_total; i += 256) s_line[i] = 0.f;
    __syncthreads();

    int pid[3][4];
    float pacc[3][4];
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
        for (int c = 0; c < 4; c++) { pid[p][c] = -1; pacc[p][c] = 0.f; }
    int lpid[3][2];
    float lacc[3][2];
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
        for (int c = 0; c < 2; c++) { lpid[p][c] = -1; lacc[p][c] = 0.f; }
    float* gsp[3] = {a.grads[lvl][0], a.grads[lvl][1], a.grads[lvl][3]};
    float gmul[4];                                      // d(ix)/d(normalised coord) when not clipped
#pragma unroll
    for (int k = 0; k < 4; k++) gmul[k] = (float)(a.res[lvl][k] - 1) / 2.f;

    const int c_begin = wave * chunks_per_wave, c_end = min(nchunks, c_begin + chunks_per_wave);
    for (int chunk = c_begin; chunk < c_end; chunk++) {
        const int gi = chunk * kChunk4 + lane;
        const int g_mine = gi < a.P ? (a.order ? (int)a.order[gi] : gi) : -1;
        float4* wrec = s_recs + wv * kChunk4 * 6;
        __builtin_amdgcn_wave_barrier();
        if (g_mine >= 0) {
            float c[4];
            norm_coords(a, xyz, g_mine, c);
#pragma unroll
            for (int p = 0; p < 6; p++) wrec[lane * 6 + p] = make_rec4(c[kCombA[p]], c[kCombB[p]], a.res[lvl][kCombA[p]], a.res[lvl][kCombB[p]]);
        }
        __builtin_amdgcn_wave_barrier();
        const int npts = min(kChunk4, a.P - chunk * kChunk4);
        const int n_half = max(0, min(32, npts - 32 * h));
        // software pipeline: records + the 24 texel loads + dfeat of point i+1 are in flight while point i is processed
        float4 nr[6], r[6];
        float n00[6], n01[6], n10[6], n11[6], t00[6], t01[6], t10[6], t11[6];
        float go = 0.f, ngo = 0.f;
        float dx_mine[3] = {0.f, 0.f, 0.f};
        int ng = 0;
        auto fetch = [&](int i) {
            ng = __shfl(g_mine, 32 * h + i);
#pragma unroll
            for (int p = 0; p < 6; p++) {
                nr[p] = wrec[(32 * h + i) * 6 + p];
                const int raw = __float_as_int(nr[p].x);
                const int o00 = raw & ~31, sx = (raw & 1) ? 32 : 0, sy = (raw & 2) ? a.res[lvl][kCombA[p]] * 32 : 0;
                const float* __restrict__ pl = a.planes[lvl][p] + ch;
                n00[p] = TEX(pl, o00);
                n01[p] = TEX(pl, o00 + sx);
                n10[p] = TEX(pl, o00 + sy);
                n11[p] = TEX(pl, o00 + sy + sx);
            }
            ngo = DFEAT(dfeat[(size_t)ng * (a.levels * 32) + lvl * 32 + ch]);
        };
        if (n_half > 0) fetch(0);
        for (int i = 0; i < n_half; i++) {
            go = ngo;
#pragma unroll
            for (int p = 0; p < 6; p++) { r[p] = nr[p]; t00[p] = n00[p]; t01[p] = n01[p]; t10[p] = n10[p]; t11[p] = n11[p]; }
            if (i + 1 < n_half) fetch(i + 1);
            Corner4 c[6];
            float v[6];
#pragma unroll
            for (int p = 0; p < 6; p++) {
                c[p] = decode4(r[p], a.res[lvl][kCombA[p]]);
                float acc = 0.f;
                acc += t00[p] * c[p].w00;
                acc += t01[p] * c[p].w01;
                acc += t10[p] * c[p].w10;
                acc += t11[p] * c[p].w11;
                v[p] = acc;
            }
            float pre[7], suf[7];
            pre[0] = 1.f;
#pragma unroll
            for (int p = 0; p < 6; p++) pre[p + 1] = pre[p] * v[p];
            suf[6] = 1.f;
#pragma unroll
            for (int p = 5; p >= 0; p--) suf[p] = suf[p + 1] * v[p];
            float gc[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int p = 0; p < 6; p++) {
                const float gv = go * pre[p] * suf[p + 1];
                const int ca = kCombA[p], cb = kCombB[p];
                const int raw = __float_as_int(r[p].x);
                const bool has_x1 = raw & 1, has_y1 = raw & 2;
                if (cb == 3) {
                    // space-time plane: sum gv * (x weights) into the level's LDS line; the t weights are applied at the end
                    // consecutive points (Morton order) mostly share x0: keep the two line rows in registers and
                    // touch LDS only when the row changes
                    const int li = p == 2 ? 0 : (p == 4 ? 1 : 2);
                    const int x0 = __float_as_int(r[p].w);
                    // a pending row sits in the slot of its PARITY, so it keeps its slot when the walk moves to the next column
                    // (rows x0 and x0 + 1 always have different parities): slot k takes corner k ^ (x0 & 1)
                    const bool odd_x = x0 & 1;
                    const int id_a = x0, id_b = has_x1 ? x0 + 1 : -2;
                    const float w_a = gv * c[p].ax, w_b = gv * c[p].bx;
                    const int lid[2] = {odd_x ? id_b : id_a, odd_x ? id_a : id_b};
                    const float lw[2] = {odd_x ? w_b : w_a, odd_x ? w_a : w_b};
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        if (lid[k] == lpid[li][k]) {
                            lacc[li][k] += lw[k];
                        } else if (lid[k] >= 0) {
                            if (lpid[li][k] >= 0) LATOM(&s_line[line_off[li] + lpid[li][k] * 32 + ch], lacc[li][k]);
                            lpid[li][k] = lid[k];
                            lacc[li][k] = lw[k];
                        }
                    }
                } else {
                    const int si = p == 0 ? 0 : (p == 1 ? 1 : 2);
                    // corners outside the plane carry weight 0: give them id -2 so that they never start a row
                    // A pending row sits in the slot (parity of its y, parity of its x).  The four corners of a texel always take
                    // four different slots, and a row keeps its slot when the walk moves to a neighbouring texel, so e.g. after
                    // x0 -> x0 + 1 the two rows of column x0 + 1 stay pending instead of being flushed and restarted: exactly the
                    // flush count of a 4-way associative set (tools/sim_hexplane_runs.py: 40 % fewer rows than one slot per
                    // corner).  Slot k takes corner k ^ s, s = 2 (y0 & 1) + (x0 & 1): two conditional swap stages.
                    int ids[4] = {c[p].o00, has_x1 ? c[p].o01 : -2, has_y1 ? c[p].o10 : -2, (has_x1 && has_y1) ? c[p].o11 : -2};
                    float ws[4] = {c[p].w00, c[p].w01, c[p].w10, c[p].w11};
                    {
                        const bool sx1 = __float_as_int(r[p].w) & 1, sy1 = raw & 16;
                        const int i0 = sx1 ? ids[1] : ids[0], i1 = sx1 ? ids[0] : ids[1], i2 = sx1 ? ids[3] : ids[2], i3 = sx1 ? ids[2] : ids[3];
                        const float f0 = sx1 ? ws[1] : ws[0], f1 = sx1 ? ws[0] : ws[1], f2 = sx1 ? ws[3] : ws[2], f3 = sx1 ? ws[2] : ws[3];
                        ids[0] = sy1 ? i2 : i0; ids[1] = sy1 ? i3 : i1; ids[2] = sy1 ? i0 : i2; ids[3] = sy1 ? i1 : i3;
                        ws[0] = sy1 ? f2 : f0; ws[1] = sy1 ? f3 : f1; ws[2] = sy1 ? f0 : f2; ws[3] = sy1 ? f1 : f3;
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if (ids[k] == pid[si][k]) {
                            pacc[si][k] += gv * ws[k];
                        } else if (ids[k] >= 0) {
                            if (pid[si][k] >= 0) GATOM(&gsp[si][(size_t)pid[si][k] + ch], pacc[si][k]);
                            pid[si][k] = ids[k];
                            pacc[si][k] = gv * ws[k];
                        }
                    }
                }
                // grid gradient (ATen grid_sampler_2d backward); out-of-plane corners have value 0 there
                const float v01 = has_x1 ? t01[p] : 0.f, v10 = has_y1 ? t10[p] : 0.f, v11 = (has_x1 && has_y1) ? t11[p] : 0.f;
                float gix = 0.f, giy = 0.f;
                gix -= t00[p] * c[p].ay * gv;
                giy -= t00[p] * c[p].ax * gv;
                gix += v01 * c[p].ay * gv;
                giy -= v01 * c[p].bx * gv;
                gix -= v10 * c[p].by * gv;
                giy += v10 * c[p].ax * gv;
                gix += v11 * c[p].by * gv;
                giy += v11 * c[p].bx * gv;
                if (ca < 3) gc[ca] += gix * ((raw & 4) ? gmul[ca] : 0.f);
                if (cb < 3) gc[cb] += giy * ((raw & 8) ? gmul[cb] : 0.f);
            }
            if (dxyz) {
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const float tot = half_wave_sum(gc[k]) * (2.0f / (a.a1[k] - a.a0[k]));
                    if (ch == i) dx_mine[k] = tot;                       // lane i of the half keeps point i's total
                }
            }
        }
        // lane (32h + i) holds point (32h + i) == its own phase-A point: one wave-wide add per component
        if (dxyz && g_mine >= 0) {
#pragma unroll
            for (int k = 0; k < 3; k++) GATOM(&dxyz[3 * g_mine + k], dx_mine[k]);   // the other level adds its share too
        }
    }
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
        for (int c = 0; c < 4; c++)
            if (pid[p][c] >= 0) GATOM(&gsp[p][(size_t)pid[p][c] + ch], pacc[p][c]);
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
        for (int c = 0; c < 2; c++)
            if (lpid[p][c] >= 0) LATOM(&s_line[line_off[p] + lpid[p][c] * 32 + ch], lacc[p][c]);
    __syncthreads();
    // S lines -> the two global rows t0 / t1 of each space-time plane
    int t0, t1;
    float wt0, wt1;
    time_sample(a.time, Wt, t0, t1, wt0, wt1);
    float* gtp[3] = {a.grads[lvl][2], a.grads[lvl][4], a.grads[lvl][5]};
    const int Wl[3] = {Wx, Wy, Wz};
#pragma unroll
    for (int li = 0; li < 3; li++)
        for (int i = threadIdx.x; i < Wl[li] * 32; i += 256) {
            const float sv = s_line[line_off[li] + i];
            if (sv != 0.f) {
                if (t0 >= 0) GATOM(&gtp[li][(size_t)t0 * Wl[li] * 32 + i], sv * wt0);
                if (t1 >= 0) GATOM(&gtp[li][(size_t)t1 * Wl[li] * 32 + i], sv * wt1);
            }
        }
}

}  // namespace

extern "C" int mom_hexplane_forward(const MomHexPlane* hp, int P, const float* xyz, const float* times, float time,
                                    const uint32_t* order, float* feat, mom_stream_t stream)
{
    if (!hp || hp->channels != 32 || hp->levels < 1 || hp->levels > 4 || P < 0) return MOM_EINVAL;
    if (P == 0) return MOM_OK;
    if (!xyz || !feat) return MOM_EINVAL;
    HexArgs a;
    a.P = P; a.levels = hp->levels; a.time = time; a.times = times; a.order = order;
    for (int l = 0; l < 4; l++)
        for (int k = 0; k < 4; k++) a.res[l][k] = hp->res[l][k];
    for (int l = 0; l < 4; l++)
        for (int p = 0; p < 6; p++) { a.planes[l][p] = hp->planes[l][p]; a.grads[l][p] = nullptr; }
    for (int k = 0; k < 3; k++) { a.a0[k] = hp->aabb[k]; a.a1[k] = hp->aabb[3 + k]; }
    const long long units = (long long)P * hp->levels;
    MomProfScope ps(MOM_P_HEX_FWD, (hipStream_t)stream);
    if (!getenv("MOM_HEX_V1")) {
        const int nchunks = (P + kChunk4 - 1) / kChunk4;
        int blocks = (nchunks + 3) / 4;
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(hexplane_fwd4_kernel, dim3(blocks, hp->levels), dim3(256), 0, (hipStream_t)stream, a, nchunks, xyz, feat);
        return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
    }
    hipLaunchKernelGGL(hexplane_fwd_kernel, dim3((unsigned)((units + 7) / 8)), dim3(256), 0, (hipStream_t)stream, a, xyz, feat);
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}

extern "C" int mom_hexplane_backward(const MomHexPlane* hp, int P, const float* xyz, const float* times, float time,
                                     const uint32_t* order, const float* dfeat, float* dxyz, mom_stream_t stream)
{
    if (!hp || hp->channels != 32 || hp->levels < 1 || hp->levels > 4 || P < 0) return MOM_EINVAL;
    if (P == 0) return MOM_OK;
    if (!xyz || !dfeat) return MOM_EINVAL;
    HexArgs a;
    a.P = P; a.levels = hp->levels; a.time = time; a.times = times; a.order = order;
    for (int l = 0; l < 4; l++)
        for (int k = 0; k < 4; k++) a.res[l][k] = hp->res[l][k];
    for (int l = 0; l < 4; l++)
        for (int p = 0; p < 6; p++) {
            a.planes[l][p] = hp->planes[l][p];
            a.grads[l][p] = hp->grads[l][p];
            if (l < hp->levels && (!a.planes[l][p] || !a.grads[l][p])) return MOM_EINVAL;
        }
    for (int k = 0; k < 3; k++) { a.a0[k] = hp->aabb[k]; a.a1[k] = hp->aabb[3 + k]; }
    const long long units = (long long)P * hp->levels;
    MomProfScope ps(MOM_P_HEX_BWD, (hipStream_t)stream);
    // aggregated path: one shared timestamp and space-time lines that fit in LDS
    int wmax = 0;
    for (int l = 0; l < hp->levels; l++) {
        const int w = hp->res[l][0] + hp->res[l][1] + hp->res[l][2];
        if (w > wmax) wmax = w;
    }
    const size_t lds_bytes = (size_t)wmax * 32 * sizeof(float);
    const size_t lds4 = sizeof(float) * ((size_t)4 * kChunk4 * 6 * 4 + (size_t)wmax * 32);
    if (!times && lds4 <= 160 * 1024 && !getenv("MOM_HEX_V2")) {
        static bool attr_set = false;
        if (!attr_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(hexplane_bwd4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024) != hipSuccess)
                return MOM_ELAUNCH;
            attr_set = true;
        }
        const int nchunks = (P + kChunk4 - 1) / kChunk4;
        static int blocks4 = 0;
        if (!blocks4) {
            const char* e = getenv("MOM_HEX_BLOCKS");
            blocks4 = e ? atoi(e) : 768;
        }
        const int waves = blocks4 * 4;
        const int cpw = (nchunks + waves - 1) / waves;     // contiguous chunks per wave (run-length aggregation)
        hipLaunchKernelGGL(hexplane_bwd4_kernel, dim3(blocks4, hp->levels), dim3(256), lds4, (hipStream_t)stream, a, cpw, nchunks, xyz,
                           dfeat, dxyz);
        return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
    }
    if (!times && lds_bytes <= 64 * 1024) {
        static int blocks = 0;                        // persistent half-waves walking contiguous chunks of the order
        if (!blocks) {
            const char* e = getenv("MOM_HEX_BLOCKS");
            blocks = e ? atoi(e) : 512;
        }
        const int chunk = (P + blocks * 8 - 1) / (blocks * 8);
        hipLaunchKernelGGL(hexplane_bwd_agg_kernel, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, a, chunk, xyz, dfeat, dxyz);
        return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
    }
    hipLaunchKernelGGL(hexplane_bwd_kernel, dim3((unsigned)((units + 7) / 8)), dim3(256), 0, (hipStream_t)stream, a, xyz, dfeat, dxyz);
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}
