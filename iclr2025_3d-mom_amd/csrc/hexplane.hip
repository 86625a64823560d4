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
#include "hexplane_dev.h"
#include <stdlib.h>

namespace {

struct PlaneSample {
    int i00, i01, i10, i11;      // texel indices (row-major over [H][W]), -1 when out of bounds
    float w00, w01, w10, w11;    // nw, ne, sw, se
    float gx_mul, gy_mul;        // d(ix)/d(coord) incl. border-clip mask
    float ix, iy;
    int ixn, iyn;
};

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


// =================================================================================================================
// Chunked kernels: sample parameters are computed ONCE per (point, level) -- by one lane, in "phase A" of a chunk of
// points -- and parked in LDS as small records; "phase B" then walks the chunk with lane = channel and reads each
// point's records as LDS broadcasts.  This removes the 32-fold replication of the coordinate arithmetic of the
// generic kernels above (every channel lane redoes it there).  Workgroups specialise by level (blockIdx.y).
//
// Record of one (point, plane): {off00 | flags, bx, by, aux} with off00 = texel (y0, x0) in floats (a multiple of 32,
// so the low bits are free): bit0 = x0+1 is inside, bit1 = y0+1 is inside, bit2 / bit3 = the x / y coordinate was
// NOT clipped at the border (its gradient multiplier is (size-1)/2, else 0), bit4 = y0 is odd.  bx = ix - x0 and by = iy - y0 are
// exact; ax = 1 - bx, ay = 1 - by are bit-identical to ATen's (x0+1) - ix (Sterbenz), so the four weights are
// ATen's.  A corner that is outside gets weight exactly 0 and is redirected to the texel next to it.
// =================================================================================================================

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

__device__ __forceinline__ float ld_f32(const float* __restrict__ base, unsigned byte_off)
{
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);     // uniform base + 32-bit lane offset
}

// Forward record of one (point, plane), ready to use: the four corners' byte offsets and their four weights.  Decoding the
// compact record above cost every channel lane a dozen instructions per (point, plane) -- half of the forward kernel's
// instruction stream (25.5 M wave instructions per launch on 76 % of the issue cycles); it is done once per point here.
__device__ __forceinline__ void make_rec_fwd(float cx, float cy, int Wd, int Hd, uint4& O, float4& Wt)
{
    const Corner4 c = decode4(make_rec4(cx, cy, Wd, Hd), Wd);
    O = make_uint4((unsigned)c.o00 * 4u, (unsigned)c.o01 * 4u, (unsigned)c.o10 * 4u, (unsigned)c.o11 * 4u);
    Wt = make_float4(c.w00, c.w01, c.w10, c.w11);
}
constexpr int kChunkF = 32;            // points per wave and chunk in the forward: each half-wave walks 16

__global__ void __launch_bounds__(256)
hexplane_fwd4_kernel(HexArgs a, int nchunks, const float* __restrict__ xyz, float* __restrict__ feat)
{
    __shared__ uint4 s_off[4][kChunkF][6];
    __shared__ float4 s_wt[4][kChunkF][6];
    const int lane = threadIdx.x & 63, ch = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6;
    const int lvl = blockIdx.y;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (gridDim.x * 256) >> 6;
    for (int chunk = wave; chunk < nchunks; chunk += nwaves) {
        const int gi = chunk * kChunkF + lane;
        const int g_mine = (lane < kChunkF && gi < a.P) ? (a.order ? (int)a.order[gi] : gi) : -1;
        __builtin_amdgcn_wave_barrier();
        if (g_mine >= 0) {
            float c[4];
            norm_coords(a, xyz, g_mine, c);
#pragma unroll
            for (int p = 0; p < 6; p++)
                make_rec_fwd(c[kCombA[p]], c[kCombB[p]], a.res[lvl][kCombA[p]], a.res[lvl][kCombB[p]], s_off[wv][lane][p], s_wt[wv][lane][p]);
        }
        __builtin_amdgcn_wave_barrier();
        const int npts = min(kChunkF, a.P - chunk * kChunkF);
        const int n_half = max(0, min(kChunkF / 2, npts - (kChunkF / 2) * h));     // this half walks points [16h, 16h + n_half)
        for (int i = 0; i < n_half; i++) {
            const int pt = (kChunkF / 2) * h + i;
            const int g = __shfl(g_mine, pt);
            float prod = 1.f;
#pragma unroll
            for (int p = 0; p < 6; p++) {
                const uint4 o = s_off[wv][pt][p];
                const float4 w = s_wt[wv][pt][p];
                const float* __restrict__ pl = a.planes[lvl][p] + ch;
                float v = 0.f;
                v += ld_f32(pl, o.x) * w.x;
                v += ld_f32(pl, o.y) * w.y;
                v += ld_f32(pl, o.z) * w.z;
                v += ld_f32(pl, o.w) * w.w;
                prod = prod * v;
            }
            feat[(size_t)g * (a.levels * 32) + lvl * 32 + ch] = prod;
        }
    }
}

// =================================================================================================================
// Backward, one timestamp for all points (the render() case), in TWO passes.
//
// The gradient of a plane texel is a sum over the points around it; the reference (autograd of grid_sample) issues one
// scatter-add per (point, corner, channel).  Here:
//
//  pass 1 (GATHER, points in 3-D Morton order, one half-wave per point, lane = channel): per (point, level) the six plane
//    samples, their product, and for each plane gv = dfeat * (product of the other five).  Every gv row (128 B) is STORED
//    -- no atomics, no LDS accumulation in this pass -- at the point's position in the order of the space plane it will be
//    scattered with.  The position gradient (ATen's grid_sampler_2d backward, regrouped) is reduced over the 32 channels
//    and added to dxyz.
//  pass 2 (SCATTER, one launch over the three space planes; points in that plane's own order = sorted by the finest
//    level's texel cell, 2-D Morton over cells): streams the gv rows sequentially and keeps four pending texel rows per
//    (plane, level) in registers -- slot = (parity of y, parity of x): a 4-way set that keeps a row while the walk moves to
//    a neighbouring cell -- issuing one 128-byte row of float atomics only when a row leaves its slot.  All points of a
//    cell are consecutive, so a row is flushed once per cell visit.  Each space plane carries ONE space-time plane that
//    shares an axis with it -- (x,y) carries (x,t), (x,z) carries (z,t), (y,z) carries (y,t): in the space plane's order the
//    shared coordinate's cell changes as rarely as the cell itself, so the space-time plane's line S[x][ch] = sum gv * wx
//    (the two time rows are the same for every point) accumulates in two pending rows, spills into a per-workgroup LDS line
//    when a row changes, and is drained with the two time weights at the end of the workgroup.
//
// Both passes compute the per-point sample parameters once, by one lane, into LDS records (phase A) and walk the chunk with
// lane = channel (phase B); everything uniform over the channels (byte offsets, fractions, gradient multipliers, slot
// permutation) is done in phase A, so phase B is loads, a dozen FMAs per plane and the stores.
// =================================================================================================================
constexpr int kChunk5 = 32;            // points per wave and chunk in pass 1: each half-wave walks 16

// order slot (0: (x,y), 1: (x,z), 2: (y,z)) whose sorted position a plane's gv row is stored at
__device__ __forceinline__ constexpr int order_slot_of_plane(int p) { return p == 0 ? 0 : (p == 1 ? 1 : (p == 2 ? 0 : (p == 3 ? 2 : (p == 4 ? 2 : 1)))); }

// pass-1 record of one (point, plane): R1 = {byte offset of texel (y0, x0), byte step to x0+1 (0 if outside), byte step to
// y0+1 (0 if outside), byte offset of the point's gv row}; R2 = {bx, by, gx, gy}: the fractions and d(ix)/d(world coordinate)
// (0 when the coordinate was clipped at the border, and for the time axis)
__device__ __forceinline__ void make_rec5(float cx, float cy, int Wd, int Hd, unsigned row_off, float gsx, float gsy, uint4& R1, float4& R2)
{
    float gxm, gym;
    const float ix = unnorm_clip(cx, Wd, gxm), iy = unnorm_clip(cy, Hd, gym);
    const int x0 = (int)floorf(ix), y0 = (int)floorf(iy);
    R1 = make_uint4((unsigned)(y0 * Wd + x0) * 128u, (x0 + 1 < Wd) ? 128u : 0u, (y0 + 1 < Hd) ? (unsigned)Wd * 128u : 0u, row_off);
    R2 = make_float4(ix - (float)x0, iy - (float)y0, gxm != 0.f ? gsx : 0.f, gym != 0.f ? gsy : 0.f);
}


#ifndef HX_GATHER_WAVES
#define HX_GATHER_WAVES 4
#endif
__global__ void __launch_bounds__(256, HX_GATHER_WAVES)
hexplane_bwd5_gather_kernel(HexArgs a, int nchunks, const float* __restrict__ xyz, const float* __restrict__ dfeat,
                            float* __restrict__ dxyz, const uint32_t* __restrict__ inv /* [3][P] */,
                            float* __restrict__ gvbuf /* [3 slots][levels][P][2][32] */)
{
    __shared__ uint4 s_r1[4][kChunk5][6];
    __shared__ float4 s_r2[4][kChunk5][6];
    const int lane = threadIdx.x & 63, ch = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6;
    const int lvl = blockIdx.y;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (gridDim.x * 256) >> 6;
    const unsigned chb = (unsigned)ch * 4u;
    const size_t plane_floats = (size_t)a.P * 32;                    // one (plane, level) buffer of gv rows
    // d(ix)/d(world coordinate) per axis when the coordinate is not clipped: (size-1)/2 * 2/(aabb1 - aabb0)
    float gscale[4];
#pragma unroll
    for (int k = 0; k < 3; k++) gscale[k] = ((float)(a.res[lvl][k] - 1) / 2.f) * (2.0f / (a.a1[k] - a.a0[k]));
    gscale[3] = 0.f;

    for (int chunk = wave; chunk < nchunks; chunk += nwaves) {
        // phase A: lane j < 32 prepares planes 0..2 of point j, lane 32 + j planes 3..5 of the same point
        const int gi = chunk * kChunk5 + ch;
        const int g_mine = gi < a.P ? (a.order ? (int)a.order[gi] : gi) : -1;
        __builtin_amdgcn_wave_barrier();
        if (g_mine >= 0) {
            float c[4];
            norm_coords(a, xyz, g_mine, c);
            unsigned pos[3];
#pragma unroll
            for (int k = 0; k < 3; k++) pos[k] = inv[((size_t)k * a.levels + lvl) * a.P + g_mine] * 256u;   // a slot's two rows are adjacent
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const int p0 = q, p1 = 3 + q;            // h == 0: (x,y) (x,z) (x,t); h == 1: (y,z) (y,t) (z,t)
                const int ca = h ? kCombA[p1] : kCombA[p0], cb = h ? kCombB[p1] : kCombB[p0];
                const int slot = h ? order_slot_of_plane(p1) : order_slot_of_plane(p0);
                uint4 R1; float4 R2;
                make_rec5(c[ca], c[cb], a.res[lvl][ca], a.res[lvl][cb], pos[slot], gscale[ca], gscale[cb], R1, R2);
                s_r1[wv][ch][3 * h + q] = R1;
                s_r2[wv][ch][3 * h + q] = R2;
            }
        }
        __builtin_amdgcn_wave_barrier();
        const int npts = min(kChunk5, a.P - chunk * kChunk5);
        const int n_half = max(0, min(16, npts - 16 * h));     // this half walks points [16h, 16h + n_half)
        float t00[6], t01[6], t10[6], t11[6];
        float go = 0.f;
        float dx_mine[3] = {0.f, 0.f, 0.f};
        auto fetch = [&](int i) {
            const int ng = __shfl(g_mine, 16 * h + i);
#pragma unroll
            for (int p = 0; p < 6; p++) {
                const uint4 R1 = s_r1[wv][16 * h + i][p];
                const float* __restrict__ pl = a.planes[lvl][p];
                const unsigned o = R1.x + chb;
                t00[p] = ld_f32(pl, o);
                t01[p] = ld_f32(pl, o + R1.y);
                t10[p] = ld_f32(pl, o + R1.z);
                t11[p] = ld_f32(pl, o + R1.y + R1.z);
            }
            go = dfeat[(size_t)ng * (a.levels * 32) + lvl * 32 + ch];
        };
        if (n_half > 0) fetch(0);
        for (int i = 0; i < n_half; i++) {
            // first half of the iteration: consume the texels (bilinear sample and the two raw position derivatives per plane);
            // after it the 24 texel registers are dead and the next point's loads can land in them while the second half runs
            float v[6], dgx[6], dgy[6];
#pragma unroll
            for (int p = 0; p < 6; p++) {
                const float4 R2 = s_r2[wv][16 * h + i][p];
                const float d0 = t01[p] - t00[p], d1 = t11[p] - t10[p];
                const float tx0 = __builtin_fmaf(R2.x, d0, t00[p]), tx1 = __builtin_fmaf(R2.x, d1, t10[p]);
                const float dy = tx1 - tx0;                          // = ax (t10 - t00) + bx (t11 - t01): d sample / d iy
                v[p] = __builtin_fmaf(R2.y, dy, tx0);
                dgx[p] = __builtin_fmaf(R2.y, d1 - d0, d0) * R2.z;   // (ay (t01 - t00) + by (t11 - t10)) * d ix / d coord
                dgy[p] = dy * R2.w;
            }
            const float gcur = go;
            const int icur = i;
            if (i + 1 < n_half) fetch(i + 1);
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
                const float gv = gcur * (pre[p] * suf[p + 1]);
                const int ca = kCombA[p], cb = kCombB[p];
                // [slot][level][position][space row | time row][32]: the two rows a scatter pass reads for one position are one 256-byte piece
                float* __restrict__ dst = gvbuf + ((size_t)order_slot_of_plane(p) * a.levels + lvl) * 2 * plane_floats + ((p == 2 || p == 4 || p == 5) ? 32 : 0);          // uniform
                *reinterpret_cast<float*>(reinterpret_cast<char*>(dst) + (s_r1[wv][16 * h + icur][p].w + chb)) = gv;
                gc[ca] = __builtin_fmaf(gv, dgx[p], gc[ca]);
                if (cb < 3) gc[cb] = __builtin_fmaf(gv, dgy[p], gc[cb]);
            }
            if (dxyz) {
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const float tot = half_wave_sum(gc[k]);
                    if (ch == icur) dx_mine[k] = tot;                    // lane i of the half keeps point 16h + i's total
                }
            }
        }
        if (dxyz) {
            const int gp = __shfl(g_mine, 16 * h + (ch & 15));           // lanes 0..15 of each half: point 16h + ch
            if (ch < n_half && gp >= 0) {
#pragma unroll
                for (int k = 0; k < 3; k++) atomicAdd(&dxyz[3 * gp + k], dx_mine[k]);   // the other levels add their share too
            }
        }
    }
}

// pass 2: blockIdx.y = order slot (0: (x,y) + (x,t), 1: (x,z) + (z,t), 2: (y,z) + (y,t)); blockIdx.z = level.  Every level has
// its own orders (its cells do not nest in the finer level's: align_corners scales by size - 1).  Each HALF-wave walks its own
// contiguous range of sorted positions.
constexpr int kChunk5s = 32;           // sorted positions per half-wave and chunk
constexpr int kBatch5s = 16;           // gv rows in flight per lane and plane
constexpr int kRec5s = 4 + 4 + 4 + 2;  // dwords of one pass-2 record: int4 ids | float4 ws | float4 {lw0, lw1, flag, -} | int2 ride ids

// CROWS: the gather left ONE row per (order slot, position) -- c = dfeat * (product of the four planes outside the slot), see
// hexplane_bwd6_gather_kernel -- instead of the slot's two gv rows; this pass forms them itself: gv(space plane) = c * (time line's
// sample), from the frame's line values staged in LDS, and gv(time plane) = c * (space plane's sample), from the four texel rows
// it keeps pending anyway (their VALUES ride along with the pending gradient rows and are fetched when a slot takes a new row).
template <bool CROWS>
__global__ void __launch_bounds__(256)
hexplane_bwd5_scatter_kernel(HexArgs a, int per_half, const float* __restrict__ xyz,
                             const uint32_t* __restrict__ order /* [3][levels][P] */, const float* __restrict__ gvbuf,
                             const float* __restrict__ lines, LineTab lt)
{
    extern __shared__ float s_dyn5[];                  // [4 waves] x (int4 ids[64] | float4 ws[64] | float4 rd[64] | int2 rid[64]) | line [W][32] (| line values [W][32])
    constexpr int kPer = 64 * kRec5s;                  // dwords per wave
    float* s_line = s_dyn5 + 4 * kPer;
    const int lane = threadIdx.x & 63, ch = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6;
    const int si = blockIdx.y, lvl = blockIdx.z;
    const int p = si == 0 ? 0 : (si == 1 ? 1 : 3), ca = si == 2 ? 1 : 0, cb = si == 0 ? 1 : 2;
    const int pt = si == 0 ? 2 : (si == 1 ? 5 : 4);    // the space-time plane carried along: shares axis `cs` with this plane
    const bool ride_on_b = si == 1;                     // (x,z) carries (z,t): the shared axis is this plane's second one
    const int cs = ride_on_b ? cb : ca;
    const int Wd = a.res[lvl][ca], Hd = a.res[lvl][cb], Ws = a.res[lvl][cs];
    float* __restrict__ gp = a.grads[lvl][p] + ch;
    const size_t plane_floats = (size_t)a.P * 32;
    // 6-row form: [slot][level][position][space row | time row][32]; CROWS: [slot][level][position][32]
    const float* __restrict__ src_s = gvbuf + ((size_t)si * a.levels + lvl) * (CROWS ? 1 : 2) * plane_floats + ch;
    const float* __restrict__ src_t = src_s + 32;
    constexpr int kRowF = CROWS ? 32 : 64;             // floats between consecutive positions
    const float* __restrict__ plane_v = a.planes[lvl][p] + ch;
    const uint32_t* __restrict__ ord = order + ((size_t)si * a.levels + lvl) * a.P;
    const float lo_a = a.a0[ca], sc_a = 2.0f / (a.a1[ca] - a.a0[ca]), lo_b = a.a0[cb], sc_b = 2.0f / (a.a1[cb] - a.a0[cb]);
    float* __restrict__ my_line = s_line + ch;
    // CROWS: the carried time plane's line at this frame's timestamp, read from the table itself when a line row changes (rare in
    // this order; the 73 KB table lives in the L1 / L2 -- a copy in LDS cost the kernel two of its five workgroups per CU)
    const float* __restrict__ my_lval = CROWS ? lines + lt.off[lvl][cs] + ch : nullptr;
    for (int i = threadIdx.x; i < Ws * 32; i += 256) s_line[i] = 0.f;
    __syncthreads();
    float tv[4] = {0.f, 0.f, 0.f, 0.f}, lv[2] = {0.f, 0.f};   // CROWS: the VALUES of the pending texel rows / line rows

    int pid[4] = {-1, -1, -1, -1};
    float pacc[4] = {0.f, 0.f, 0.f, 0.f};
    int lpid[2] = {-1, -1};
    float lacc[2] = {0.f, 0.f};
    int prev_cell = -1, prev_row = -1;                  // of the position before this chunk (uniform per half-wave)
    // this half-wave's contiguous range of sorted positions
    const long long hw = ((long long)blockIdx.x * 4 + wv) * 2 + h;
    const long long r_begin = hw * per_half;
    const int r_end = (int)(r_begin + per_half < (long long)a.P ? r_begin + per_half : (long long)a.P);
    float* rec = s_dyn5 + wv * kPer;
    int4* w_ids = reinterpret_cast<int4*>(rec);
    float4* w_ws = reinterpret_cast<float4*>(rec + 4 * 64);
    float4* w_rd = reinterpret_cast<float4*>(rec + 8 * 64);
    int2* w_rid = reinterpret_cast<int2*>(rec + 12 * 64);
    for (long long base_ll = r_begin; base_ll < r_end; base_ll += kChunk5s) {
        const int base = (int)base_ll;
        const int npts = min(kChunk5s, r_end - base);
        // phase A: lane (32 h + j) prepares position base + j of its half: corner ids and weights, already permuted into their
        // slots.  Slot k takes corner k ^ s, s = 2 (y0 & 1) + (x0 & 1): the four corners of a texel always take four different
        // slots and a row keeps its slot when the walk moves to a neighbouring texel.  The carried plane's two line rows take the
        // slot of their parity the same way.  A position whose cell (ride row) equals its predecessor's, both with every corner
        // inside, gets flag bit 0 (1) clear: phase B then only accumulates, without looking at the ids.
        __builtin_amdgcn_wave_barrier();
        {
            const bool on = ch < npts;
            const int g = (int)ord[base + (on ? ch : 0)];
            const float cx = (xyz[3 * g + ca] - lo_a) * sc_a - 1.0f, cy = (xyz[3 * g + cb] - lo_b) * sc_b - 1.0f;
            float gxm, gym;
            const float ix = unnorm_clip(cx, Wd, gxm), iy = unnorm_clip(cy, Hd, gym);
            const int x0 = (int)floorf(ix), y0 = (int)floorf(iy);
            const bool hx = x0 + 1 < Wd, hy = y0 + 1 < Hd;
            const int o00 = (y0 * Wd + x0) * 32;
            const float bx = ix - (float)x0, by = iy - (float)y0, ax = 1.f - bx, ay = 1.f - by;
            const int ids[4] = {o00, hx ? o00 + 32 : -2, hy ? o00 + Wd * 32 : -2, (hx && hy) ? o00 + Wd * 32 + 32 : -2};
            const float ws[4] = {ax * ay, bx * ay, ax * by, bx * by};
            const bool sx1 = x0 & 1, sy1 = y0 & 1;
            const int i0 = sx1 ? ids[1] : ids[0], i1 = sx1 ? ids[0] : ids[1], i2 = sx1 ? ids[3] : ids[2], i3 = sx1 ? ids[2] : ids[3];
            const float f0 = sx1 ? ws[1] : ws[0], f1 = sx1 ? ws[0] : ws[1], f2 = sx1 ? ws[3] : ws[2], f3 = sx1 ? ws[2] : ws[3];
            // carried plane: rows r0 (weight 1 - b) and r0 + 1 (weight b) along the shared axis
            const int r0 = ride_on_b ? y0 : x0;
            const bool hr = ride_on_b ? hy : hx;
            const float bw = ride_on_b ? by : bx, aw = 1.f - bw;
            const bool odd = r0 & 1;
            const int ra = r0 * 32, rb = hr ? (r0 + 1) * 32 : -2;
            // run detection: a position with a corner outside gets a unique negative value, so neither it nor its successor
            // compares equal
            const int cell = (hx && hy) ? o00 : -2 - ch, row = hr ? r0 : -2 - ch;
            int pc = __shfl_up(cell, 1), pr = __shfl_up(row, 1);
            if (ch == 0) { pc = prev_cell; pr = prev_row; }
            const int flag = (cell != pc ? 1 : 0) | (row != pr ? 2 : 0);
            prev_cell = __shfl(cell, 32 * h + max(npts, 1) - 1);
            prev_row = __shfl(row, 32 * h + max(npts, 1) - 1);
            if (on) {
                w_ids[lane] = make_int4(sy1 ? i2 : i0, sy1 ? i3 : i1, sy1 ? i0 : i2, sy1 ? i1 : i3);
                w_ws[lane] = make_float4(sy1 ? f2 : f0, sy1 ? f3 : f1, sy1 ? f0 : f2, sy1 ? f1 : f3);
                w_rd[lane] = make_float4(odd ? bw : aw, odd ? aw : bw, __int_as_float(flag), 0.f);
                w_rid[lane] = make_int2(odd ? rb : ra, odd ? ra : rb);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // phase B: the gv rows of this order are consecutive in memory: a batch is loaded, then consumed with no vector-memory
        // wait inside (the rare atomics never sit between a load and its use)
        for (int b0 = 0; b0 < npts; b0 += kBatch5s) {
            float val[kBatch5s], vat[kBatch5s];
#pragma unroll
            for (int j = 0; j < kBatch5s; j++) {
                const int q = min(b0 + j, npts - 1);
                val[j] = src_s[(size_t)(base + q) * kRowF];
                vat[j] = CROWS ? 0.f : src_t[(size_t)(base + q) * kRowF];
            }
#pragma unroll
            for (int j = 0; j < kBatch5s; j++) {
                if (b0 + j >= npts) continue;
                const float4 w4 = w_ws[32 * h + b0 + j];
                const float4 rd = w_rd[32 * h + b0 + j];
                const int flag = __float_as_int(rd.z);
                const float ws[4] = {w4.x, w4.y, w4.z, w4.w};
                const float lw[2] = {rd.x, rd.y};
                float g_space = val[j], g_time = vat[j];      // the two planes' gv at this position
                if (CROWS) {
                    // the time line's rows first (their values give the space plane's gv)
                    if (flag & 2) {
                        const int2 r2 = w_rid[32 * h + b0 + j];
                        const int lid[2] = {r2.x, r2.y};
#pragma unroll
                        for (int k = 0; k < 2; k++) {
                            if (lid[k] != lpid[k] && lid[k] >= 0) {
                                if (lpid[k] >= 0) atomicAdd(&my_line[lpid[k]], lacc[k]);
                                lpid[k] = lid[k];
                                lacc[k] = 0.f;
                                lv[k] = my_lval[lid[k]];
                            }
                        }
                    }
                    g_space = val[j] * (lw[0] * lv[0] + lw[1] * lv[1]);
                }
                if (flag & 1) {
                    // the cell changed (or touches the border): a slot whose row differs flushes its pending row and restarts
                    const int4 id4 = w_ids[32 * h + b0 + j];
                    const int ids[4] = {id4.x, id4.y, id4.z, id4.w};
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if (ids[k] == pid[k]) {
                            pacc[k] = __builtin_fmaf(g_space, ws[k], pacc[k]);
                        } else if (ids[k] >= 0) {
                            if (pid[k] >= 0) atomicAdd(&gp[pid[k]], pacc[k]);
                            pid[k] = ids[k];
                            pacc[k] = g_space * ws[k];
                            if (CROWS) tv[k] = plane_v[ids[k]];
                        }
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++) pacc[k] = __builtin_fmaf(g_space, ws[k], pacc[k]);
                }
                if (CROWS) {
                    // the space plane's sample from the pending rows' values (a corner outside the plane has weight exactly 0 and
                    // keeps whatever finite value its slot held), then the time plane's gv into the line accumulators
                    g_time = val[j] * (ws[0] * tv[0] + ws[1] * tv[1] + ws[2] * tv[2] + ws[3] * tv[3]);
#pragma unroll
                    for (int k = 0; k < 2; k++) lacc[k] = __builtin_fmaf(g_time, lw[k], lacc[k]);
                } else if (flag & 2) {
                    const int2 r2 = w_rid[32 * h + b0 + j];
                    const int lid[2] = {r2.x, r2.y};
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        if (lid[k] == lpid[k]) {
                            lacc[k] = __builtin_fmaf(g_time, lw[k], lacc[k]);
                        } else if (lid[k] >= 0) {
                            if (lpid[k] >= 0) atomicAdd(&my_line[lpid[k]], lacc[k]);
                            lpid[k] = lid[k];
                            lacc[k] = g_time * lw[k];
                        }
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 2; k++) lacc[k] = __builtin_fmaf(g_time, lw[k], lacc[k]);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (pid[k] >= 0) atomicAdd(&gp[pid[k]], pacc[k]);
#pragma unroll
    for (int k = 0; k < 2; k++)
        if (lpid[k] >= 0) atomicAdd(&my_line[lpid[k]], lacc[k]);
    __syncthreads();
    // the carried plane's line -> its two global time rows t0 / t1 (the same for every point)
    int t0, t1;
    float wt0, wt1;
    time_sample(a.time, a.res[lvl][3], t0, t1, wt0, wt1);
    float* __restrict__ gt = a.grads[lvl][pt];
    for (int i = threadIdx.x; i < Ws * 32; i += 256) {
        const float sv = s_line[i];
        if (sv != 0.f) {
            if (t0 >= 0) atomicAdd(&gt[(size_t)t0 * Ws * 32 + i], sv * wt0);
            if (t1 >= 0) atomicAdd(&gt[(size_t)t1 * Ws * 32 + i], sv * wt1);
        }
    }
}

// sort key of a point for one space plane: 2-D Morton code of the texel cell it falls into at resolution (Wd, Hd)
__device__ __forceinline__ unsigned spread16(unsigned x)
{
    x &= 0xFFFFu;
    x = (x | (x << 8)) & 0x00FF00FFu;
    x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}
__global__ void __launch_bounds__(256)
plane_key_kernel(HexArgs a, int ca, int cb, int Wd, int Hd, const float* __restrict__ xyz, unsigned* __restrict__ keys,
                 unsigned* __restrict__ idx)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.P) return;
    const float cx = (xyz[3 * i + ca] - a.a0[ca]) * (2.0f / (a.a1[ca] - a.a0[ca])) - 1.0f;
    const float cy = (xyz[3 * i + cb] - a.a0[cb]) * (2.0f / (a.a1[cb] - a.a0[cb])) - 1.0f;
    float gm;
    const int x0 = (int)floorf(unnorm_clip(cx, Wd, gm)), y0 = (int)floorf(unnorm_clip(cy, Hd, gm));
    keys[i] = spread16((unsigned)x0) | (spread16((unsigned)y0) << 1);
    idx[i] = (unsigned)i;
}
__global__ void __launch_bounds__(256)
invert_perm_kernel(int n, const unsigned* __restrict__ order, unsigned* __restrict__ order_out, unsigned* __restrict__ inv)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned g = order[i];
    order_out[i] = g;
    inv[g] = (unsigned)i;
}

}  // namespace

// stable LSD radix sort of (u32 key, u32 value) pairs on the low `bits` bits (knn.hip); returns the index (0 / 1) of the
// buffer pair that holds the result, or a negative MOM_E* code
int mom_sort_pairs_u32(int n, int bits, unsigned* keys[2], unsigned* vals[2], unsigned* counts, hipStream_t s);
size_t mom_sort_pairs_counts_bytes(int n);

extern "C" int mom_hexplane_forward(const MomHexPlane* hp, int P, const float* xyz, const float* times, float time,
                                    const uint32_t* order, float* feat, mom_stream_t stream)
{
    if (!hp || hp->channels != 32 || hp->levels < 1 || hp->levels > 4 || P < 0) return MOM_EINVAL;
    if (P == 0) return MOM_OK;
    if (!xyz || !feat) return MOM_EINVAL;
    HexArgs a;
    fill_args(hp, P, times, time, order, false, &a);
    const long long units = (long long)P * hp->levels;
    MomProfScope ps(MOM_P_HEX_FWD, (hipStream_t)stream);
    if (!times) {
        const int nchunks = (P + kChunkF - 1) / kChunkF;
        int blocks = (nchunks + 3) / 4;
        // one chunk per wave while that stays under 8192 workgroups per level: the dispatcher balances whole workgroups; a cap of
        // 1024 (round 1) gave half of the waves two chunks and the rest one
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(hexplane_fwd4_kernel, dim3(blocks, hp->levels), dim3(256), 0, (hipStream_t)stream, a, nchunks, xyz, feat);
        return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
    }
    hipLaunchKernelGGL(hexplane_fwd_kernel, dim3((unsigned)((units + 7) / 8)), dim3(256), 0, (hipStream_t)stream, a, xyz, feat);
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}

extern "C" size_t mom_hexplane_orders_scratch_bytes(int P)
{
    const size_t n = (size_t)(P > 0 ? P : 1);
    return 4 * mom_align_up(n * 4) + mom_align_up(mom_sort_pairs_counts_bytes(P)) + MOM_ALIGN;
}

extern "C" int mom_hexplane_orders(const MomHexPlane* hp, int P, const float* xyz, uint32_t* order, uint32_t* inverse, void* scratch,
                                   mom_stream_t stream)
{
    if (!hp || hp->levels < 1 || hp->levels > 4 || P < 0) return MOM_EINVAL;
    if (P == 0) return MOM_OK;
    if (!xyz || !order || !inverse || !scratch) return MOM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    HexArgs a;
    fill_args(hp, P, nullptr, 0.f, nullptr, false, &a);
    const size_t n = (size_t)P;
    char* base = mom_align_ptr(scratch);
    unsigned* keys[2]; unsigned* vals[2];
    keys[0] = (unsigned*)base; base += mom_align_up(n * 4);
    keys[1] = (unsigned*)base; base += mom_align_up(n * 4);
    vals[0] = (unsigned*)base; base += mom_align_up(n * 4);
    vals[1] = (unsigned*)base; base += mom_align_up(n * 4);
    unsigned* counts = (unsigned*)base;
    // one order per (space plane, level): a level's cells do not nest in a finer level's (align_corners scales by size - 1)
    const int axes[3][2] = {{0, 1}, {0, 2}, {1, 2}};
    for (int si = 0; si < 3; si++)
        for (int l = 0; l < hp->levels; l++) {
            const int ca = axes[si][0], cb = axes[si][1], Wd = hp->res[l][ca], Hd = hp->res[l][cb];
            if (Wd < 1 || Hd < 1 || Wd > 65536 || Hd > 65536) return MOM_EINVAL;
            int bits = 0;
            while ((1 << bits) < (Wd > Hd ? Wd : Hd)) bits++;
            hipLaunchKernelGGL(plane_key_kernel, dim3((P + 255) / 256), dim3(256), 0, s, a, ca, cb, Wd, Hd, xyz, keys[0], vals[0]);
            const int cur = mom_sort_pairs_u32(P, 2 * bits, keys, vals, counts, s);
            if (cur < 0) return cur;
            const size_t off = ((size_t)si * hp->levels + l) * n;
            hipLaunchKernelGGL(invert_perm_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, vals[cur], order + off, inverse + off);
        }
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}

// deform_field.hip: the gather in the fused forward's layout, for fields mom_deform_field_supported() accepts
size_t mom_hexplane_lines_bytes(const MomHexPlane* hp);
int mom_launch_hexplane_gather6(const MomHexPlane* hp, int P, const float* xyz, float time, const uint32_t* order, const float* dfeat,
                                float* dxyz, const uint32_t* plane_inverse, float* gvbuf, float* lines, bool lines_ready, bool crows, hipStream_t s);
extern "C" int mom_deform_field_supported(const MomHexPlane* hp);

static size_t gv_bytes(const MomHexPlane* hp, int P) { return mom_align_up((size_t)6 * (size_t)P * (size_t)hp->levels * 32 * sizeof(float)); }

extern "C" size_t mom_hexplane_backward_scratch_bytes(const MomHexPlane* hp, int P)
{
    if (!hp || P <= 0) return MOM_ALIGN;
    return gv_bytes(hp, P) + mom_hexplane_lines_bytes(hp) + MOM_ALIGN;    // gv rows | this frame's time lines
}

static int hexplane_backward(const MomHexPlane* hp, int P, const float* xyz, const float* times, float time,
                             const uint32_t* order, const float* dfeat, float* dxyz, const uint32_t* plane_order,
                             const uint32_t* plane_inverse, void* scratch, const void* field_scratch, mom_stream_t stream);

extern "C" int mom_hexplane_backward(const MomHexPlane* hp, int P, const float* xyz, const float* times, float time,
                                     const uint32_t* order, const float* dfeat, float* dxyz, const uint32_t* plane_order,
                                     const uint32_t* plane_inverse, void* scratch, mom_stream_t stream)
{
    return hexplane_backward(hp, P, xyz, times, time, order, dfeat, dxyz, plane_order, plane_inverse, scratch, nullptr, stream);
}

extern "C" int mom_hexplane_backward_lines(const MomHexPlane* hp, int P, const float* xyz, float time, const uint32_t* order,
                                           const float* dfeat, float* dxyz, const uint32_t* plane_order,
                                           const uint32_t* plane_inverse, void* scratch, const void* field_scratch,
                                           mom_stream_t stream)
{
    if (!field_scratch || !plane_order || !plane_inverse || !scratch) return MOM_EINVAL;
    return hexplane_backward(hp, P, xyz, nullptr, time, order, dfeat, dxyz, plane_order, plane_inverse, scratch, field_scratch, stream);
}

static int hexplane_backward(const MomHexPlane* hp, int P, const float* xyz, const float* times, float time,
                             const uint32_t* order, const float* dfeat, float* dxyz, const uint32_t* plane_order,
                             const uint32_t* plane_inverse, void* scratch, const void* field_scratch, mom_stream_t stream)
{
    if (!hp || hp->channels != 32 || hp->levels < 1 || hp->levels > 4 || P < 0) return MOM_EINVAL;
    if (P == 0) return MOM_OK;
    if (!xyz || !dfeat) return MOM_EINVAL;
    HexArgs a;
    fill_args(hp, P, times, time, order, true, &a);
    for (int l = 0; l < hp->levels; l++)
        for (int p = 0; p < 6; p++)
            if (!a.planes[l][p] || !a.grads[l][p]) return MOM_EINVAL;
    const long long units = (long long)P * hp->levels;
    MomProfScope ps(MOM_P_HEX_BWD, (hipStream_t)stream);
    int wmax = 0;
    for (int l = 0; l < hp->levels; l++)
        for (int k = 0; k < 3; k++)
            if (hp->res[l][k] > wmax) wmax = hp->res[l][k];
    const size_t lds_s = sizeof(float) * ((size_t)4 * 64 * kRec5s + (size_t)wmax * 32);
    // gv rows are addressed with 32-bit byte offsets inside one (plane, level) buffer
    const bool fits32 = (unsigned long long)P * 256ull < (1ull << 32);
    if (!times && plane_order && plane_inverse && scratch && lds_s <= 160 * 1024 && fits32) {
        // two-pass path: one shared timestamp, per-plane orders and the gv scratch given
        static bool attr_set = false;
        if (!attr_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(hexplane_bwd5_scatter_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(hexplane_bwd5_scatter_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024) != hipSuccess)
                return MOM_ELAUNCH;
            attr_set = true;
        }
        float* gvbuf = (float*)mom_align_ptr(scratch);
        static int blocks_g = 0, blocks_s = 0;
        if (!blocks_g) {
            const char* e = getenv("MOM_HEX_BLOCKS");
            blocks_g = e ? atoi(e) : 1536;
            const char* e2 = getenv("MOM_HEX_SBLOCKS");
            blocks_s = e2 ? atoi(e2) : 512;
        }
        // MOM_HEX_GATHER=5: the lane-per-channel gather (measurement, and the comparison in tests/test_ops_gpu.py; read per call)
        const char* e_g = getenv("MOM_HEX_GATHER");
        const int gather6 = (e_g && e_g[0] == '5') ? 0 : 1;
        const size_t lds_c = lds_s;
        const bool use6 = gather6 && mom_deform_field_supported(hp);
        // One common-factor row per (slot, position) instead of the slot's two gv rows (see the scatter kernel); MOM_HEX_CROWS=0: the
        // six-row form.  Measured at 200 k Gaussians: alone the two passes take 83 + 81 us instead of 105 + 84 and move 307 MB
        // less; beside the MLP's weight-gradient kernel, where they run in the training step, 117 + 141 against 160 + 105 (the
        // scatter now waits for the texel values of every new cell in the middle of its walk, and a walk that waits suffers more
        // from a neighbour than one that streams).  The step, same box, alternating runs: +0.2 ... +1.4 % at 200 k Gaussians,
        // +5 % at 1 M (1080p), +6 % at 4 M.  (Staging the line values in LDS cost two of five workgroups per CU: 106 us; requesting
        // the flagged positions' texel values with the batch's rows, after a scan of the batch's flags: 116 us.)
        const char* e_c = getenv("MOM_HEX_CROWS");
        const bool crows = use6 && !(e_c && e_c[0] == '0');
        const float* lines_c = nullptr;
        LineTab lt;
        line_table(hp, &lt);
        if (use6) {
            // the frame's time lines: the table mom_deform_field_forward left at the head of its scratch, or computed here
            float* lines = field_scratch ? (float*)mom_align_ptr(const_cast<void*>(field_scratch))
                                         : (float*)mom_align_ptr((char*)gvbuf + gv_bytes(hp, P));
            int rc = mom_launch_hexplane_gather6(hp, P, xyz, time, order, dfeat, dxyz, plane_inverse, gvbuf, lines, field_scratch != nullptr,
                                                 crows, (hipStream_t)stream);
            if (rc) return rc;
            lines_c = lines;
        } else {
            const int nchunks = (P + kChunk5 - 1) / kChunk5;
            int blocks = (nchunks + 3) / 4;
            if (blocks > blocks_g) blocks = blocks_g;
            hipLaunchKernelGGL(hexplane_bwd5_gather_kernel, dim3(blocks, hp->levels), dim3(256), 0, (hipStream_t)stream, a, nchunks, xyz,
                               dfeat, dxyz, plane_inverse, gvbuf);
            if (hipGetLastError() != hipSuccess) return MOM_ELAUNCH;
        }
        {
            // every half-wave walks one contiguous range of sorted positions (a multiple of the chunk size)
            const int halves = blocks_s * 8;
            int per_half = (P + halves - 1) / halves;
            per_half = ((per_half + kChunk5s - 1) / kChunk5s) * kChunk5s;
            const int blocks = (int)(((long long)P + (long long)per_half * 8 - 1) / ((long long)per_half * 8));
            if (crows)       // the gather left one common-factor row per (slot, position): this pass forms the two gv rows itself
                hipLaunchKernelGGL(hexplane_bwd5_scatter_kernel<true>, dim3(blocks, 3, hp->levels), dim3(256), lds_c, (hipStream_t)stream, a,
                                   per_half, xyz, plane_order, gvbuf, lines_c, lt);
            else
                hipLaunchKernelGGL(hexplane_bwd5_scatter_kernel<false>, dim3(blocks, 3, hp->levels), dim3(256), lds_s, (hipStream_t)stream, a,
                                   per_half, xyz, plane_order, gvbuf, lines_c, lt);
        }
        return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
    }
    // generic path (per-point timestamps, or no orders / scratch given): one half-wave per (point, level), 24 atomic rows each
    hipLaunchKernelGGL(hexplane_bwd_kernel, dim3((unsigned)((units + 7) / 8)), dim3(256), 0, (hipStream_t)stream, a, xyz, dfeat, dxyz);
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}
