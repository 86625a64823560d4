// Device helpers of the tile binning shared by raster_binning.hip and raster_preprocess.hip (the projection kernel also decides
// which (splat, tile) instances are binned and counts them per tile: the per-(splat, tile) test sits in a kernel that waits for
// memory most of the time, and the separate histogram launch with its second read of the records is gone).
#pragma once
#include "mom_common.h"

namespace {

// Splats touching at most this many tiles are enumerated by their own lane; larger ones by the whole wave, one tile per lane, which
// costs a broadcast of the splat (fifteen v_readlane) and a wave-wide pass per splat.  With the threshold at 8 (rounds 1-3) a third
// of the visible splats took the wave-wide path and it was 80 % of tile_hist's vector instructions; measured at config 2,
// threshold -> tile_hist: 4 -> 37 us, 8 -> 32, 12 -> 23, 16 -> 21, 24 -> 20, 32 -> 20, 48 -> 21, 64 -> 25 (config 3: 74 -> 46 at 24).
#ifndef MOM_SMALL_RECT
#define MOM_SMALL_RECT 24
#endif
constexpr int kSmallRect = MOM_SMALL_RECT;
constexpr int kMaxLdsTiles = 16384;    // 64 KiB LDS histogram


// What the tile cull needs of a splat (mom_rect_reach, mom_common.h).  cull == 0: every tile of the rectangle is kept, as
// the reference does (MomRasterArgs.keep_all_tiles).
struct Reach {
    float cx, cy, a, b, c, bound, inv_a, inv_c;
    int cull;
};
__device__ __forceinline__ bool tile_reached(const Reach& r, int tx, int ty)
{
    if (!r.cull) return true;
    const float xa = (float)(tx * MOM_TILE), ya = (float)(ty * MOM_TILE);
    return mom_rect_reach(r.cx, r.cy, r.a, r.b, r.c, r.bound, r.inv_a, r.inv_c, xa, xa + (float)(MOM_TILE - 1), ya,
                          ya + (float)(MOM_TILE - 1));
}
__device__ __forceinline__ float bcast(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }

struct WaveSplat {       // one lane's splat, broadcast to the wave
    int x0, y0, w, cnt;
    uint32_t payload;
    uint64_t mask;
    Reach rc;
};
__device__ __forceinline__ WaveSplat bcast_splat(int x0, int y0, int w, int cnt, uint32_t payload, uint64_t mask, const Reach& rc, int src)
{
    WaveSplat o;
    o.x0 = __builtin_amdgcn_readlane(x0, src);
    o.y0 = __builtin_amdgcn_readlane(y0, src);
    o.w = __builtin_amdgcn_readlane(w, src);
    o.cnt = __builtin_amdgcn_readlane(cnt, src);
    o.payload = (uint32_t)__builtin_amdgcn_readlane((int)payload, src);
    o.mask = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(mask >> 32), src) << 32) |
             (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mask, src);
    o.rc.cx = bcast(rc.cx, src); o.rc.cy = bcast(rc.cy, src); o.rc.a = bcast(rc.a, src); o.rc.b = bcast(rc.b, src);
    o.rc.c = bcast(rc.c, src); o.rc.bound = bcast(rc.bound, src); o.rc.inv_a = bcast(rc.inv_a, src);
    o.rc.inv_c = bcast(rc.inv_c, src);
    o.rc.cull = __builtin_amdgcn_readlane(rc.cull, src);
    return o;
}
// i / w for 0 <= i, 1 <= w <= 64 * 64: (i + 0.5) / w is at least 0.5 / w away from an integer, far more than the error of
// the hardware reciprocal and the product
__device__ __forceinline__ int small_div(int i, float inv_w) { return (int)(((float)i + 0.5f) * inv_w); }

// Pass 1 (tile_hist): decides, for every (Gaussian, tile) instance of this wave's 64 Gaussians, whether it is binned
// (tile_reached), calls f(tile) for those that are, and returns the lane's own decisions as a mask: bit i = tile i of the
// rectangle, row-major, for i < 64 (tile_scatter evaluates tiles beyond 64 again).  Lanes own small rectangles; large ones
// are walked by the whole wave, one tile per lane, and the ballot IS the mask.
template <class F>
__device__ __forceinline__ uint64_t decide_instances(int x0, int y0, int x1, int y1, int gx, const Reach& rc, F f)
{
    const int lane = mom_lane();
    const int w = x1 - x0;
    const int cnt = w * (y1 - y0);
    uint64_t mask = 0;
    if (cnt <= kSmallRect) {
        int tx = x0, ty = y0;
        for (int i = 0; i < cnt; i++) {
            if (tile_reached(rc, tx, ty)) {
                mask |= 1ull << i;
                f(ty * gx + tx);
            }
            if (++tx == x1) { tx = x0; ty++; }
        }
    }
    unsigned long long big = __ballot(cnt > kSmallRect);
    while (big) {
        const int src = __ffsll((long long)big) - 1;
        big &= big - 1;
        const WaveSplat b = bcast_splat(x0, y0, w, cnt, 0u, 0ull, rc, src);
        const float inv_w = __builtin_amdgcn_rcpf((float)b.w);
        for (int base = 0; base < b.cnt; base += MOM_WAVE) {
            const int i = base + lane;
            bool r = false;
            if (i < b.cnt) {
                const int q = small_div(i, inv_w);
                const int tx = b.x0 + i - q * b.w, ty = b.y0 + q;
                r = tile_reached(b.rc, tx, ty);
                if (r) f(ty * gx + tx);
            }
            if (base == 0) {
                const uint64_t bal = __ballot(r);
                if (lane == src) mask = bal;
            }
        }
    }
    return mask;
}


}  // namespace

// ---- per-tile sort ------------------------------------------------------------
// Bitonic network in the "flip then disperse" form: every compare-exchange puts
// the smaller key at the lower index, so virtual +inf padding above n never moves
// and is simply skipped.
//
// All strides are powers of two: indices come from shifts and masks (a division by a run-time stride costs ~40
// instructions per compare-exchange).  Thread t works on the compare-exchanges i = t, t + nthreads, ...; a wave's 64
// consecutive i touch one aligned block of 128 keys whenever the stage's span (kk for a flip, 2j for a disperse step)
// is at most 128, so two such stages in a row exchange data inside the wave only and need no workgroup barrier --
// for 1024 keys that leaves 6 of 55.  A barrier is kept wherever either neighbour stage is wider.  With BLOCK_SYNC
// false (the global-memory path of oversized buckets) every stage keeps its barrier.
template <bool LOCAL_STAGES, class KeyPtr>
__device__ __forceinline__ void bitonic_sort(KeyPtr k, int n, int nthreads, int tid)
{
    int lm = 0;
    while ((1 << lm) < n) lm++;
    const int half_m = (1 << lm) >> 1;
    bool prev_wide = true;                                   // the loads before the first stage came from all waves
    auto sync_before = [&](int span) {
        const bool wide = !LOCAL_STAGES || span > 128;
        if (wide || prev_wide) __syncthreads();
        else __builtin_amdgcn_wave_barrier();
        prev_wide = wide;
    };
    for (int lk = 1; lk <= lm; lk++) {
        const int kk = 1 << lk, lh = lk - 1, half = kk >> 1;
        sync_before(kk);
        for (int i = tid; i < half_m; i += nthreads) {
            const int blk = i >> lh, off = i & (half - 1);
            const int a = (blk << lk) + off, b = (blk << lk) + kk - 1 - off;
            if (b < n) {
                const uint64_t ka = k[a], kb = k[b];
                if (ka > kb) { k[a] = kb; k[b] = ka; }
            }
        }
        for (int lj = lk - 2; lj >= 0; lj--) {
            const int j = 1 << lj;
            sync_before(2 * j);
            for (int i = tid; i < half_m; i += nthreads) {
                const int a = ((i >> lj) << (lj + 1)) + (i & (j - 1)), b = a + j;
                if (b < n) {
                    const uint64_t ka = k[a], kb = k[b];
                    if (ka > kb) { k[a] = kb; k[b] = ka; }
                }
            }
        }
    }
    __syncthreads();
}


constexpr int kRenderSortCap = 1536;
