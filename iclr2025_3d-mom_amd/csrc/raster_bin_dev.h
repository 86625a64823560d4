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
// Bitonic network in the "flip then disperse" form: every compare-exchange puts the smaller key at the lower index, so virtual +inf
// padding above n never moves (it is loaded as UINT64_MAX -- every real key, depth_bits << 32 | index, is smaller -- and never stored).
//
// RADIX 4 (round 6): a pass loads FOUR keys, applies two consecutive stages of the network to them in registers and stores them.
// The stages of block size kk are  flip(kk), disperse(kk/4), disperse(kk/8), ..., disperse(1):  flip(kk) + disperse(kk/4) close over
// {a, a + kk/4, kk-1-a-kk/4, kk-1-a} (a in the block's first quarter), disperse(j) + disperse(j/2) over {x, x + j/2, x + j, x + 3j/2}
// (bits j and j/2 of x clear); a leftover disperse(1) -- and flip(2) -- take two neighbouring pairs.  Same network, same result; the
// passes drop from 45 to 25 for 512 keys and from 55 to 30 for 1024, and with them the LDS round trips and barriers a lone workgroup's
// sort is made of (16 us for 1024 keys, which is what the heaviest tiles of render_fwd spent before their first round) and the LDS
// traffic of eight workgroups per CU sorting at once.
//
// All strides are powers of two: indices come from shifts and masks.  Thread t takes the groups g = t, t + nthreads, ...; a wave's 64
// consecutive groups touch one aligned block of 256 keys whenever the pass's span (kk, or 2j) is at most 256, so consecutive such
// passes exchange data inside the wave only and need no workgroup barrier.  A barrier is kept wherever either neighbour pass is
// wider.  With LOCAL_STAGES false (the global-memory path of oversized buckets) every pass keeps its barrier.
template <bool LOCAL_STAGES, class KeyPtr>
__device__ __forceinline__ void bitonic_sort(KeyPtr k, int n, int nthreads, int tid)
{
    int lm = 0;
    while ((1 << lm) < n) lm++;
    if (lm == 0) return;
    if (lm == 1) {                                           // two keys
        __syncthreads();
        if (tid == 0) { const uint64_t ka = k[0], kb = k[1]; if (ka > kb) { k[0] = kb; k[1] = ka; } }
        __syncthreads();
        return;
    }
    const int quarter_m = (1 << lm) >> 2;
    constexpr uint64_t kInf = ~0ull;
    bool prev_wide = true;                                   // the loads before the first pass came from all waves
    auto sync_before = [&](int span) {
        const bool wide = !LOCAL_STAGES || span > 256;
        if (wide || prev_wide) __syncthreads();
        else __builtin_amdgcn_wave_barrier();
        prev_wide = wide;
    };
    auto ld = [&](int i) { return i < n ? (uint64_t)k[i] : kInf; };
    auto cx = [](uint64_t& a, uint64_t& b) { const uint64_t lo = a < b ? a : b, hi = a < b ? b : a; a = lo; b = hi; };
    // two neighbouring pairs (4g, 4g+1), (4g+2, 4g+3): flip(2), and the disperse(1) an odd number of disperse stages leaves over
    auto pairs_pass = [&]() {
        sync_before(2);
        for (int g = tid; g < quarter_m; g += nthreads) {
            const int x = 4 * g;
            if (x + 1 < n) {
                uint64_t k0 = k[x], k1 = k[x + 1];
                if (k0 > k1) { k[x] = k1; k[x + 1] = k0; }
            }
            if (x + 3 < n) {
                uint64_t k2 = k[x + 2], k3 = k[x + 3];
                if (k2 > k3) { k[x + 2] = k3; k[x + 3] = k2; }
            }
        }
    };
    auto st4 = [&](int i0, int i1, int i2, int i3, uint64_t k0, uint64_t k1, uint64_t k2, uint64_t k3) {
        k[i0] = k0;                                          // i0 < n by construction of the callers' guards
        if (i1 < n) k[i1] = k1;
        if (i2 < n) k[i2] = k2;
        if (i3 < n) k[i3] = k3;
    };
    for (int lk = 1; lk <= lm; lk++) {
        const int kk = 1 << lk;
        if (lk == 1) { pairs_pass(); continue; }
        // flip(kk) + disperse(kk / 4)
        const int q = kk >> 2;
        sync_before(kk);
        for (int g = tid; g < quarter_m; g += nthreads) {
            const int blk = g >> (lk - 2), a = g & (q - 1), base = blk << lk;
            const int i0 = base + a, i1 = i0 + q, i3 = base + kk - 1 - a, i2 = i3 - q;          // i0 < i1 < i2 < i3
            if (i0 < n && i1 < n) {                          // (with only i0 below n every partner is +inf: nothing moves)
                uint64_t k0 = k[i0], k1 = k[i1], k2 = ld(i2), k3 = ld(i3);
                cx(k0, k3); cx(k1, k2);                      // flip
                cx(k0, k1); cx(k2, k3);                      // disperse(kk / 4)
                st4(i0, i1, i2, i3, k0, k1, k2, k3);
            }
        }
        int lj = lk - 3;
        for (; lj >= 1; lj -= 2) {                           // disperse(2^lj) + disperse(2^(lj-1))
            const int j = 1 << lj, h = j >> 1;
            sync_before(2 * j);
            for (int g = tid; g < quarter_m; g += nthreads) {
                const int x = ((g >> (lj - 1)) << (lj + 1)) | (g & (h - 1));
                const int i0 = x, i1 = x + h, i2 = x + j, i3 = x + j + h;
                if (i1 < n) {                                // (i0 < i1: with i1 >= n the group holds one real key at most)
                    uint64_t k0 = k[i0], k1 = k[i1], k2 = ld(i2), k3 = ld(i3);
                    cx(k0, k2); cx(k1, k3);                  // disperse(j)
                    cx(k0, k1); cx(k2, k3);                  // disperse(j / 2)
                    st4(i0, i1, i2, i3, k0, k1, k2, k3);
                }
            }
        }
        if (lj == 0) pairs_pass();                           // the leftover disperse(1)
    }
    __syncthreads();
}


// Tiles of up to this many keys are sorted inside render_fwd (in the LDS it stages its splats in afterwards); the binning's own sort
// launch takes the rest.  Measured with the radix-4 network (tile_sort us / render_fwd us / step us, one call, three rounds):
// cap 1536: 8.7 / 117.5 / 859;  cap 1024: 14.1 / 116.4 / 861;  with the radix-2 network cap 512 read 31.5 / 112 / 895 -- a lone
// workgroup's sort is latency bound, so what the forward's heavy tiles save by arriving sorted the extra launch pays twice.
constexpr int kRenderSortCap = 1536;
