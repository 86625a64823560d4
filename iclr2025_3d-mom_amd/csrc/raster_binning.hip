// Tile binning + per-tile depth sort for the rasterizer, gfx950.
//
// Replaces InclusiveSum + duplicateWithKeys + DeviceRadixSort::SortPairs +
// identifyTileRanges (reference rasterizer_impl.cu:70-138,278-318).  The
// reference sorts all (tile<<32 | depth_bits) keys globally with a stable LSD
// radix sort over instances emitted in ascending Gaussian index, i.e. its
// point_list is ordered by (tile, depth_bits, gaussian_idx) -- a TOTAL order.
// Any algorithm that realises that total order yields the same bits, so this
// file does an MSD split instead:
//   1. tile_hist    : per-workgroup LDS histogram of the instances per tile
//                     (wave-cooperative for large splats), flushed with one
//                     global atomic per non-empty (workgroup, tile);
//   2. tile_scan    : one workgroup scans the <=36k tile counters -> ranges,
//                     bucket cursors and the instance count;
//   3. tile_scatter : same enumeration, one returning global atomic per
//                     non-empty (workgroup, tile) reserves a slice of the tile's
//                     bucket, LDS atomics rank inside the slice; writes the
//                     64-bit key (depth_bits<<32 | idx) -- order inside a bucket
//                     is arbitrary here;
//   4. tile_sort    : one workgroup per tile sorts its bucket by the 64-bit key
//                     in LDS (bitonic network, all compare-exchanges ascending,
//                     so no padding is ever materialised) and writes point_list.
// Instances never travel through HBM more than: 8 B write, 8 B read, 4 B write.
#include "raster_bin_dev.h"
#include <stdlib.h>

namespace {

constexpr int kSortLdsCap = 8192;      // 64 KiB of 64-bit keys per workgroup: the largest tile sorted in LDS
#ifndef MOM_SORT_SMALL
#define MOM_SORT_SMALL 2048
#endif
constexpr int kSortSmallCap = MOM_SORT_SMALL;   // tiles up to this size go to the launch with the small LDS footprint
constexpr int kOrderBins = 128;        // weight classes of the tile order (32 instances each; the last one open-ended)

// Pass 2 (tile_scatter): calls f(tile, src_lane, src_payload) for the instances pass 1 kept.  A lane walks the set bits of
// its own mask when they are few; otherwise the wave takes one tile per lane.
template <class F>
__device__ __forceinline__ void for_each_kept(int x0, int y0, int x1, int y1, int gx, uint32_t payload, uint64_t mask, const Reach& rc, F f)
{
    const int lane = mom_lane();
    const int w = x1 - x0;
    const int cnt = w * (y1 - y0);
    const bool own = cnt <= MOM_WAVE && __popcll(mask) <= kSmallRect;
    if (own && cnt > 0) {
        const float inv_w = __builtin_amdgcn_rcpf((float)w);
        uint64_t m = mask;
        while (m) {
            const int i = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int q = small_div(i, inv_w);
            f((y0 + q) * gx + x0 + i - q * w, lane, payload);
        }
    }
    unsigned long long big = __ballot(!own && cnt > 0);
    while (big) {
        const int src = __ffsll((long long)big) - 1;
        big &= big - 1;
        const WaveSplat b = bcast_splat(x0, y0, w, cnt, payload, mask, rc, src);
        const float inv_w = __builtin_amdgcn_rcpf((float)b.w);
        for (int base = 0; base < b.cnt; base += MOM_WAVE) {
            const int i = base + lane;
            if (i < b.cnt) {
                const int q = small_div(i, inv_w);
                const int tx = b.x0 + i - q * b.w, ty = b.y0 + q;
                const bool r = base == 0 ? ((b.mask >> lane) & 1) : tile_reached(b.rc, tx, ty);
                if (r) f(ty * gx + tx, src, b.payload);
            }
        }
    }
}

// What the binning reads of a Gaussian, asked for in ONE go (fetch_rect) and turned into its tile rectangle later (make_rect): the
// conic is not asked for behind the radius test (tile_hist 18.0 -> 16.9 us, tile_scatter 22.1 -> 20.4 at 960x540, 200 k; these kernels'
// waves live as long as their dependent loads).  tile_scatter also asks for its first chunk BEFORE it clears its LDS histogram
// (-0.3 us; the same in tile_hist cost 0.5 us and is not done) and keeps a lone chunk in registers for its second pass (-1.2 us).
struct RectIn {
    float4 r0, r1;
    int radius;
};
__device__ __forceinline__ RectIn fetch_rect(const float4* __restrict__ rec, int g, int P)
{
    RectIn in;
    in.r0 = in.r1 = make_float4(0.f, 0.f, 0.f, 0.f);
    in.radius = 0;
    if (g < P) {
        in.r0 = rec[3 * (size_t)g];
        in.r1 = rec[3 * (size_t)g + 1];
        in.radius = __float_as_int(rec[3 * (size_t)g + 2].w);
    }
    return in;
}
// ry0, ry1: the tile rows this launch bins (tile-row shard); a splat's rectangle is cut to them.
__device__ __forceinline__ void make_rect(const RectIn& in, int gx, int gy, int ry0, int ry1, int cull,
                                          int& x0, int& y0, int& x1, int& y1, uint32_t& depth_bits, Reach& rc)
{
    x0 = y0 = x1 = y1 = 0;
    depth_bits = 0;
    rc = Reach{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0};
    const float4 r0 = in.r0, r1 = in.r1;
    const int radius = in.radius;
    asm volatile("" :: "v"(r0.x), "v"(r0.y), "v"(r0.z), "v"(r1.x), "v"(r1.y), "v"(r1.z), "v"(r1.w), "v"(radius));     // all of it has arrived here
    if (radius > 0) {                                       // (zero for g >= P: fetch_rect)
        mom_get_rect(r0.x, r0.y, radius, gx, gy, x0, y0, x1, y1);
        y0 = max(y0, ry0);
        y1 = min(y1, ry1);
        if (y1 <= y0) x0 = y0 = x1 = y1 = 0;
        depth_bits = __float_as_uint(r0.z);
        // a degenerate conic counts as reachable everywhere, like in the compositing kernels' strip test
        if (cull && r1.x > 0.f && r1.z > 0.f)
            rc = Reach{r0.x, r0.y, r1.x, r1.y, r1.z, mom_power_bound(r1.w), __builtin_amdgcn_rcpf(r1.x), __builtin_amdgcn_rcpf(r1.z), 1};
    }
}

template <bool LDS_HIST>
__global__ void __launch_bounds__(256) tile_hist_kernel(int P, int chunks, int gx, int gy, int ry0, int ry1, int cull, const float4* __restrict__ rec,
                                                       uint32_t* __restrict__ tile_counts, unsigned long long* __restrict__ reach)
{
    extern __shared__ uint32_t s_cnt[];
    const int tiles = gx * gy;
    if (LDS_HIST) {
        for (int t = threadIdx.x; t < tiles; t += 256) s_cnt[t] = 0;
        __syncthreads();
    }
    for (int c = 0; c < chunks; c++) {
        const int g = (blockIdx.x * chunks + c) * 256 + threadIdx.x;
        int x0, y0, x1, y1;
        uint32_t db;
        Reach rc;
        make_rect(fetch_rect(rec, g, P), gx, gy, ry0, ry1, cull, x0, y0, x1, y1, db, rc);
        const uint64_t mask = decide_instances(x0, y0, x1, y1, gx, rc, [&](int tile) {
            if (LDS_HIST)
                atomicAdd(&s_cnt[tile], 1u);
            else
                atomicAdd(&tile_counts[tile], 1u);
        });
        if (g < P) reach[g] = mask;
    }
    if (LDS_HIST) {
        __syncthreads();
        for (int t = threadIdx.x; t < tiles; t += 256) {
            const uint32_t n = s_cnt[t];
            if (n) atomicAdd(&tile_counts[t], n);
        }
    }
}

// One workgroup (1024 threads): exclusive scan of the tile counters.
__global__ void __launch_bounds__(1024) tile_scan_kernel(int tiles, const uint32_t* __restrict__ tile_counts,
                                                        uint32_t* __restrict__ tile_cursor, uint2* __restrict__ ranges,
                                                        uint32_t* __restrict__ hdr, uint32_t* __restrict__ num_rendered_dev,
                                                        uint32_t* __restrict__ num_rendered_host, int t0, int nt,
                                                        uint32_t* __restrict__ tile_order)
{
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    __shared__ uint32_t s_bin[kOrderBins + 1];
    __shared__ uint32_t s_big;             // tiles of at least kRenderSortCap instances: the head of the order
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    for (int i = threadIdx.x; i <= kOrderBins; i += 1024) s_bin[i] = 0;
    auto bin_of = [&](uint32_t n) { return kOrderBins - 1 - (int)min((uint32_t)(kOrderBins - 1), n >> 5); };
    __syncthreads();
    for (int base = 0; base < tiles; base += 1024) {
        const int t = base + threadIdx.x;
        const uint32_t n = t < tiles ? tile_counts[t] : 0u;
        if (t >= t0 && t < t0 + nt) atomicAdd(&s_bin[bin_of(n) + 1], 1u);     // class sizes of the tile order (below)
        uint32_t incl = n;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; w++) woff += s_wave[w];
        const uint32_t carry = s_carry;
        const uint32_t start = carry + woff + incl - n;
        if (t < tiles) {
            tile_cursor[t] = start;
            ranges[t] = n ? make_uint2(start, start + n) : make_uint2(0u, 0u);  // empty tiles stay (0,0)
        }
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = start + n;
        __syncthreads();
    }
    // The launch's tiles [t0, t0 + nt) in order of decreasing instance count (a counting sort over kOrderBins classes of 32
    // instances; the order inside a class is whatever the atomics make it -- it only places work).  The compositing kernels
    // take their tiles in this order: the dispatcher deals consecutive workgroups to different XCDs and CUs, so every CU gets
    // tiles of every weight class, and the workgroups that start late, when the first ones retire, are the light ones.
    if (threadIdx.x < 64) {                 // s_bin[b + 1] = size of class b  ->  s_bin[b] = first position of class b (one wave, two classes per lane)
        static_assert(kOrderBins == 128, "two classes per lane of one wave");
        const uint32_t c0 = s_bin[2 * lane + 1], c1 = s_bin[2 * lane + 2];
        uint32_t incl = c0 + c1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        const uint32_t excl = incl - (c0 + c1);
        s_bin[2 * lane] = excl;
        s_bin[2 * lane + 1] = excl + c0;
        // the classes are 32 instances wide and kRenderSortCap is a multiple of 32: every tile the binning's own sort launch has to
        // take (more keys than the compositing forward sorts itself) sits in front of the first class below kRenderSortCap / 32
        static_assert(kRenderSortCap % 32 == 0 && kRenderSortCap / 32 < kOrderBins, "class boundary");
        constexpr int kFirstSmall = kOrderBins - kRenderSortCap / 32;          // first class whose tiles all have < kRenderSortCap keys
        if (2 * lane == kFirstSmall) s_big = excl;
        if (2 * lane + 1 == kFirstSmall) s_big = excl + c0;
    }
    __syncthreads();
    for (int t = t0 + threadIdx.x; t < t0 + nt; t += 1024) tile_order[atomicAdd(&s_bin[bin_of(tile_counts[t])], 1u)] = (uint32_t)t;
    if (threadIdx.x == 0) {
        hdr[0] = s_carry;
        hdr[3] = (uint32_t)t0;              // the range the order covers: a later launch over other rows ignores the order
        hdr[4] = (uint32_t)nt;
        hdr[5] = s_big;                     // how many entries at the head of the order can exceed kRenderSortCap keys
        *num_rendered_dev = s_carry;
        // device-accessible pinned host memory: the count reaches the host without a copy command behind this kernel
        if (num_rendered_host) __hip_atomic_store(num_rendered_host, s_carry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <bool LDS_HIST>
__global__ void __launch_bounds__(256) tile_scatter_kernel(int P, int chunks, int gx, int gy, int ry0, int ry1, int cull, const float4* __restrict__ rec,
                                                          const unsigned long long* __restrict__ reach,
                                                          uint32_t* __restrict__ tile_cursor, uint64_t* __restrict__ keys,
                                                          uint32_t capacity, uint32_t* __restrict__ hdr, uint32_t* __restrict__ sticky,
                                                          uint32_t tag)
{
    extern __shared__ uint32_t s_cnt[];
    const int tiles = gx * gy;
    bool overflow = false;
    // the first chunk's inputs: asked for before the histogram is cleared, and -- a workgroup with ONE chunk, every launch up to
    // 524 288 Gaussians -- kept in registers for the second pass instead of being fetched again
    const int g_first = blockIdx.x * chunks * 256 + threadIdx.x;
    RectIn in = fetch_rect(rec, g_first, P);
    unsigned long long rmask = g_first < P ? reach[g_first] : 0ull;
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    uint32_t db = 0;
    Reach rc;
    if (LDS_HIST) {
        for (int t = threadIdx.x; t < tiles; t += 256) s_cnt[t] = 0;
        __syncthreads();
        for (int c = 0; c < chunks; c++) {
            const int g = (blockIdx.x * chunks + c) * 256 + threadIdx.x;
            if (c) {
                rmask = g < P ? reach[g] : 0ull;
                in = fetch_rect(rec, g, P);
            }
            make_rect(in, gx, gy, ry0, ry1, cull, x0, y0, x1, y1, db, rc);
            for_each_kept(x0, y0, x1, y1, gx, 0u, rmask, rc, [&](int tile, int, uint32_t) { atomicAdd(&s_cnt[tile], 1u); });
        }
        __syncthreads();
        // reserve this workgroup's slice of every non-empty bucket.  Eight tiles per thread at a time with all eight returning atomics
        // in flight before the first is waited for: written as a plain loop the compiler waits for each one before it issues the next
        // (read count, atomic, s_waitcnt vmcnt(0), write base -- eight L2 round trips in a row at 960x540)
        for (int t0 = threadIdx.x; t0 < tiles; t0 += 8 * 256) {
            uint32_t n[8], base[8];
#pragma unroll
            for (int k = 0; k < 8; k++) n[k] = t0 + 256 * k < tiles ? s_cnt[t0 + 256 * k] : 0u;
#pragma unroll
            for (int k = 0; k < 8; k++) base[k] = n[k] ? atomicAdd(&tile_cursor[t0 + 256 * k], n[k]) : 0u;
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (n[k]) s_cnt[t0 + 256 * k] = base[k];
        }
        __syncthreads();
    }
    for (int c = 0; c < chunks; c++) {
        const int g = (blockIdx.x * chunks + c) * 256 + threadIdx.x;
        const int wave_g0 = g - mom_lane();
        if (chunks > 1 || !LDS_HIST) {                       // (one chunk with the LDS histogram: everything is still in registers)
            rmask = g < P ? reach[g] : 0ull;
            if (c || LDS_HIST) in = fetch_rect(rec, g, P);
            make_rect(in, gx, gy, ry0, ry1, cull, x0, y0, x1, y1, db, rc);
        }
        for_each_kept(x0, y0, x1, y1, gx, db, rmask, rc, [&](int tile, int src, uint32_t sdb) {
            const uint32_t pos = LDS_HIST ? atomicAdd(&s_cnt[tile], 1u) : atomicAdd(&tile_cursor[tile], 1u);
            if (pos < capacity)
                keys[pos] = ((uint64_t)sdb << 32) | (uint32_t)(wave_g0 + src);
            else
                overflow = true;
        });
    }
    if (overflow) {
        atomicOr(&hdr[1], 1u);
        if (sticky) atomicCAS(sticky, 0u, tag);         // the FIRST overflow since the caller last cleared the word leaves its tag
    }
}

// ---- per-tile sort: bitonic_sort lives in raster_bin_dev.h (the compositing forward sorts the small tiles itself) ----
// CAP: keys this instantiation sorts in LDS; it handles the tiles with LO < n <= CAP (n > kSortLdsCap: in global memory) and
// leaves the others to the sibling launch.  A single kernel sized for the worst case reserved 64 KB of LDS for every tile
// and fitted two workgroups per CU, while the average tile at 960x540 has ~650 keys.
// head_count (may be null): a device word holding how many entries at the head of tile_order can qualify (tile_scan, hdr[5]); the
// launch is then a small fixed grid whose workgroups stride over that head, instead of one workgroup per tile of which nearly
// all look at their tile's count and leave (2040 workgroups for a handful of large tiles: 11 us at 960x540).
template <int LO, int CAP>
__global__ void __launch_bounds__(256) tile_sort_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ tile_order,
                                                       uint64_t* __restrict__ keys, uint32_t* __restrict__ point_list, uint32_t capacity,
                                                       const uint32_t* __restrict__ head_count, int nt)
{
    __shared__ uint64_t s_keys[CAP];
    const int limit = head_count ? min((int)*head_count, nt) : nt;
    for (int slot = blockIdx.x; slot < limit; slot += gridDim.x) {
        const uint2 r = ranges[tile_order[slot]];            // the launch's tiles, heaviest first (tile_scan)
        uint32_t end = r.y < capacity ? r.y : capacity;
        if (r.x >= end) continue;
        const int n = (int)(end - r.x);
        if (n <= LO || (n > CAP && CAP < kSortLdsCap)) continue;    // the sibling launch's tile
        uint64_t* gk = keys + r.x;
        __syncthreads();                                    // the previous tile's keys have left s_keys
        if (n <= CAP) {
            for (int i = threadIdx.x; i < n; i += 256) s_keys[i] = gk[i];
            __syncthreads();
            if (n > 1) bitonic_sort<true>(s_keys, n, 256, threadIdx.x);
            for (int i = threadIdx.x; i < n; i += 256) point_list[r.x + i] = (uint32_t)s_keys[i];
        } else {
            // oversized bucket: same network directly on the (L2-resident) global bucket
            __syncthreads();
            bitonic_sort<false>(gk, n, 256, threadIdx.x);
            __threadfence_block();
            for (int i = threadIdx.x; i < n; i += 256) point_list[r.x + i] = (uint32_t)gk[i];
        }
    }
}

}  // namespace

int mom_launch_binning_count(const MomRasterArgs* a, const GeomView& g, const ImageView& im, uint32_t* num_rendered_dev,
                             uint32_t* num_rendered_host, bool hist_done, hipStream_t s)
{
    const int gx = (a->W + MOM_TILE - 1) / MOM_TILE, gy = (a->H + MOM_TILE - 1) / MOM_TILE;
    const int tiles = gx * gy;
    int ry0, ry1;
    mom_tile_rows(a, gy, &ry0, &ry1);
    const int cull = a->keep_all_tiles ? 0 : 1;
    // the header and the tile counters were cleared by the projection kernel (raster_api.hip): no fill launch here.  (Folding
    // the scan into the histogram kernel too -- the last workgroup to finish scans -- was measured and dropped: with the
    // device-scope fence the ticket needs, 782 workgroups each wrote the L2 back, 27 -> 72 us.)
    int chunks = (a->P + 256 * 2048 - 1) / (256 * 2048);
    if (chunks < 1) chunks = 1;
    {
        static int forced = -1;           // MOM_BIN_CHUNKS: 256-Gaussian chunks per workgroup (measurement)
        if (forced < 0) { const char* e = getenv("MOM_BIN_CHUNKS"); forced = e ? atoi(e) : 0; }
        if (forced > 0) chunks = forced;
    }
    const int blocks = (a->P + 256 * chunks - 1) / (256 * chunks);
    mom_prof_begin(MOM_P_HIST, s);
    if (hist_done) {
        // the projection kernel counted the instances (raster_preprocess.hip, HIST)
    } else if (tiles <= kMaxLdsTiles)
        hipLaunchKernelGGL(tile_hist_kernel<true>, dim3(blocks), dim3(256), (size_t)tiles * 4, s, a->P, chunks, gx, gy, ry0, ry1, cull,
                           g.rec, im.tile_counts, g.reach);
    else
        hipLaunchKernelGGL(tile_hist_kernel<false>, dim3(blocks), dim3(256), 0, s, a->P, chunks, gx, gy, ry0, ry1, cull, g.rec,
                           im.tile_counts, g.reach);
    mom_prof_end(MOM_P_HIST, s);
    if (hipGetLastError() != hipSuccess) return MOM_ELAUNCH;
    MomProfScope ps(MOM_P_SCAN, s);
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, s, tiles, im.tile_counts, im.tile_cursor, im.ranges, im.hdr,
                       num_rendered_dev, num_rendered_host, gx * ry0, gx * (ry1 - ry0), im.tile_order);
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}

int mom_launch_binning_sort(const MomRasterArgs* a, const GeomView& g, const BinView& b, const ImageView& im, size_t capacity,
                            uint32_t* status_dev, bool render_sorts_small, hipStream_t s)
{
    const int gx = (a->W + MOM_TILE - 1) / MOM_TILE, gy = (a->H + MOM_TILE - 1) / MOM_TILE;
    const int tiles = gx * gy;
    int ry0, ry1;
    mom_tile_rows(a, gy, &ry0, &ry1);
    const int cull = a->keep_all_tiles ? 0 : 1;
    int chunks = (a->P + 256 * 2048 - 1) / (256 * 2048);
    if (chunks < 1) chunks = 1;
    {
        static int forced = -1;           // MOM_BIN_CHUNKS: 256-Gaussian chunks per workgroup (measurement)
        if (forced < 0) { const char* e = getenv("MOM_BIN_CHUNKS"); forced = e ? atoi(e) : 0; }
        if (forced > 0) chunks = forced;
    }
    const int blocks = (a->P + 256 * chunks - 1) / (256 * chunks);
    const uint32_t cap = capacity > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)capacity;
    const uint32_t tag = a->overflow_tag ? a->overflow_tag : 1u;
    mom_prof_begin(MOM_P_SCATTER, s);
    if (tiles <= kMaxLdsTiles)
        hipLaunchKernelGGL(tile_scatter_kernel<true>, dim3(blocks), dim3(256), (size_t)tiles * 4, s, a->P, chunks, gx, gy,
                           ry0, ry1, cull, g.rec, g.reach, im.tile_cursor, b.keys, cap, im.hdr, status_dev, tag);
    else
        hipLaunchKernelGGL(tile_scatter_kernel<false>, dim3(blocks), dim3(256), 0, s, a->P, chunks, gx, gy, ry0, ry1, cull, g.rec,
                           g.reach, im.tile_cursor, b.keys, cap, im.hdr, status_dev, tag);
    mom_prof_end(MOM_P_SCATTER, s);
    if (hipGetLastError() != hipSuccess) return MOM_ELAUNCH;
    MomProfScope ps(MOM_P_SORT, s);
    const int nt = gx * (ry1 - ry0);      // the rows the geometry stage binned: the order covers exactly these tiles
    if (nt == 0) return MOM_OK;
    if (render_sorts_small) {
        // tiles of up to kRenderSortCap keys are sorted by the compositing forward itself (raster_render.hip), in the LDS it stages
        // its splats in afterwards: one launch and one pass over the keys less; this launch takes the rest
        hipLaunchKernelGGL((tile_sort_kernel<kRenderSortCap, kSortLdsCap>), dim3(nt < 128 ? nt : 128), dim3(256), 0, s, im.ranges,
                           im.tile_order, b.keys, b.point_list, cap, im.hdr + 5, nt);
        return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
    }
    hipLaunchKernelGGL((tile_sort_kernel<0, kSortSmallCap>), dim3(nt), dim3(256), 0, s, im.ranges, im.tile_order, b.keys, b.point_list,
                       cap, (const uint32_t*)nullptr, nt);
    if (hipGetLastError() != hipSuccess) return MOM_ELAUNCH;
    hipLaunchKernelGGL((tile_sort_kernel<kSortSmallCap, kSortLdsCap>), dim3(nt), dim3(256), 0, s, im.ranges, im.tile_order, b.keys,
                       b.point_list, cap, (const uint32_t*)nullptr, nt);
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}
