// Tile compositing, forward and backward, gfx950.
//
// Replaces renderCUDA<3> forward (reference forward.cu:261-379) and backward
// (backward.cu:415-590).  One 256-thread workgroup (4 wave64) per 16x16 tile,
// one pixel per lane; a wave owns a 16x4 pixel strip.  Splat records (48 B:
// xy+depth | conic+opacity | rgb) are gathered once per workgroup into LDS in
// chunks of 256 and then read back as LDS broadcasts.
//
// Backward: instead of the reference's 10 global float atomics per (pixel,
// splat) pair, each wave reduces its 64 pixels' contributions in registers
// (a packed DPP reduce-scatter, reduce_scatter10) and issues ONE 40-byte atomic
// instruction per (wave, splat) into the per-Gaussian accumulator record
// gacc[P][12].
#include "mom_common.h"
#include "raster_bin_dev.h"

namespace {

__device__ __forceinline__ int remap_tile(int b, int nt, int C)
{
    // Workgroups are dealt round-robin over the 8 XCDs (XCD = b & 7).  Runs of C consecutive tiles go to the XCDs in turn:
    // neighbouring tiles, which share splats, mostly share an L2, and every XCD gets every part of the image -- with one
    // contiguous band of tiles per XCD the dense middle of the frame kept two XCDs busy long after the others had finished
    // (960x540: compositing forward 161 -> 138 us, backward 377 -> 333 us at C = a quarter of a tile row; C = 1: 145 / 336,
    // C = a whole row: 151 / 347).  Speed only; bijective for any nt.
    const int full = nt / (8 * C) * (8 * C);
    if (b >= full) return b;
    const int xcd = b & 7, slot = b >> 3, q = slot / C;
    return (q * 8 + xcd) * C + (slot - q * C);
}
static inline int tile_run(int gx) { return gx >= 8 ? gx / 4 : 1; }
#ifndef MOM_TILE_ORDER
#define MOM_TILE_ORDER 1
#endif

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v)
{
    const int o = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
    return v + __int_as_float(o);
}
// Sum over the 64 lanes; the total is returned in every lane (via v_readlane of lane 63).
__device__ __forceinline__ float wave_sum(float v)
{
    v = dpp_add<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
    v = dpp_add<0x124, 0xF>(v);  // row_ror:4
    v = dpp_add<0x128, 0xF>(v);  // row_ror:8   -> every lane holds its 16-lane row sum
    v = dpp_add<0x142, 0xA>(v);  // row_bcast:15 into rows 1,3
    v = dpp_add<0x143, 0xC>(v);  // row_bcast:31 into rows 2,3 -> row 3 holds the wave sum
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Pair step of a reduce-scatter: lanes whose `upper` bit is clear end up with the pair-sum of x, the others with the
// pair-sum of y (the partner lane, CTRL, differs in exactly that bit).
template <int CTRL>
__device__ __forceinline__ float dpp_pair(float x, float y, bool upper)
{
    const float keep = upper ? y : x, send = upper ? x : y;
    return keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), CTRL, 0xF, 0xF, false));
}
// Sums ten per-lane values over the 64 lanes of the wave; lane k (k < 10) returns the total of v[k] (the other
// lanes return totals of some component too -- callers only use lanes 0..9).  About half the instructions of ten
// wave_sum calls: two scatter steps inside each quad pack the ten values into three registers (a lane's low two bits
// pick its component), two row rotations sum those over the 16-lane rows, a lane select merges the three registers,
// and gfx950's v_permlane16_swap / v_permlane32_swap add the four rows.
__device__ __forceinline__ float reduce_scatter10(const float (&v)[10], int lane)
{
    const bool b0 = lane & 1, b1 = lane & 2;
    const float a0 = dpp_pair<0xB1>(v[0], v[1], b0);     // quad_perm [1,0,3,2]
    const float a1 = dpp_pair<0xB1>(v[2], v[3], b0);
    const float a2 = dpp_pair<0xB1>(v[4], v[5], b0);
    const float a3 = dpp_pair<0xB1>(v[6], v[7], b0);
    const float a4 = dpp_pair<0xB1>(v[8], v[9], b0);
    float c0 = dpp_pair<0x4E>(a0, a1, b1);               // quad_perm [2,3,0,1]: lane l holds v[l & 3] over its quad
    float c1 = dpp_pair<0x4E>(a2, a3, b1);               //                      v[4 + (l & 3)]
    float c2 = dpp_add<0x4E, 0xF>(a4);                   //                      v[8 + (l & 1)]
    c0 = dpp_add<0x124, 0xF>(c0); c1 = dpp_add<0x124, 0xF>(c1); c2 = dpp_add<0x124, 0xF>(c2);   // row_ror:4
    c0 = dpp_add<0x128, 0xF>(c0); c1 = dpp_add<0x128, 0xF>(c1); c2 = dpp_add<0x128, 0xF>(c2);   // row_ror:8
    const int k = lane & 15;
    float m = k < 4 ? c0 : (k < 8 ? c1 : c2);            // lane k of every row: row total of v[k]
    {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
        m = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
        m = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    return m;
}

// The staging thread stores mom_power_bound(opacity) (mom_common.h) in the LDS copy of the record (r0.w, whose tile count
// the compositing kernels do not use).  The loops then skip a splat for the whole wave with `!__any(!(power < bound))`
// before paying for exp; every pair that survives still takes the exact tests, so results do not change.
__device__ __forceinline__ float power_bound(float opacity) { return mom_power_bound(opacity); }

// Exponent of the splat's Gaussian at a pixel.  render_fwd and render_bwd must round it identically, or a pair sitting on
// the 1/255 threshold could be composited by one pass and not by the other; with contraction left to the compiler the
// same source expression was fused differently from one kernel (and one edit) to the next.  So: contraction off, the
// fusions spelled out.
__device__ __forceinline__ float splat_power(const float4 r0, const float4 r1, float pxf, float pyf, float& dx, float& dy)
{
#pragma clang fp contract(off)
    dx = r0.x - pxf;
    dy = r0.y - pyf;
    const float ax = r1.x * dx, cy = r1.z * dy, bx = r1.y * dx;
    const float quad = __builtin_fmaf(cy, dy, ax * dx);          // conic.x dx^2 + conic.z dy^2
    return __builtin_fmaf(-0.5f, quad, -(bx * dy));             // -0.5 quad - conic.y dx dy
}

// ---- per-wave splat lists ------------------------------------------------------------------------------------------
// A wave owns a 16x4 strip of the tile, and about two thirds of the tile's splats (of those the binning kept: it applies the
// same test to the whole tile) cannot reach alpha >= 1/255 anywhere in a given strip.  While a splat is staged into LDS its
// staging thread decides, per strip, whether any point of the strip's rectangle can reach the splat's power bound
// (mom_rect_reach, mom_common.h).  Each wave then compacts the indices of its reachable splats (ballot + rank) and loops
// over those only.  The exact per-pixel tests still run: results are bit-identical.
// Footprint of a wave inside the 16x16 tile: kFW x kFH pixels, kWX footprints across.  (16x4 strips: 16,4,1; 8x8 blocks: 8,8,2.)
#ifndef MOM_FOOT_W
#define MOM_FOOT_W 16
#endif
constexpr int kFW = MOM_FOOT_W, kFH = 64 / kFW, kWX = 16 / kFW;
__device__ __forceinline__ uint32_t strip_reach_mask(const float4 r0, const float4 r1, float x0, float y0)
{
    const float a = r1.x, c = r1.z;
    if (!(a > 0.f) || !(c > 0.f)) return 0xFu;
    const float inv_a = __builtin_amdgcn_rcpf(a), inv_c = __builtin_amdgcn_rcpf(c);     // two reciprocals serve all 16 edges
    uint32_t m = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const float xa = x0 + (float)(kFW * (w % kWX)), ya = y0 + (float)(kFH * (w / kWX));
        m |= mom_rect_reach(r0.x, r0.y, a, r1.y, c, r0.w, inv_a, inv_c, xa, xa + (float)(kFW - 1), ya, ya + (float)(kFH - 1)) ? (1u << w) : 0u;
    }
    return m;
}
// Both compositing kernels gain from occupancy more than they lose to a tighter register budget (forward: 140 us at 8 waves
// per SIMD against 168 us at 6, same instructions): the budget is pinned instead of left to the allocator, which moved it
// by a wave or two from one unrelated edit to the next.
#ifndef MOM_FWD_WAVES
#define MOM_FWD_WAVES 8
#endif
#ifndef MOM_BWD_WAVES
#define MOM_BWD_WAVES 5
#endif
#ifndef MOM_BWD_MIN
#define MOM_BWD_MIN 5
#endif
// Optional loss epilogue of the forward kernel (MomRasterArgs.l1_target): target null = none.
struct L1Epilogue {
    const float* target;
    float* grad;
    float* sums;
    float inv_n;
};
// Splats staged per round (a multiple of 256; each thread stages kRound / 256 of them).
// The backward stages 512 (297 against 307 us: fewer rounds, each with two barriers and a list build, for tiles that still hold
// ~330 splats on average), the forward 256 (139 against 147 us at 512: it stops early and wastes part of its last round).
#ifndef MOM_ROUND
#define MOM_ROUND 256
#endif
#ifndef MOM_ROUND_BWD
#define MOM_ROUND_BWD 512
#endif
constexpr int kRound = MOM_ROUND, kRoundChunks = kRound / 64;
constexpr int kRoundB = MOM_ROUND_BWD, kRoundChunksB = kRoundB / 64;
// Compacts, for wave `wv`, the indices j < kRound whose mask has bit wv: afterwards lane k of list[c] holds entry 64 c + k of
// the wave's list (in increasing j, so the compositing order is kept); returns the list length.
template <int CHUNKS>
__device__ __forceinline__ int build_wave_list(const uint8_t* s_mask, uint16_t* s_list, int wv, int lane, int (&list)[CHUNKS])
{
    int n = 0;
#pragma unroll
    for (int c = 0; c < CHUNKS; c++) {
        const int j = 64 * c + lane;
        const bool bit = (s_mask[j] >> wv) & 1;
        const uint64_t bal = __ballot(bit);
        const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if (bit) s_list[n + rank] = (uint16_t)j;
        n += __popcll(bal);
    }
#pragma unroll
    for (int c = 0; c < CHUNKS; c++) list[c] = s_list[64 * c + lane];
    return n;
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MOM_FWD_WAVES, MOM_FWD_WAVES)))
render_fwd_kernel(const uint2* __restrict__ ranges, uint32_t* point_list /* read AND, for the tiles sorted here, written: no restrict */, int W, int H, int gx, int nt, int t0, int run,
                  const uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ order_hdr, const float4* __restrict__ rec, const float* __restrict__ bg, float* __restrict__ final_T,
                  uint32_t* __restrict__ n_contrib, float* __restrict__ out_color, float* __restrict__ out_depth,
                  uint32_t capacity, L1Epilogue l1, const uint64_t* __restrict__ sort_keys)
{
    __shared__ float4 s_rec[kRound * 3];
    __shared__ uint8_t s_mask[kRound];
    __shared__ uint16_t s_lists[4][kRound];
    // t0: first tile of this launch's rows (tile-row shard); the tiles come heaviest first (tile_scan)
    // (the order was made for the rows of the forward's geometry stage, kept in header words 3 and 4; a launch over other rows -- the
    // backward of a tile-row shard that rendered a halo -- falls back to the positional mapping)
    const bool ordered = MOM_TILE_ORDER && order_hdr[3] == (uint32_t)t0 && order_hdr[4] == (uint32_t)nt;
    const int tile = ordered ? (int)tile_order[blockIdx.x] : t0 + remap_tile(blockIdx.x, nt, run);
    const int tx = tile % gx, ty = tile / gx;
    const int lx = kFW * ((threadIdx.x >> 6) % kWX) + (threadIdx.x & 63) % kFW;       // wave footprint: see strip_reach_mask
    const int ly = kFH * ((threadIdx.x >> 6) / kWX) + (threadIdx.x & 63) / kFW;
    const int px = tx * MOM_TILE + lx, py = ty * MOM_TILE + ly;
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    bool done = !inside;

    uint2 range = ranges[tile];
    if (range.y > capacity) range.y = capacity;
    if (range.x > range.y) range.x = range.y;
    int toDo = (int)(range.y - range.x);
    const int rounds = (toDo + kRound - 1) / kRound;

    // The tile's depth sort, for tiles whose keys fit the LDS the rounds below stage their splats in (sort_keys == null: the
    // binning sorted every tile).  The sorted indices go to point_list -- the backward walks it too -- and are read back from
    // there by this workgroup (same CU: its stores are visible to its loads after the barrier).
    static_assert(kRenderSortCap * 8 <= kRound * 3 * 16, "the sort aliases the splat staging area");
    if (sort_keys && toDo > 0 && toDo <= kRenderSortCap) {
        uint64_t* sk = reinterpret_cast<uint64_t*>(s_rec);
        const uint64_t* __restrict__ gk = sort_keys + range.x;
        for (int i = threadIdx.x; i < toDo; i += 256) sk[i] = gk[i];
        __syncthreads();
        if (toDo > 1) bitonic_sort<true>(sk, toDo, 256, (int)threadIdx.x);
        for (int i = threadIdx.x; i < toDo; i += 256) point_list[range.x + i] = (uint32_t)sk[i];
        __threadfence_block();
        __syncthreads();
    }

    float T = 1.0f;
    uint32_t last_contributor = 0;
    float C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f;

    for (int i = 0; i < rounds; i++, toDo -= kRound) {
        if (__syncthreads_count(done) == 256) break;
#pragma unroll
        for (int sl = 0; sl < kRound / 256; sl++) {
            const int slot = threadIdx.x + 256 * sl, progress = i * kRound + slot;
            uint32_t reach = 0;
            if (range.x + progress < range.y) {
                const size_t id = point_list[range.x + progress];
                float4 q0 = rec[3 * id + 0];
                const float4 q1 = rec[3 * id + 1];
                q0.w = power_bound(q1.w);
                reach = strip_reach_mask(q0, q1, (float)(tx * MOM_TILE), (float)(ty * MOM_TILE));
                s_rec[slot * 3 + 0] = q0;
                s_rec[slot * 3 + 1] = q1;
                s_rec[slot * 3 + 2] = rec[3 * id + 2];
            }
            s_mask[slot] = (uint8_t)reach;                  // slots past the end of the list: unreachable
        }
        __syncthreads();
        int list[kRoundChunks];
        const int n_w = build_wave_list(s_mask, s_lists[wv], wv, lane, list);
#pragma unroll
        for (int c = 0; c < kRoundChunks; c++) {
            const int nk = min(64, n_w - 64 * c);
            for (int k = 0; k < nk; k++) {
                if (__all(done)) break;
                const int j = __builtin_amdgcn_readlane(list[c], k);
                const float4 r0 = s_rec[j * 3 + 0];
                const float4 r1 = s_rec[j * 3 + 1];
                float dx, dy;
                const float power = splat_power(r0, r1, pxf, pyf, dx, dy);
                bool valid = !done && !(power < r0.w) && !(power > 0.0f);
                if (!__any(valid)) continue;                // no lane of the wave can reach 1/255 (power_bound)
                const float alpha = fminf(0.99f, r1.w * mom_exp(power));
                valid = valid && !(alpha < 1.0f / 255.0f);
                const float test_T = T * (1.f - alpha);
                if (valid && test_T < 0.0001f) {
                    done = true;
                    valid = false;
                }
                if (valid) {
                    const float4 r2 = s_rec[j * 3 + 2];
                    const float w = alpha * T;
                    C0 += r2.x * w;
                    C1 += r2.y * w;
                    C2 += r2.z * w;
                    D += r0.z * w;
                    T = test_T;
                    last_contributor = (uint32_t)(i * kRound + j + 1);  // position in the tile's list, counted from 1
                }
            }
        }
    }
    if (inside) {
        const int pix = py * W + px;
        if (final_T) final_T[pix] = T;                      // null in forward-only rendering: nothing will read them
        if (n_contrib) n_contrib[pix] = last_contributor;
        const size_t HW = (size_t)H * W;
        C0 += T * bg[0];
        C1 += T * bg[1];
        C2 += T * bg[2];
        out_color[pix] = C0;
        out_color[HW + pix] = C1;
        out_color[2 * HW + pix] = C2;
        out_depth[pix] = D;
    }
    if (l1.target) {
        // L1 loss against a target image, its sums and its gradient, while the pixel is still in registers: what
        // mom_l1_loss_acc computes from the stored image (same expressions; the sums' order of addition differs)
        float a1 = 0.f, a2 = 0.f;
        if (inside) {
            const int pix = py * W + px;
            const size_t HW = (size_t)H * W;
            const float c[3] = {C0, C1, C2};
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float d = c[k] - l1.target[k * HW + pix];
                a1 += fabsf(d);
                a2 += d * d;
                l1.grad[k * HW + pix] = d > 0.f ? l1.inv_n : (d < 0.f ? -l1.inv_n : 0.f);
            }
        }
        a1 = wave_sum(a1);
        a2 = wave_sum(a2);
        float* s_sum = reinterpret_cast<float*>(s_lists);     // the lists are dead
        __syncthreads();
        if (lane == 0) { s_sum[2 * wv] = a1; s_sum[2 * wv + 1] = a2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicAdd(&l1.sums[0], (s_sum[0] + s_sum[2]) + (s_sum[4] + s_sum[6]));
            atomicAdd(&l1.sums[1], (s_sum[1] + s_sum[3]) + (s_sum[5] + s_sum[7]));
        }
    }
}

#ifndef MOM_BWD_PEND
#define MOM_BWD_PEND 4                // splats per atomic instruction (1..4)
#endif
// The same for nine values (no depth gradient: training): v[8] needs no pair step.
__device__ __forceinline__ float reduce_scatter9(const float (&v)[9], int lane)
{
    const bool b0 = lane & 1, b1 = lane & 2;
    const float a0 = dpp_pair<0xB1>(v[0], v[1], b0);
    const float a1 = dpp_pair<0xB1>(v[2], v[3], b0);
    const float a2 = dpp_pair<0xB1>(v[4], v[5], b0);
    const float a3 = dpp_pair<0xB1>(v[6], v[7], b0);
    const float a4 = dpp_add<0xB1, 0xF>(v[8]);
    float c0 = dpp_pair<0x4E>(a0, a1, b1);
    float c1 = dpp_pair<0x4E>(a2, a3, b1);
    float c2 = dpp_add<0x4E, 0xF>(a4);                   // every lane of the quad: v[8] over the quad
    c0 = dpp_add<0x124, 0xF>(c0); c1 = dpp_add<0x124, 0xF>(c1); c2 = dpp_add<0x124, 0xF>(c2);
    c0 = dpp_add<0x128, 0xF>(c0); c1 = dpp_add<0x128, 0xF>(c1); c2 = dpp_add<0x128, 0xF>(c2);
    const int k = lane & 15;
    float m = k < 4 ? c0 : (k < 8 ? c1 : c2);
    {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
        m = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
        m = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    return m;
}

// DEPTH: a gradient arrives for the depth image too (dL_dpixel_depths != null; never in training).
// The record this kernel leaves per Gaussian (gacc, 12 floats) holds RAW sums: slots 0-1 the mean's, without the factors
// -W/2 and -H/2 of d(pixel)/d(ndc) and the sign; slots 2-4 the conic's, without their -1/2.  The projection backward
// (raster_backward.hip) applies those constants once per Gaussian instead of this loop once per (pixel, splat) pair.
template <bool DEPTH>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MOM_BWD_MIN, MOM_BWD_WAVES)))   // LDS (31 KB) allows 5 workgroups per CU
render_bwd_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H, int gx, int nt, int t0, int run,
                  const uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ order_hdr, const float4* __restrict__ rec, const float* __restrict__ bg, const float* __restrict__ final_Ts,
                  const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpixels,
                  const float* __restrict__ dL_dpixel_depths, float* __restrict__ gacc, uint32_t capacity)
{
    __shared__ float4 s_rec[kRoundB * 3];
    __shared__ uint32_t s_id[kRoundB];
    __shared__ uint8_t s_mask[kRoundB];
    __shared__ uint16_t s_lists[4][kRoundB];
    // t0: first tile of this launch's rows (tile-row shard); the tiles come heaviest first (tile_scan)
    // (the order was made for the rows of the forward's geometry stage, kept in header words 3 and 4; a launch over other rows -- the
    // backward of a tile-row shard that rendered a halo -- falls back to the positional mapping)
    const bool ordered = MOM_TILE_ORDER && order_hdr[3] == (uint32_t)t0 && order_hdr[4] == (uint32_t)nt;
    const int tile = ordered ? (int)tile_order[blockIdx.x] : t0 + remap_tile(blockIdx.x, nt, run);
    const int tx = tile % gx, ty = tile / gx;
    const int lx = kFW * ((threadIdx.x >> 6) % kWX) + (threadIdx.x & 63) % kFW;       // wave footprint: see strip_reach_mask
    const int ly = kFH * ((threadIdx.x >> 6) / kWX) + (threadIdx.x & 63) / kFW;
    const int px = tx * MOM_TILE + lx, py = ty * MOM_TILE + ly;
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;

    uint2 range = ranges[tile];
    if (range.y > capacity) range.y = capacity;
    if (range.x > range.y) range.x = range.y;
    int toDo = (int)(range.y - range.x);
    const int rounds = (toDo + kRoundB - 1) / kRoundB;

    const int pix = inside ? py * W + px : 0;
    const size_t HW = (size_t)H * W;
    const float T_final = inside ? final_Ts[pix] : 0.f;
    float T = T_final;
    uint32_t contributor = (uint32_t)toDo;
    const int last_contributor = inside ? (int)n_contrib[pix] : 0;

    float accum0 = 0.f, accum1 = 0.f, accum2 = 0.f, accum_d = 0.f;
    float dp0 = 0.f, dp1 = 0.f, dp2 = 0.f, dpd = 0.f;
    if (inside) {
        dp0 = dL_dpixels[pix];
        dp1 = dL_dpixels[HW + pix];
        dp2 = dL_dpixels[2 * HW + pix];
        dpd = DEPTH ? dL_dpixel_depths[pix] : 0.f;
    }
    float last_alpha = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f, last_depth = 0.f;
    constexpr int kVals = DEPTH ? 10 : 9;
    float pend = 0.f;                 // reduced records waiting for their atomic: row q of the wave holds the q-th
    uint32_t pend_id = 0;
    int npend = 0;                    // wave-uniform
    const float bg_dot_dpixel = bg[0] * dp0 + bg[1] * dp1 + bg[2] * dp2;
    // splats behind the last contributor of every pixel of this wave need no work at all
    int wave_last = last_contributor;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wave_last = max(wave_last, __shfl_xor(wave_last, d));
    wave_last = __builtin_amdgcn_readfirstlane(wave_last);

    for (int i = 0; i < rounds; i++, toDo -= kRoundB) {
        __syncthreads();
#pragma unroll
        for (int sl = 0; sl < kRoundB / 256; sl++) {
            const int slot = threadIdx.x + 256 * sl, progress = i * kRoundB + slot;
            uint32_t reach = 0;
            if (range.x + progress < range.y) {
                const uint32_t id = point_list[range.y - progress - 1];
                s_id[slot] = id;
                float4 q0 = rec[3 * (size_t)id + 0];
                const float4 q1 = rec[3 * (size_t)id + 1];
                q0.w = power_bound(q1.w);
                reach = strip_reach_mask(q0, q1, (float)(tx * MOM_TILE), (float)(ty * MOM_TILE));
                s_rec[slot * 3 + 0] = q0;
                s_rec[slot * 3 + 1] = q1;
                s_rec[slot * 3 + 2] = rec[3 * (size_t)id + 2];
            }
            s_mask[slot] = (uint8_t)reach;                  // slots past the end of the list: unreachable
        }
        __syncthreads();
        // this wave's splats of the round, still back to front (render_fwd explains the lists)
        int list[kRoundChunksB];
        const int n_w = build_wave_list(s_mask, s_lists[wv], wv, lane, list);
#pragma unroll
        for (int c = 0; c < kRoundChunksB; c++) {
          const int nk = min(64, n_w - 64 * c);
          for (int k = 0; k < nk; k++) {
            const int j = __builtin_amdgcn_readlane(list[c], k);
            contributor = (uint32_t)(toDo - j - 1);        // position of splat j in the tile's list, counted from 0
            if ((int)contributor >= wave_last) continue;   // wave-uniform: occluded for all 64 pixels
            const float4 r0 = s_rec[j * 3 + 0];
            const float4 r1 = s_rec[j * 3 + 1];
            float dx, dy;
            const float power = splat_power(r0, r1, pxf, pyf, dx, dy);
            if (!__any(!(power < r0.w))) continue;         // no lane of the wave can reach 1/255 (power_bound)
            const float G = mom_exp(power);
            const float alpha = fminf(0.99f, r1.w * G);
            const bool valid = inside && ((int)contributor < last_contributor) && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
            if (!__any(valid)) continue;  // wave-uniform skip

            float g_mx = 0.f, g_my = 0.f, g_cx = 0.f, g_cy = 0.f, g_cw = 0.f, g_op = 0.f, g_c0 = 0.f, g_c1 = 0.f, g_c2 = 0.f,
                  g_d = 0.f;
            if (valid) {
                const float4 r2 = s_rec[j * 3 + 2];
                // the reference divides (T = T / (1 - alpha), backward.cu:502); the hardware reciprocal is within 1 ulp of that
                // quotient and costs one instruction instead of ten
                const float inv_1ma = __builtin_amdgcn_rcpf(1.f - alpha);      // shared by T and the background term below
                T = T * inv_1ma;
                const float w = alpha * T;
                float dL_dalpha = 0.f;
                accum0 = last_alpha * lc0 + (1.f - last_alpha) * accum0;
                lc0 = r2.x;
                dL_dalpha += (r2.x - accum0) * dp0;
                accum1 = last_alpha * lc1 + (1.f - last_alpha) * accum1;
                lc1 = r2.y;
                dL_dalpha += (r2.y - accum1) * dp1;
                accum2 = last_alpha * lc2 + (1.f - last_alpha) * accum2;
                lc2 = r2.z;
                dL_dalpha += (r2.z - accum2) * dp2;
                g_c0 = w * dp0;
                g_c1 = w * dp1;
                g_c2 = w * dp2;
                if (DEPTH) {
                    accum_d = last_alpha * last_depth + (1.f - last_alpha) * accum_d;
                    last_depth = r0.z;
                    dL_dalpha += (r0.z - accum_d) * dpd;
                    g_d = w * dpd;
                }
                dL_dalpha *= T;
                last_alpha = alpha;
                dL_dalpha += (-T_final * inv_1ma) * bg_dot_dpixel;
                // no derivative for the 0.99 cap, exactly as the reference (backward.cu:571).  With a = dL/dG * G:
                //   dL/d mean   = -(W/2, H/2) * a * (conic (dx, dy))      dL/d conic = -1/2 * a * (dx^2, dx dy, dy^2)
                // (backward.cu:573-586; the constant factors wait for the projection backward)
                g_op = G * dL_dalpha;
                const float a = r1.w * g_op;
                const float ax = a * dx, ay = a * dy;
                g_mx = ax * r1.x + ay * r1.y;
                g_my = ay * r1.z + ax * r1.y;
                g_cx = ax * dx;
                g_cy = ax * dy;
                g_cw = ay * dy;
            }
            // (Letting the lanes of a splat that reaches only one or two pixels of the strip add their nine values themselves --
            // nine one-lane atomic instructions instead of the reduction and one nine-lane instruction -- was measured: 284 / 298 us
            // for thresholds 1 / 2 against 274.  An atomic INSTRUCTION costs the CU more than the 33 vector instructions.)
            float v;
            if (DEPTH) {
                const float gv[10] = {g_mx, g_my, g_cx, g_cy, g_cw, g_op, g_c0, g_c1, g_c2, g_d};
                v = reduce_scatter10(gv, lane);
            } else {
                const float gv[9] = {g_mx, g_my, g_cx, g_cy, g_cw, g_op, g_c0, g_c1, g_c2};
                v = reduce_scatter9(gv, lane);
            }
            // Every 16-lane row now holds the totals (lane k of a row: value k).  Row q keeps them for the q-th splat since the
            // last flush, and ONE atomic instruction adds four splats' records.  Measured 272 against 274 us with one instruction
            // per (wave, splat) pair, the step the same: the atomic instruction rate is not this kernel's limit either (it is
            // bound by vector-instruction issue); kept because it is never slower and quarters the atomic instructions.
            const bool mine = (lane >> 4) == npend;
            pend = mine ? v : pend;
            pend_id = mine ? s_id[j] : pend_id;
            if (++npend == MOM_BWD_PEND) {
                if ((lane & 15) < kVals && (lane >> 4) < MOM_BWD_PEND) atomicAdd(&gacc[(size_t)pend_id * 12 + (lane & 15)], pend);
                npend = 0;
            }
          }
        }
    }
    if (npend && (lane & 15) < kVals && (lane >> 4) < npend) atomicAdd(&gacc[(size_t)pend_id * 12 + (lane & 15)], pend);
}

}  // namespace

int mom_launch_render_fwd(const MomRasterArgs* a, const GeomView& g, const BinView& b, const ImageView& im, size_t capacity,
                          float* out_color, float* out_depth, bool sort_small, hipStream_t s)
{
    const int gx = (a->W + MOM_TILE - 1) / MOM_TILE, gy = (a->H + MOM_TILE - 1) / MOM_TILE;
    const uint32_t cap = capacity > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)capacity;
    MomProfScope ps(MOM_P_RENDER_FWD, s);
    int ry0, ry1;
    mom_tile_rows(a, gy, &ry0, &ry1);
    const int nt = gx * (ry1 - ry0);
    if (nt == 0) return MOM_OK;
    L1Epilogue l1 = {a->l1_target, a->l1_grad, a->l1_sums, 1.0f / (3.0f * (float)a->W * (float)a->H)};
    if (!l1.grad || !l1.sums) l1.target = nullptr;
    hipLaunchKernelGGL(render_fwd_kernel, dim3(nt), dim3(256), 0, s, im.ranges, b.point_list, a->W, a->H, gx, nt, gx * ry0, tile_run(gx), im.tile_order, im.hdr,
                       g.rec, a->background, a->forward_only ? nullptr : im.final_T, a->forward_only ? nullptr : im.n_contrib, out_color,
                       out_depth, cap, l1, sort_small ? b.keys : nullptr);
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}

int mom_launch_render_bwd(const MomRasterArgs* a, const GeomView& g, const BinView& b, const ImageView& im, size_t capacity,
                          const float* dL_dpix, const float* dL_ddepth, hipStream_t s)
{
    const int gx = (a->W + MOM_TILE - 1) / MOM_TILE, gy = (a->H + MOM_TILE - 1) / MOM_TILE;
    const uint32_t cap = capacity > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)capacity;
    if (!a->accum_cleared && hipMemsetAsync(g.gacc, 0, (size_t)a->P * 48, s) != hipSuccess) return MOM_ELAUNCH;
    MomProfScope ps(MOM_P_RENDER_BWD, s);
    int ry0, ry1;
    mom_tile_rows(a, gy, &ry0, &ry1);
    const int nt = gx * (ry1 - ry0);
    if (nt == 0) return MOM_OK;
    if (dL_ddepth)
        hipLaunchKernelGGL(render_bwd_kernel<true>, dim3(nt), dim3(256), 0, s, im.ranges, b.point_list, a->W, a->H, gx, nt, gx * ry0, tile_run(gx), im.tile_order, im.hdr,
                           g.rec, a->background, im.final_T, im.n_contrib, dL_dpix, dL_ddepth, g.gacc, cap);
    else
        hipLaunchKernelGGL(render_bwd_kernel<false>, dim3(nt), dim3(256), 0, s, im.ranges, b.point_list, a->W, a->H, gx, nt, gx * ry0, tile_run(gx), im.tile_order, im.hdr,
                           g.rec, a->background, im.final_T, im.n_contrib, dL_dpix, dL_ddepth, g.gacc, cap);
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}

// wave_sum self test: out[w] = sum of in[64*w .. 64*w+63]
namespace {
__global__ void wave_sum_test_kernel(const float* in, float* out)
{
    const float t = wave_sum(in[blockIdx.x * 64 + threadIdx.x]);
    if (threadIdx.x == 17) out[blockIdx.x] = t;
}
}  // namespace
extern "C" int mom_selftest_wave_sum(const float* in, float* out, int waves, mom_stream_t s)
{
    hipLaunchKernelGGL(wave_sum_test_kernel, dim3(waves), dim3(64), 0, (hipStream_t)s, in, out);
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}
