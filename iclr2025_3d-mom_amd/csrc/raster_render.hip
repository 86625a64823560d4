// Tile compositing, forward and backward, gfx950.
//
// Replaces renderCUDA<3> forward (reference forward.cu:261-379) and backward
// (backward.cu:415-590).  One 256-thread workgroup (4 wave64) per 16x16 tile,
// one pixel per lane; a wave owns an 8x8 pixel block (its "strip" below: the footprint was 16x4 until round 6).  Splat records (48 B:
// xy+depth | conic+opacity | rgb) are gathered once per workgroup into LDS in
// chunks of 256 and then read back as LDS broadcasts.
//
// Backward: instead of the reference's 10 global float atomics per (pixel,
// splat) pair, each wave reduces its 64 pixels' contributions in registers
// (a packed DPP reduce-scatter inside the 16-lane rows, row_totals; the four rows
// of four consecutive splats are summed by one transposition, rows_transpose_sum)
// and issues ONE atomic instruction per four (wave, splat) pairs into the
// per-Gaussian accumulator record gacc[P][MOM_GACC_FLOATS].
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
// A wave owns one footprint ("strip") of the tile, and about two thirds of the tile's splats (of those the binning kept: it applies the
// same test to the whole tile) cannot reach alpha >= 1/255 anywhere in a given strip.  While a splat is staged into LDS its
// staging thread decides, per strip, whether any point of the strip's rectangle can reach the splat's power bound
// (mom_rect_reach, mom_common.h).  Each wave then compacts the indices of its reachable splats (ballot + rank) and loops
// over those only.  The exact per-pixel tests still run: results are bit-identical.
// Footprint of a wave inside the 16x16 tile: kFW x kFH pixels, kWX footprints across.  (16x4 strips: 16,4,1; 8x8 blocks: 8,8,2.)
// 8x8 blocks: a square is reached by fewer splats than a 16x4 strip of the same area (same call, 960x540 / 200 k: forward 113.7 ->
// 111.7 us, backward 194.4 -> 191.8; 1 M Gaussians at 1352x1014: 372 -> 360 and 719 -> 694).
constexpr int kFW = 8, kFH = 64 / kFW, kWX = 16 / kFW;
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
#define MOM_FWD_WAVES 8       // (macros, not constexpr: the launch-bounds attributes below take literals)
#define MOM_BWD_WAVES 5
// Optional loss epilogue of the forward kernel (MomRasterArgs.l1_target): target null = none.
struct L1Epilogue {
    const float* target;
    float* grad;
    float* sums;
    float inv_n;
    float* partials;        // null, or [tiles][2]: the tile's two sums are stored here instead of being added to `sums`
};
// Splats staged per round (a multiple of 256; each thread stages kRound / 256 of them).
// The backward stages 512 (297 against 307 us: fewer rounds, each with two barriers and a list build, for tiles that still hold
// ~330 splats on average), the forward 256 (139 against 147 us at 512: it stops early and wastes part of its last round).
constexpr int kRound = 256, kRoundChunks = kRound / 64;
constexpr int kRoundB = 512, kRoundChunksB = kRoundB / 64;
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

// -DFWD_STAMPS (tools/variants.sh; tools/probe/fwd_stamps.py reads them): where a workgroup of render_fwd / render_bwd spends its life.
// Thread 0 of every workgroup leaves s_memtime at its phase boundaries (words 0-7), s_memrealtime at entry and exit (8, 9: one clock
// for the chip), XCC_ID | HW_ID (10) and tile | list length (11).  Compiled out of the shipped library.
#ifdef FWD_STAMPS
constexpr int kStampWords = 16, kStampGroups = 16384;      // words 12-15: cycles each of the four waves spent at the rounds' barriers (render_fwd)
__device__ unsigned long long g_fwd_stamps[kStampGroups * kStampWords];
__device__ unsigned long long g_bwd_stamps[kStampGroups * kStampWords];
// [kernel 0 fwd / 1 bwd][bucket of live (fwd) or valid (bwd) lanes: 0, 1-2, 3-4, 5-8, 9-16, 17-32, 33-64][0: entries walked, 1: entries that pass the wave-level tests]
__device__ unsigned long long g_lane_hist[2][7][2];
__device__ __forceinline__ int lane_bucket(uint64_t m) { const int n = __popcll(m); return n == 0 ? 0 : (n <= 2 ? 1 : (n <= 4 ? 2 : (n <= 8 ? 3 : (n <= 16 ? 4 : (n <= 32 ? 5 : 6))))); }
#ifdef FWD_LANE_HIST      // (with -DFWD_STAMPS: one global atomic per walked entry -- it stretches the launch a hundredfold, so never together with timing stamps)
#define LANE_HIST(kernel, mask, passed) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_lane_hist[kernel][lane_bucket(mask)][passed], 1ull); } while (0)
#else
#define LANE_HIST(kernel, mask, passed) do { } while (0)
#endif
#define FWD_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < kStampGroups) g_fwd_stamps[blockIdx.x * kStampWords + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define BWD_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < kStampGroups) g_bwd_stamps[blockIdx.x * kStampWords + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define STAMP_ID(arr) do { if (threadIdx.x == 0 && blockIdx.x < kStampGroups) { \
        arr[blockIdx.x * kStampWords + 8] = __builtin_amdgcn_s_memrealtime(); \
        arr[blockIdx.x * kStampWords + 10] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492); } } while (0)
#define STAMP_LIST(arr, tile, n) do { if (threadIdx.x == 0 && blockIdx.x < kStampGroups) arr[blockIdx.x * kStampWords + 11] = ((unsigned long long)(uint32_t)(tile) << 32) | (uint32_t)(n); } while (0)
#define STAMP_EXIT(arr) do { if (threadIdx.x == 0 && blockIdx.x < kStampGroups) arr[blockIdx.x * kStampWords + 9] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define STAMP_WAIT() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#define BARRIER_T0() const unsigned long long bw0_ = __builtin_amdgcn_s_memtime()
#define BARRIER_T1() bar_wait_ += __builtin_amdgcn_s_memtime() - bw0_
#else
#define BARRIER_T0() do { } while (0)
#define BARRIER_T1() do { } while (0)
#define FWD_STAMP(k) do { } while (0)
#define BWD_STAMP(k) do { } while (0)
#define STAMP_ID(arr) do { } while (0)
#define STAMP_LIST(arr, tile, n) do { } while (0)
#define STAMP_EXIT(arr) do { } while (0)
#define STAMP_WAIT() do { } while (0)
#define LANE_HIST(kernel, mask, passed) do { } while (0)
#endif

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MOM_FWD_WAVES, MOM_FWD_WAVES)))
render_fwd_kernel(const uint2* __restrict__ ranges, uint32_t* point_list /* read AND, for the tiles sorted here, written: no restrict */, int W, int H, int gx, int nt, int t0, int run,
                  const uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ order_hdr, const float4* __restrict__ rec, const float* __restrict__ bg, float* __restrict__ final_T,
                  uint32_t* __restrict__ n_contrib, float* __restrict__ out_color, float* __restrict__ out_depth,
                  uint32_t capacity, L1Epilogue l1, const uint64_t* __restrict__ sort_keys, uint32_t* __restrict__ tile_walked,
                  unsigned long long* status_post, uint32_t status_serial)
{
    __shared__ float4 s_rec[kRound * 3];
    __shared__ uint8_t s_mask[kRound];
    __shared__ uint16_t s_lists[4][kRound];
    __shared__ __attribute__((aligned(16))) uint32_t s_live[4];
    FWD_STAMP(0);
    STAMP_ID(g_fwd_stamps);
    // t0: first tile of this launch's rows (tile-row shard); the tiles come heaviest first (tile_scan)
    // (the order was made for the rows of the forward's geometry stage, kept in header words 3 and 4; a launch over other rows -- the
    // backward of a tile-row shard that rendered a halo -- falls back to the positional mapping)
    const bool ordered = order_hdr[3] == (uint32_t)t0 && order_hdr[4] == (uint32_t)nt;
    const int tile = ordered ? (int)tile_order[blockIdx.x] : t0 + remap_tile(blockIdx.x, nt, run);
    const int tx = tile % gx, ty = tile / gx;

    uint2 range = ranges[tile];
    if (range.y > capacity) range.y = capacity;
    if (range.x > range.y) range.x = range.y;
    int toDo = (int)(range.y - range.x);
    STAMP_WAIT();
    FWD_STAMP(1);                                             // the tile's range is here
    STAMP_LIST(g_fwd_stamps, tile, toDo);
    const int rounds = (toDo + kRound - 1) / kRound;
    const int list_len = toDo;
    int walked_rounds = rounds;                               // rounds of the list this workgroup walks before every pixel is done

    // The tile's depth sort, for tiles whose keys fit the LDS the rounds below stage their splats in (sort_keys == null: the
    // binning sorted every tile).  The sorted indices go to point_list -- the backward walks it too -- and are read back from
    // there by this workgroup (same CU: its stores are visible to its loads after the barrier).
    static_assert(kRenderSortCap * 8 <= kRound * 3 * 16, "the sort aliases the splat staging area");
    bool have_first = false;                                  // a tile sorted here: the first round's index comes out of LDS
    uint32_t first_id = 0;
    if (sort_keys && toDo > 0 && toDo <= kRenderSortCap) {
        uint64_t* sk = reinterpret_cast<uint64_t*>(s_rec);
        const uint64_t* __restrict__ gk = sort_keys + range.x;
        for (int i = threadIdx.x; i < toDo; i += 256) sk[i] = gk[i];
        __syncthreads();
        if (toDo > 1) bitonic_sort<true>(sk, toDo, 256, (int)threadIdx.x);
        for (int i = threadIdx.x; i < toDo; i += 256) point_list[range.x + i] = (uint32_t)sk[i];
        have_first = true;
        if ((int)threadIdx.x < toDo) first_id = (uint32_t)sk[threadIdx.x];     // (the keys alias s_rec: read before round 0 stages into it)
        __threadfence_block();
        __syncthreads();
    }

    FWD_STAMP(2);                                             // sorted (or nothing to sort)
    // (the pixel's coordinates are formed HERE, behind the sort: the radix-4 passes hold four 64-bit keys per lane, and with the
    // pixel state live across them the kernel left its 64-register budget -- 24 bytes of scratch, tests/test_isa.py)
    const int lx = kFW * ((threadIdx.x >> 6) % kWX) + (threadIdx.x & 63) % kFW;       // wave footprint: see strip_reach_mask
    const int ly = kFH * ((threadIdx.x >> 6) / kWX) + (threadIdx.x & 63) / kFW;
    const int px = tx * MOM_TILE + lx, py = ty * MOM_TILE + ly;
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float T = 1.0f;
    uint32_t last_contributor = 0;
    float C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f;
    uint64_t live = __builtin_amdgcn_ballot_w64(inside);      // wave-uniform: the lanes still compositing (= !done, as a mask)

    // A round's records are asked for one round ahead: the index and the three record loads that depend on it are two memory round
    // trips, and a tile's eight workgroups-per-CU neighbours are all there is to cover them (at the headline size every tile is
    // resident from the start of the launch: nothing new is scheduled onto a SIMD that waits).  The values wait in registers
    // through the compositing of the round before and go to LDS once every wave has left that round.
    static_assert(kRound == 256, "one staged splat per thread");
    float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), p1 = p0, p2 = p0;
    bool pv = false;
    auto fetch = [&](int round) {
        const uint32_t at = range.x + (uint32_t)(round * kRound) + threadIdx.x;
        pv = round < rounds && at < range.y;
        if (pv) {
            const size_t id = (round == 0 && have_first) ? first_id : point_list[at];
            p0 = rec[3 * id + 0];
            p1 = rec[3 * id + 1];
            p2 = rec[3 * id + 2];
        }
    };
    fetch(0);
    STAMP_WAIT();
    FWD_STAMP(3);                                             // round 0's records are in registers
#ifdef FWD_STAMPS
    unsigned long long bar_wait_ = 0;
#endif
    // "every pixel of the tile is done" (forward.cu:305-312) from four wave-uniform flags: each wave leaves `live != 0` in LDS at the
    // end of a round, and the barrier that protects the staging area anyway makes the four flags visible.  (__syncthreads_count
    // compiled into a DPP reduction, an LDS atomic and THREE barriers per round.)
    if (lane == 0) s_live[wv] = live != 0;
    for (int i = 0; i < rounds; i++, toDo -= kRound) {
        { BARRIER_T0(); __syncthreads(); BARRIER_T1(); }
        const uint4 lv = *reinterpret_cast<const uint4*>(s_live);
        if (__builtin_amdgcn_readfirstlane((int)(lv.x | lv.y | lv.z | lv.w)) == 0) { walked_rounds = i; break; }
        {
            uint32_t reach = 0;
            if (pv) {
                float4 q0 = p0;
                q0.w = power_bound(p1.w);
                // (the tile's origin goes through an opaque scalar: left alone the compiler hoists the four strips' rectangle bounds --
                // eight wave-uniform floats -- out of the rounds into vector registers the compositing loop has no room for)
                int ox = tx * MOM_TILE, oy = ty * MOM_TILE;
                asm volatile("" : "+s"(ox), "+s"(oy));
                reach = strip_reach_mask(q0, p1, (float)ox, (float)oy);
                s_rec[threadIdx.x * 3 + 0] = q0;
                s_rec[threadIdx.x * 3 + 1] = p1;
                s_rec[threadIdx.x * 3 + 2] = p2;
            }
            s_mask[threadIdx.x] = (uint8_t)reach;
        }
        { BARRIER_T0(); __syncthreads(); BARRIER_T1(); }
        fetch(i + 1);
        int list[kRoundChunks];
        const int n_w = build_wave_list(s_mask, s_lists[wv], wv, lane, list);
#pragma unroll
        for (int c = 0; c < kRoundChunks; c++) {
            const int nk = __builtin_amdgcn_readfirstlane(min(64, n_w - 64 * c));
            for (int k = 0; k < nk; k++) {
                // The wave-uniform tests below work on lane MASKS built from ballots of single comparisons, which are the
                // comparisons' own results and combine in the scalar unit; __all / __any of a combined bool made the compiler
                // rebuild a mask in the vector unit (v_cndmask + v_cmp) -- six of the forty vector instructions of an iteration
                // went into that and into re-materialising the record's address (the same finding as in render_bwd below).
                if (live == 0) break;                       // every pixel of the strip is saturated (or outside the image)
                const int j = __builtin_amdgcn_readlane(list[c], k);
                uint32_t rec_off = (uint32_t)j * 48u;
                asm volatile("" : "+v"(rec_off));          // the record's LDS address, formed once
                const char* rec_j = reinterpret_cast<const char*>(s_rec) + rec_off;
                const float4 r0 = *reinterpret_cast<const float4*>(rec_j);
                const float4 r1 = *reinterpret_cast<const float4*>(rec_j + 16);
                // the opacity arrives WITH the conic, as one b128: left alone the compiler reads three dwords here and the fourth behind the
                // reject test -- which 98 % of the walked entries pass (tools/probe/lane_hist.py) -- i.e. a second LDS round trip on every
                // entry's dependent chain: 121.4 -> 114.2 us.  (The colour record as well: three more live registers, 16 bytes of scratch, 115.0;
                // with the four list registers packed into one to make room for it: 113.7 against 113.8 -- its round trip is not on the chain.)
                asm volatile("" :: "v"(r1.w));
                float dx, dy;
                const float power = splat_power(r0, r1, pxf, pyf, dx, dy);
                uint64_t vm = live & __builtin_amdgcn_ballot_w64(!(power < r0.w)) & __builtin_amdgcn_ballot_w64(!(power > 0.0f));
                LANE_HIST(0, live, 0);
                if (vm == 0) continue;                      // no lane of the wave can reach 1/255 (power_bound)
                LANE_HIST(0, live, 1);
                const float alpha = fminf(0.99f, r1.w * mom_exp(power));
                const float test_T = T * (1.f - alpha);
                vm &= __builtin_amdgcn_ballot_w64(!(alpha < 1.0f / 255.0f));
                const uint64_t sat = __builtin_amdgcn_ballot_w64(test_T < 0.0001f);
                live &= ~(vm & sat);                        // the lanes this splat would saturate stop here, without it (forward.cu:340-345)
                vm &= ~sat;
                const bool valid = __builtin_amdgcn_inverse_ballot_w64(vm);         // the mask IS the condition: no vector compare
                if (valid) {
                    const float4 r2 = *reinterpret_cast<const float4*>(rec_j + 32);
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
        if (lane == 0) s_live[wv] = live != 0;                // (read behind the next round's first barrier)
    }
    // "entries processed before block exit" (SURVEY 8d's Q = 256 x the sum of this over the tiles): the list is walked in rounds of
    // 256 like the reference's (forward.cu:305-327), and a round is entered unless every pixel of the tile is done.  One plain
    // store per tile into the scatter's cursor array, which nothing reads after the binning.
#ifdef FWD_STAMPS
    if ((threadIdx.x & 63) == 0 && blockIdx.x < kStampGroups) g_fwd_stamps[blockIdx.x * kStampWords + 12 + (threadIdx.x >> 6)] = bar_wait_;
#endif
    FWD_STAMP(4);                                             // wave 0 has left the loop
    if (threadIdx.x == 0) tile_walked[tile] = (uint32_t)min(list_len, walked_rounds * kRound);
    // MomRasterArgs.status_post: the frame's status bits (header word 1, final since the binning) go to a pinned host word with the
    // caller's serial number -- one store by one thread of the launch instead of a copy kernel and a marker behind every frame
    if (status_post && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(status_post, ((unsigned long long)status_serial << 32) | order_hdr[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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
    FWD_STAMP(5);                                             // the pixel's stores are issued
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
            const float t1 = (s_sum[0] + s_sum[2]) + (s_sum[4] + s_sum[6]), t2 = (s_sum[1] + s_sum[3]) + (s_sum[5] + s_sum[7]);
            if (l1.partials) {
                // one plain store per tile: 2040 workgroups adding into ONE line are serialised by the device, ~8 ns each (6 us at the
                // tail of the launch; tools/probe/lds_row_atomic.hip has the same-line rate), and the value is rarely read at all
                reinterpret_cast<float2*>(l1.partials)[tile] = make_float2(t1, t2);
            } else {
                atomicAdd(&l1.sums[0], t1);
                atomicAdd(&l1.sums[1], t2);
            }
        }
    }
    FWD_STAMP(6);                                             // wave 0 at the end of the L1 epilogue (it waited for the other waves' loops there)
    STAMP_WAIT();
    FWD_STAMP(7);                                             // ... with its stores acknowledged
    STAMP_EXIT(g_fwd_stamps);
}

// ---- the backward's reduction, priced by instruction class -----------------------------------------------------------------
// tools/probe/valu_rate.hip (cycles a wave64 instruction holds the SIMD, two or more waves offering work): v_mul / v_add / v_sub /
// v_mov / v_fma with three distinct registers 2.5; DPP adds, v_cmp, v_cndmask with an SGPR-pair mask, v_min / v_max, any operand
// from an SGPR 4.2; v_exp / v_rcp / v_readlane / v_permlane{16,32}_swap 8.2; v_cndmask reading VCC 23.5.  Summed over the loop
// body of round 3 these prices give 416 cycles per executed (wave, splat) pair = the measured 242 us, of which 176 were the
// reduction and its bookkeeping.  What changed against that version, all of it arithmetic-neutral (the same sums in the same
// order within a quad; the order across rows differs):
//  * the two in-quad pair steps use the DPP bank mask instead of two selects and one add: a lane's low two bits ARE its bank,
//    so `v_add_f32_dpp d, x, x quad_perm:[1,0,3,2] bank_mask:0x5` + the same on y with bank_mask:0xa leaves the pair sum of x in
//    the even lanes and of y in the odd ones: 2 instructions instead of 3 (the compiler cannot be made to emit partial-mask DPP
//    adds from builtins, hence the asm block; it begins with the wait states a DPP read needs after a write of EXEC);
//  * the step over bit 2 is a pair step too (two selects, one rotation) instead of an all-reduce of both registers;
//  * the four rows are NOT summed per splat.  The row totals of four consecutive splats wait in four registers and are summed
//    by a 4 x 4 transposition: v_permlane16_swap(m0, m1) + add, the same on (m2, m3), v_permlane32_swap of the two sums + add
//    = 3 swaps and 3 adds per FOUR splats, leaving splat q's totals in row q -- exactly where the one atomic instruction per four
//    splats wants them.  Before: 2 swaps, 2 adds, 2 moves and a select on VCC per splat.
#define MOM_ROR(d, x, y, n, bank) "v_add_f32_dpp " d ", " x ", " y " row_ror:" n " row_mask:0xf bank_mask:" bank "\n"
#define MOM_QP(d, x, y, perm) "v_add_f32_dpp " d ", " x ", " y " quad_perm:[" perm "] row_mask:0xf bank_mask:0xf\n"
// Row totals of N values: afterwards, in every 16-lane row, the lane mom_row_slot() == k holds the row's total of v[k].
// A DPP bank is a QUAD of the row (bank i = lanes 4 i .. 4 i + 3; tools/probe/swap_probe.hip), i.e. bits 2 and 3 of the lane
// index, so the two pair steps that need no select are the ROTATIONS:
//   row_ror:4 (the lane four below has the other value of bit 2):  a_i = pair (v[2i], v[2i+1]); quads 0, 2 (bank mask 0x5) keep and
//              sum the first, quads 1, 3 (0xa) the second; a4 = v8 (+ v9 the same way)
//   row_ror:8 (bit 3):  c0 = pair (a0, a1), c1 = pair (a2, a3): quads 0, 1 (0x3) the first, quads 2, 3 (0xc) the second.
//              Quad q now holds v[q] (c0), v[4 + q] (c1) and v[8] (c2; with ten values v[8 + (q & 1)]), each lane the sum over its
//              column (the four lanes of the row with its index modulo 4)
//   quad_perm [1,0,3,2] (bit 0):  the last pair step, with selects: even lanes keep c0, odd lanes c1
//   quad_perm [2,3,0,1] (bit 1):  an all-reduce
//   select:    lanes with index 2 modulo 4 take c2
// bit0 / sel2: the lanes with bit 0 set / with index 2 modulo 4, as 64-bit masks (loop invariants in scalar registers).
template <int N>
__device__ __forceinline__ float row_totals(const float (&v)[N], uint64_t bit0, uint64_t sel2)
{
    static_assert(N == 9 || N == 10, "nine values (training) or ten (with a depth gradient)");
    float a0, a1, a2, a3, a4, c0, c1, c2, out;
    // s_nop 1: the two wait states a DPP read needs after a vector write of its source (what the compiler's hazard recogniser
    // would insert; it does not look inside an asm statement).  EXEC is only ever written by scalar instructions in front of this
    // block (s_or_b64 at the end of the divergent block), which is no DPP hazard: that one -- five wait states -- is for VECTOR
    // writes of EXEC (v_cmpx), and the kernel has none (tests/test_isa.py checks).  Inside the block every DPP source is written
    // at least two instructions before it is read.
#define MOM_ROW_NOP "s_nop 1\n"
#define MOM_ROW_TAIL \
        MOM_ROR("%5", "%0", "%0", "8", "0x3") MOM_ROR("%6", "%2", "%2", "8", "0x3")                                               \
        MOM_ROR("%5", "%1", "%1", "8", "0xc") MOM_ROR("%6", "%3", "%3", "8", "0xc")                                               \
        MOM_ROR("%7", "%4", "%4", "8", "0xf")                                                                                      \
        "v_cndmask_b32_e64 %1, %6, %5, %9\n"          /* send: odd lanes c0 */                                                    \
        "v_cndmask_b32_e64 %0, %5, %6, %9\n"          /* keep: odd lanes c1 */                                                    \
        MOM_QP("%7", "%7", "%7", "1,0,3,2")                                                                                        \
        MOM_QP("%0", "%1", "%0", "1,0,3,2")                                                                                        \
        "s_nop 0\n"                                                                                                                 \
        MOM_QP("%7", "%7", "%7", "2,3,0,1")                                                                                        \
        MOM_QP("%0", "%0", "%0", "2,3,0,1")                                                                                        \
        "v_cndmask_b32_e64 %8, %0, %7, %10\n"
    if constexpr (N == 10) {
        asm volatile(MOM_ROW_NOP
                     MOM_ROR("%0", "%11", "%11", "4", "0x5") MOM_ROR("%1", "%13", "%13", "4", "0x5") MOM_ROR("%2", "%15", "%15", "4", "0x5")
                     MOM_ROR("%3", "%17", "%17", "4", "0x5") MOM_ROR("%4", "%19", "%19", "4", "0x5")
                     MOM_ROR("%0", "%12", "%12", "4", "0xa") MOM_ROR("%1", "%14", "%14", "4", "0xa") MOM_ROR("%2", "%16", "%16", "4", "0xa")
                     MOM_ROR("%3", "%18", "%18", "4", "0xa") MOM_ROR("%4", "%20", "%20", "4", "0xa")
                     MOM_ROW_TAIL
                     : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(out)
                     : "s"(bit0), "s"(sel2), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]),
                       "v"(v[8]), "v"(v[N - 1]));
    } else {
        asm volatile(MOM_ROW_NOP
                     MOM_ROR("%0", "%11", "%11", "4", "0x5") MOM_ROR("%1", "%13", "%13", "4", "0x5") MOM_ROR("%2", "%15", "%15", "4", "0x5")
                     MOM_ROR("%3", "%17", "%17", "4", "0x5") MOM_ROR("%4", "%19", "%19", "4", "0xf")
                     MOM_ROR("%0", "%12", "%12", "4", "0xa") MOM_ROR("%1", "%14", "%14", "4", "0xa") MOM_ROR("%2", "%16", "%16", "4", "0xa")
                     MOM_ROR("%3", "%18", "%18", "4", "0xa")
                     MOM_ROW_TAIL
                     : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(out)
                     : "s"(bit0), "s"(sel2), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]),
                       "v"(v[8]));
    }
#undef MOM_ROW_TAIL
    return out;
}
// Which of the N values a lane holds after row_totals (>= N: none): quad q = bits 2-3 of the lane index; lanes 0 / 1 / 2 modulo 4
// hold v[q] / v[4 + q] / v[8 + q].
__device__ __forceinline__ int mom_row_slot(int lane)
{
    const int q = (lane >> 2) & 3, r = lane & 3;
    return r == 3 ? 15 : 4 * r + q;
}
// The two halves of the 4 x 4 transposition that sums the rows of four splats (see above): swap_add16(a, b) leaves, in row
// pairs (0,1) and (2,3), the pair's sum of a (even row) and of b (odd row); swap_add32 does the same with the wave's halves.
// (In-place asm forms of the two -- swap, add, and the waiting registers updated by read-modify-write statements only -- were
// tried to lose the two copies per iteration the register allocator makes around them: it made more, 1391 / 1429 vector
// instructions in the kernel against 1374.)
__device__ __forceinline__ float swap_add16(float a, float b)
{
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap_add32(float a, float b)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// DEPTH: a gradient arrives for the depth image too (dL_dpixel_depths != null; never in training).
// The record this kernel leaves per Gaussian (gacc, MOM_GACC_FLOATS = 12 floats, ten used) holds RAW sums: slots 0-1 sum a dx and sum a dy (a = opacity G
// dL/dalpha), i.e. the mean's gradient before the splat's conic matrix, the factors -W/2 and -H/2 of d(pixel)/d(ndc) and the sign
// are applied; slots 2-4 the conic's, without their -1/2.  The projection backward (raster_backward.hip) applies the matrix and
// those constants once per Gaussian instead of this loop once per (pixel, splat) pair.  The record stays linear in dL/dpixel, so
// the ranks of a tile-row shard still sum it (mom_raster_backward_render in include/mom4d.h).
template <bool DEPTH>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MOM_BWD_WAVES, MOM_BWD_WAVES)))   // LDS (31 KB) allows 5 workgroups per CU
render_bwd_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H, int gx, int nt, int t0, int run,
                  const uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ order_hdr, const float4* __restrict__ rec, const float* __restrict__ bg, const float* __restrict__ final_Ts,
                  const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpixels,
                  const float* __restrict__ dL_dpixel_depths, float* __restrict__ gacc, uint32_t capacity)
{
    __shared__ float4 s_rec[kRoundB * 3];
    __shared__ uint32_t s_id[kRoundB];
    __shared__ uint8_t s_mask[kRoundB];
    __shared__ uint16_t s_lists[4][kRoundB];
    BWD_STAMP(0);
    STAMP_ID(g_bwd_stamps);
    // t0: first tile of this launch's rows (tile-row shard); the tiles come heaviest first (tile_scan)
    // (the order was made for the rows of the forward's geometry stage, kept in header words 3 and 4; a launch over other rows -- the
    // backward of a tile-row shard that rendered a halo -- falls back to the positional mapping)
    const bool ordered = order_hdr[3] == (uint32_t)t0 && order_hdr[4] == (uint32_t)nt;
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
    constexpr int kVals = DEPTH ? 10 : 9;
    // row totals of up to four splats wait for the transposition that sums their rows (rows_transpose_sum); their positions in
    // the round's staging area ride in two scalars, sixteen bits each
    // The transposition is taken in steps as the splats arrive, so that two registers hold what waits: h0 the row totals of an
    // even splat until its odd partner comes (swap_add16), h1 the first pair's sums until the second pair's come (swap_add32).
    // The count is wave-uniform: the branches below are scalar.  (Four registers indexed by the count became four selects on
    // VCC -- 23.5 cycles each -- or a web of moves at every merge.)
    float h0 = 0.f, h1 = 0.f;
    uint64_t jp = 0;                  // wave-uniform: the four splats' positions in the staging area, sixteen bits each
    int npend = 0;                    // wave-uniform
    const uint32_t row_shift = (lane & 16);            // 0 / 16: which half of the word holds this row's splat
    const uint64_t kBit0 = 0xAAAAAAAAAAAAAAAAull, kSel2 = 0x4444444444444444ull;     // lanes with bit 0 set / with index 2 modulo 4
    const int vslot = mom_row_slot(lane);              // which of the kVals values this lane holds after row_totals
    const uint64_t m_inside = __builtin_amdgcn_ballot_w64(inside);
    // n_real: how many of the four rows hold a splat (the last group of a round is padded with zeros)
    auto push = [&](float v, int j, int n_real) {
        jp = (jp & ~(0xFFFFull << (npend << 4))) | ((uint64_t)(uint32_t)j << (npend << 4));
        if (!(npend & 1)) {
            h0 = v;
            npend++;
        } else {
            const float t = swap_add16(h0, v);
            if (npend == 1) {
                h1 = t;
                npend = 2;
            } else {
                const float tot = swap_add32(h1, t);                   // row q: the totals of the group's splat q
                const uint32_t jq = (((lane & 32) ? (uint32_t)(jp >> 32) : (uint32_t)jp) >> row_shift) & 0xFFFFu;     // two 32-bit halves: no 64-bit vector shift
                if (vslot < kVals && (lane >> 4) < n_real) atomicAdd(&gacc[(size_t)s_id[jq] * MOM_GACC_FLOATS + vslot], tot);
                npend = 0;
            }
        }
    };
    auto drain = [&]() {
        const int n_real = npend;
        while (npend) push(0.f, 0, n_real);
    };
    const float bg_dot_dpixel = bg[0] * dp0 + bg[1] * dp1 + bg[2] * dp2;
    // splats behind the last contributor of every pixel of this wave need no work at all
    int wave_last = last_contributor;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wave_last = max(wave_last, __shfl_xor(wave_last, d));
    wave_last = __builtin_amdgcn_readfirstlane(wave_last);
    // ... and the splats behind the last contributor of the whole TILE are not even staged: the list ends there for this launch
    // (positions are counted from the front, so cutting the tail moves none of them).  A tile the forward left early -- every
    // pixel saturated -- otherwise stages, tests and skips the rest of its list round after round.
    {
        int* s_last = reinterpret_cast<int*>(s_id);
        if (lane == 0) s_last[wv] = wave_last;
        __syncthreads();
        const int tile_last = max(max(s_last[0], s_last[1]), max(s_last[2], s_last[3]));
        __syncthreads();                                   // s_id is written again by the first round's staging
        if (tile_last < toDo) {
            toDo = tile_last;
            range.y = range.x + (uint32_t)tile_last;
            contributor = (uint32_t)toDo;
        }
    }
    const int rounds = (toDo + kRoundB - 1) / kRoundB;
    BWD_STAMP(1);                                             // pixel state loaded, the list cut at the tile's last contributor
    STAMP_LIST(g_bwd_stamps, tile, toDo);

    // (Asking for the next round's records a round ahead, as render_fwd does, does not pay here: 26 more live registers spill at
    // five waves per SIMD -- 201.8 against 195.3 us; with 256-splat rounds 211; the indices alone 196.6.)
    for (int i = 0; i < rounds; i++, toDo -= kRoundB) {
        drain();                     // s_id is about to be overwritten: what waits goes out now
        __syncthreads();
#pragma unroll
        for (int sl = 0; sl < kRoundB / 256; sl++) {
            const int slot = threadIdx.x + 256 * sl;
            uint32_t reach = 0;
            const int progress = i * kRoundB + slot;
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
          const int nk = __builtin_amdgcn_readfirstlane(min(64, n_w - 64 * c));      // a scalar loop bound (left to itself the compiler compared in the vector unit)
          for (int k = 0; k < nk; k++) {
            const int j = __builtin_amdgcn_readlane(list[c], k);
            contributor = (uint32_t)(toDo - j - 1);        // position of splat j in the tile's list, counted from 0
            if ((int)contributor >= wave_last) continue;   // wave-uniform: occluded for all 64 pixels
            // the record's LDS address, formed ONCE in a vector register (the compiler otherwise re-materialises it from the
            // scalar in front of every read: three moves per iteration on a kernel bound by instruction issue)
            uint32_t rec_off = (uint32_t)j * 48u;
            asm volatile("" : "+v"(rec_off));
            const char* rec_j = reinterpret_cast<const char*>(s_rec) + rec_off;
            const float4 r0 = *reinterpret_cast<const float4*>(rec_j);
            const float4 r1 = *reinterpret_cast<const float4*>(rec_j + 16);
            // both records as ONE b128 each, complete before the exponent: left alone the compiler fetches them in four pieces and the
            // opacity behind the reject test, a second LDS round trip per pair (198.7 -> 197.0 us, one call, three rounds)
            asm volatile("" :: "v"(r0.x), "v"(r0.y), "v"(r0.z), "v"(r0.w), "v"(r1.x), "v"(r1.y), "v"(r1.z), "v"(r1.w));
            // ... and the colour record with them (the register budget at five waves per SIMD has room: 193.8 -> 192.5 us)
            const float4 r2 = *reinterpret_cast<const float4*>(rec_j + 32);
            asm volatile("" :: "v"(r2.x), "v"(r2.y), "v"(r2.z));
            float dx, dy;
            const float power = splat_power(r0, r1, pxf, pyf, dx, dy);
            if (!__any(!(power < r0.w))) continue;         // no lane of the wave can reach 1/255 (power_bound)
            const float G = mom_exp(power);
            const float alpha = fminf(0.99f, r1.w * G);
            // wave-uniform skip.  A ballot of each comparison is the comparison's own lane mask, and the masks combine in the
            // scalar unit; a ballot (or __any) of the combined bool made the compiler rebuild a mask with v_cndmask + v_cmp.
            const uint64_t vmask = m_inside & __builtin_amdgcn_ballot_w64((int)contributor < last_contributor) &
                                   __builtin_amdgcn_ballot_w64(!(power > 0.0f)) & __builtin_amdgcn_ballot_w64(!(alpha < 1.0f / 255.0f));
            LANE_HIST(1, vmask, 0);
            if (vmask == 0) continue;
            LANE_HIST(1, vmask, 1);

            // Inside the divergent block: what only the contributing lanes may do (their recurrences) and the two factors every
            // gradient carries, w = alpha T and a = opacity G dL/dalpha.  The products with them are formed outside, by all lanes
            // (an instruction costs the same whatever EXEC is), so that three registers need a zero for the other lanes, not nine.
            // no divergent block (207 against 211 us with one, same box, alternating): with alpha and G forced to zero in the lanes that do not contribute, the recurrences below are
            // identities there (T / (1 - 0), 0 c + 1 accum) and both gradient factors vanish
            const float alpha_in = alpha, G_in = G;
            float w, a, g_op;
            {
                // (asm: the selects take the lane mask the skip test above already holds in a scalar pair.  The compiler's own form
                // reads VCC -- 23.5 cycles in tools/probe/valu_rate.hip's stream of nothing but such selects, though in this loop
                // the two forms measured the same, 204.5 against 205 us)
                float alpha, G;
                asm("v_cndmask_b32_e64 %0, 0, %2, %4\n\tv_cndmask_b32_e64 %1, 0, %3, %4" : "=&v"(alpha), "=v"(G) : "v"(alpha_in), "v"(G_in), "s"(vmask));
                // the reference divides (T = T / (1 - alpha), backward.cu:502); the hardware reciprocal is within 1 ulp of that
                // quotient and costs one instruction instead of ten
                const float one_m_alpha = 1.f - alpha;
                const float inv_1ma = __builtin_amdgcn_rcpf(one_m_alpha);      // shared by T and the background term below
                T = T * inv_1ma;
                w = alpha * T;
                // accum_k: the colour composited behind this splat (backward.cu:525-541 carries last_alpha and last_color into
                // the next iteration and blends there; blending at the end of this one is the same arithmetic without the
                // four copies per iteration)
                float dL_dalpha = (r2.x - accum0) * dp0;
                dL_dalpha += (r2.y - accum1) * dp1;
                dL_dalpha += (r2.z - accum2) * dp2;
                // (spelled as scale-then-fma so that the update happens in the accumulator's own register: no copy)
                accum0 = __builtin_fmaf(alpha, r2.x, one_m_alpha * accum0);
                accum1 = __builtin_fmaf(alpha, r2.y, one_m_alpha * accum1);
                accum2 = __builtin_fmaf(alpha, r2.z, one_m_alpha * accum2);
                if (DEPTH) {
                    dL_dalpha += (r0.z - accum_d) * dpd;
                    accum_d = __builtin_fmaf(alpha, r0.z, one_m_alpha * accum_d);
                }
                dL_dalpha *= T;
                dL_dalpha += (-T_final * inv_1ma) * bg_dot_dpixel;
                // no derivative for the 0.99 cap, exactly as the reference (backward.cu:571).  With a = dL/dG * G:
                //   dL/d mean   = -(W/2, H/2) * a * (conic (dx, dy))      dL/d conic = -1/2 * a * (dx^2, dx dy, dy^2)
                // (backward.cu:573-586; the constant factors wait for the projection backward)
                g_op = G * dL_dalpha;
                a = r1.w * g_op;
            }
            const float g_c0 = w * dp0, g_c1 = w * dp1, g_c2 = w * dp2, g_d = DEPTH ? w * dpd : 0.f;
            // the mean's gradient is conic (ax, ay): the conic belongs to the splat, not the pixel, so the sums of ax and ay go
            // into the record and the projection backward applies the matrix once per Gaussian (four instructions less here)
            const float ax = a * dx, ay = a * dy;
            const float g_mx = ax, g_my = ay;
            const float g_cx = ax * dx, g_cy = ax * dy, g_cw = ay * dy;
            // (Letting the lanes of a splat that reaches only one or two pixels of the strip add their nine values themselves --
            // nine one-lane atomic instructions instead of the reduction and one nine-lane instruction -- was measured: 284 / 298 us
            // for thresholds 1 / 2 against 274.  An atomic INSTRUCTION costs the CU more than the reduction.)
            float v;
            if (DEPTH) {
                const float gv[10] = {g_mx, g_my, g_cx, g_cy, g_cw, g_op, g_c0, g_c1, g_c2, g_d};
                v = row_totals<10>(gv, kBit0, kSel2);
            } else {
                const float gv[9] = {g_mx, g_my, g_cx, g_cy, g_cw, g_op, g_c0, g_c1, g_c2};
                v = row_totals<9>(gv, kBit0, kSel2);
            }
            // every 16-lane row now holds ITS totals (lane k of a row: value k); the rows are summed four splats at a time
            push(v, j, 4);
          }
        }
    }
    BWD_STAMP(2);                                             // wave 0 has left the loop
    drain();
    BWD_STAMP(3);
    STAMP_WAIT();
    BWD_STAMP(4);                                             // its atomics acknowledged
    STAMP_EXIT(g_bwd_stamps);
}

// A launch over no tiles (a tile-row shard whose rows are empty) still owes the caller the frame's status word: whoever polls
// MomRasterArgs.status_post for this serial number would otherwise spin until its timeout.
__global__ void status_post_kernel(const uint32_t* __restrict__ order_hdr, unsigned long long* status_post, uint32_t status_serial)
{
    __hip_atomic_store(status_post, ((unsigned long long)status_serial << 32) | order_hdr[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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
    if (nt == 0) {
        if (a->status_post) hipLaunchKernelGGL(status_post_kernel, dim3(1), dim3(1), 0, s, im.hdr, a->status_post, a->status_serial);
        return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
    }
    const float l1_inv_n = 1.0f / (3.0f * (float)a->W * (float)a->H);
    L1Epilogue l1 = {a->l1_target, a->l1_grad, a->l1_sums, a->l1_grad_scale != 0.f ? l1_inv_n * a->l1_grad_scale : l1_inv_n, a->l1_partials};
    if (!l1.grad || (!l1.sums && !l1.partials)) l1.target = nullptr;
    hipLaunchKernelGGL(render_fwd_kernel, dim3(nt), dim3(256), 0, s, im.ranges, b.point_list, a->W, a->H, gx, nt, gx * ry0, tile_run(gx), im.tile_order, im.hdr,
                       g.rec, a->background, a->forward_only ? nullptr : im.final_T, a->forward_only ? nullptr : im.n_contrib, out_color,
                       out_depth, cap, l1, sort_small ? b.keys : nullptr, im.tile_cursor, a->status_post, a->status_serial);
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}

int mom_launch_render_bwd(const MomRasterArgs* a, const GeomView& g, const BinView& b, const ImageView& im, size_t capacity,
                          const float* dL_dpix, const float* dL_ddepth, hipStream_t s)
{
    const int gx = (a->W + MOM_TILE - 1) / MOM_TILE, gy = (a->H + MOM_TILE - 1) / MOM_TILE;
    const uint32_t cap = capacity > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)capacity;
    if (!a->accum_cleared && hipMemsetAsync(g.gacc, 0, (size_t)a->P * MOM_GACC_FLOATS * 4, s) != hipSuccess) return MOM_ELAUNCH;
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

#ifdef FWD_STAMPS
extern "C" int mom_debug_fwd_stamps(unsigned long long* host_dst, int groups)
{
    if (groups > kStampGroups) groups = kStampGroups;
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_fwd_stamps), (size_t)groups * kStampWords * 8) == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}
extern "C" int mom_debug_lane_hist(unsigned long long* host_dst, int reset)
{
    if (hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_lane_hist), sizeof(g_lane_hist)) != hipSuccess) return MOM_ELAUNCH;
    if (reset) { static unsigned long long zero[2 * 7 * 2] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_lane_hist), zero, sizeof(zero)) != hipSuccess) return MOM_ELAUNCH; }
    return MOM_OK;
}
extern "C" int mom_debug_bwd_stamps(unsigned long long* host_dst, int groups)
{
    if (groups > kStampGroups) groups = kStampGroups;
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_bwd_stamps), (size_t)groups * kStampWords * 8) == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}
#endif

// wave_sum self test: out[w] = sum of in[64*w .. 64*w+63]
namespace {
__global__ void wave_sum_test_kernel(const float* in, float* out)
{
    const float t = wave_sum(in[blockIdx.x * 64 + threadIdx.x]);
    if (threadIdx.x == 17) out[blockIdx.x] = t;
}
// The compositing backward's reduction, exactly as the kernel composes it: in [waves][4 splats][N values][64 lanes] ->
// out [waves][4][N] = the sums over the 64 lanes.
template <int N>
__global__ void row_reduce_test_kernel(const float* in, float* out)
{
    const int lane = threadIdx.x;
    const float* base = in + (size_t)blockIdx.x * 4 * N * 64;
    float rows[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        float v[N];
#pragma unroll
        for (int k = 0; k < N; k++) v[k] = base[(q * N + k) * 64 + lane];
        rows[q] = row_totals<N>(v, 0xAAAAAAAAAAAAAAAAull, 0x4444444444444444ull);
    }
    const float tot = swap_add32(swap_add16(rows[0], rows[1]), swap_add16(rows[2], rows[3]));
    const int slot = mom_row_slot(lane);
    if (slot < N) out[((size_t)blockIdx.x * 4 + (lane >> 4)) * N + slot] = tot;
}
}  // namespace
extern "C" int mom_selftest_row_reduce(const float* in, float* out, int waves, int nvals, mom_stream_t s)
{
    if (nvals == 9) hipLaunchKernelGGL(row_reduce_test_kernel<9>, dim3(waves), dim3(64), 0, (hipStream_t)s, in, out);
    else if (nvals == 10) hipLaunchKernelGGL(row_reduce_test_kernel<10>, dim3(waves), dim3(64), 0, (hipStream_t)s, in, out);
    else return MOM_EINVAL;
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}
extern "C" int mom_selftest_wave_sum(const float* in, float* out, int waves, mom_stream_t s)
{
    hipLaunchKernelGGL(wave_sum_test_kernel, dim3(waves), dim3(64), 0, (hipStream_t)s, in, out);
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}
