// The row-list experiment of round 5: compositing kernels in which every 16-lane DPP row of a wave is a 4x4 pixel
// block with a splat list of its own.  Included by raster_render.hip inside its anonymous namespace; compiled in only with
// -DMOM_BWD_ROWS=1 / -DMOM_FWD_ROWS=1 (tools/variants.sh), both OFF in the shipped library:
//   backward: the loop is faster (147 us against 190 with the accumulation compiled out, -DMOM_ROWS_NOATOMIC) but it emits 2.4 x the
//             accumulator rows, and the device accepts about 20 G atomic 64-byte lines per second whatever the number of lanes or compute
//             units (tools/probe/lds_row_atomic.hip): 340 us with the 48-byte records the library had when this was measured, 228 us with
//             64-byte ones (-DMOM_GACC_FLOATS=16);
//             LDS float atomics cost 2.7 cycles per active lane, so a tile-level pre-reduction in LDS is no way out either;
//   forward:  bit-identical images, 142 us against 135: the shorter loop (-12 us) does not pay for sixteen lists per tile.
// ---- the backward with one splat list per 16-lane ROW -------------------------------------------------------------------------------
// (MOM_BWD_ROWS)  A wave still owns a 16x4 strip, but its four DPP rows are the strip's four 4x4 pixel BLOCKS and every row walks a
// list of its own: one wave iteration serves one entry of each row's list, so a strip needs max(|L0| .. |L3|) iterations instead of
// |L0 u .. u L3| (tools/cull_stats.py: 0.64 of the entries walked by the strip lists at config 2), and a row's lanes are the pixels
// closest to each other, where a splat either reaches most of them or none.  What was wave-uniform -- the entry, its position in the
// tile's list, the record's address -- becomes a per-lane value (a row that has run out reads a record that cannot contribute: zero
// opacity, bound +inf); the reduction stays inside the rows (row_totals), so the 4 x 4 transposition across rows and its
// bookkeeping disappear and every iteration issues its own atomic instruction, one accumulator row per (row, splat) pair.
#ifndef MOM_BWD_ROWS
#define MOM_BWD_ROWS 0
#endif
#ifndef MOM_ROUND_ROWS
#define MOM_ROUND_ROWS 256
#endif
constexpr int kRoundR = MOM_ROUND_ROWS, kRoundChunksR = kRoundR / 64;
// bit 4 by + bx: can the splat reach the 4x4 block (bx, by) of the tile?
#ifndef MOM_BLOCK_REACH_EXACT
#define MOM_BLOCK_REACH_EXACT 0
#endif
__device__ __forceinline__ uint32_t block_reach_mask(const float4 r0, const float4 r1, float x0, float y0)
{
    const float a = r1.x, c = r1.z;
    if (!(a > 0.f) || !(c > 0.f)) return 0xFFFFu;
    const float inv_a = __builtin_amdgcn_rcpf(a), inv_c = __builtin_amdgcn_rcpf(c);
    uint32_t m = 0;
#if MOM_BLOCK_REACH_EXACT
#pragma unroll
    for (int b = 0; b < 16; b++) {
        const float xa = x0 + (float)(4 * (b & 3)), ya = y0 + (float)(4 * (b >> 2));
        m |= mom_rect_reach(r0.x, r0.y, a, r1.y, c, r0.w, inv_a, inv_c, xa, xa + 3.f, ya, ya + 3.f) ? (1u << b) : 0u;
    }
#else
    // a block is kept if the splat reaches both the 16x4 strip and the 4x16 column it lies in: eight rectangle tests instead of
    // sixteen; what it keeps beyond the exact test (thin diagonal splats) only costs the row an idle entry
    uint32_t rows = 0, cols = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float ya = y0 + (float)(4 * k), xa = x0 + (float)(4 * k);
        rows |= mom_rect_reach(r0.x, r0.y, a, r1.y, c, r0.w, inv_a, inv_c, x0, x0 + 15.f, ya, ya + 3.f) ? (0xFu << (4 * k)) : 0u;
        cols |= mom_rect_reach(r0.x, r0.y, a, r1.y, c, r0.w, inv_a, inv_c, xa, xa + 3.f, y0, y0 + 15.f) ? (0x1111u << k) : 0u;
    }
    m = rows & cols;
#endif
    return m;
}


// The forward with one list per 16-lane row (MOM_FWD_ROWS): same mapping and lists as render_bwd_rows_kernel.  Every pixel still sees
// exactly the splats that pass its own tests, in the list's order, so the image, final_T and n_contrib are bit-identical.  The lists hold
// 8-bit slot numbers (a round stages 256 splats): an entry read past the end of a row's list is a valid slot number whatever it is, and
// the row's lanes are taken out of the iteration's mask instead (act), so nothing is selected or padded.
#ifndef MOM_FWD_ROWS
#define MOM_FWD_ROWS 0
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MOM_FWD_WAVES, MOM_FWD_WAVES)))
render_fwd_rows_kernel(const uint2* __restrict__ ranges, uint32_t* point_list, int W, int H, int gx, int nt, int t0, int run,
                       const uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ order_hdr, const float4* __restrict__ rec, const float* __restrict__ bg, float* __restrict__ final_T,
                       uint32_t* __restrict__ n_contrib, float* __restrict__ out_color, float* __restrict__ out_depth,
                       uint32_t capacity, L1Epilogue l1, const uint64_t* __restrict__ sort_keys, uint32_t* __restrict__ tile_walked,
                       unsigned long long* status_post, uint32_t status_serial)
{
    static_assert(kRound == 256, "8-bit slot numbers; one staged splat per thread");
    __shared__ float4 s_rec[kRound * 3];
    __shared__ uint16_t s_mask[kRound];
    __shared__ uint8_t s_lists[16][kRound + 4];
    const bool ordered = MOM_TILE_ORDER && order_hdr[3] == (uint32_t)t0 && order_hdr[4] == (uint32_t)nt;
    const int tile = ordered ? (int)tile_order[blockIdx.x] : t0 + remap_tile(blockIdx.x, nt, run);
    const int tx = tile % gx, ty = tile / gx;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, row = lane >> 4;
    const int lx = 4 * row + (lane & 3), ly = 4 * wv + ((lane >> 2) & 3);       // row r of wave w: the 4x4 block (r, w) of the tile
    const int px = tx * MOM_TILE + lx, py = ty * MOM_TILE + ly;
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;

    uint2 range = ranges[tile];
    if (range.y > capacity) range.y = capacity;
    if (range.x > range.y) range.x = range.y;
    int toDo = (int)(range.y - range.x);
    const int rounds = (toDo + kRound - 1) / kRound;
    const int list_len = toDo;
    int walked_rounds = rounds;

    bool have_first = false;
    uint32_t first_id = 0;
    if (sort_keys && toDo > 0 && toDo <= kRenderSortCap) {
        uint64_t* sk = reinterpret_cast<uint64_t*>(s_rec);
        const uint64_t* __restrict__ gk = sort_keys + range.x;
        for (int i = threadIdx.x; i < toDo; i += 256) sk[i] = gk[i];
        __syncthreads();
        if (toDo > 1) bitonic_sort<true>(sk, toDo, 256, (int)threadIdx.x);
        for (int i = threadIdx.x; i < toDo; i += 256) point_list[range.x + i] = (uint32_t)sk[i];
        have_first = true;
        if ((int)threadIdx.x < toDo) first_id = (uint32_t)sk[threadIdx.x];
        __threadfence_block();
        __syncthreads();
    }

    float T = 1.0f;
    uint32_t last_contributor = 0;
    float C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f;
    uint64_t live = __builtin_amdgcn_ballot_w64(inside);

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
    const uint8_t* my_list = s_lists[4 * wv + row];
    for (int i = 0; i < rounds; i++, toDo -= kRound) {
        if (__syncthreads_count(!__builtin_amdgcn_inverse_ballot_w64(live)) == 256) { walked_rounds = i; break; }
        {
            uint32_t reach = 0;
            if (pv) {
                float4 q0 = p0;
                q0.w = power_bound(p1.w);
                reach = block_reach_mask(q0, p1, (float)(tx * MOM_TILE), (float)(ty * MOM_TILE));
                s_rec[threadIdx.x * 3 + 0] = q0;
                s_rec[threadIdx.x * 3 + 1] = p1;
                s_rec[threadIdx.x * 3 + 2] = p2;
            }
            s_mask[threadIdx.x] = (uint16_t)reach;
        }
        __syncthreads();
        fetch(i + 1);
        // the lists of this wave's four rows; a row whose sixteen pixels are all done gets none
        int n[4] = {0, 0, 0, 0};
        uint32_t rows_live = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) rows_live |= ((live >> (16 * r)) & 0xFFFFull) ? (1u << r) : 0u;
#pragma unroll
        for (int c = 0; c < kRoundChunks; c++) {
            const int j = 64 * c + lane;
            const uint32_t m = ((uint32_t)s_mask[j] >> (4 * wv)) & rows_live;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const bool bit = (m >> r) & 1u;
                const uint64_t bal = __ballot(bit);
                const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                if (bit) s_lists[4 * wv + r][n[r] + rank] = (uint8_t)j;
                n[r] += __popcll(bal);
            }
        }
        const int my_n = row == 0 ? n[0] : row == 1 ? n[1] : row == 2 ? n[2] : n[3];
        const int n_it = __builtin_amdgcn_readfirstlane(max(max(n[0], n[1]), max(n[2], n[3])));
        int jn = my_list[0];
        for (int k = 0; k < n_it; k++) {
            if (live == 0) break;
            const int j = jn;
            jn = my_list[k + 1];                             // the next entry is asked for now: its round trip hides behind this iteration
            const uint64_t act = __builtin_amdgcn_ballot_w64(k < my_n);
            uint32_t rec_off = (uint32_t)j * 48u;
            asm volatile("" : "+v"(rec_off));
            const char* rec_j = reinterpret_cast<const char*>(s_rec) + rec_off;
            const float4 r0 = *reinterpret_cast<const float4*>(rec_j);
            const float4 r1 = *reinterpret_cast<const float4*>(rec_j + 16);
            float dx, dy;
            const float power = splat_power(r0, r1, pxf, pyf, dx, dy);
            uint64_t vm = live & act & __builtin_amdgcn_ballot_w64(!(power < r0.w)) & __builtin_amdgcn_ballot_w64(!(power > 0.0f));
            if (vm == 0) continue;
            const float alpha = fminf(0.99f, r1.w * mom_exp(power));
            const float test_T = T * (1.f - alpha);
            vm &= __builtin_amdgcn_ballot_w64(!(alpha < 1.0f / 255.0f));
            const uint64_t sat = __builtin_amdgcn_ballot_w64(test_T < 0.0001f);
            live &= ~(vm & sat);
            vm &= ~sat;
            const bool valid = __builtin_amdgcn_inverse_ballot_w64(vm);
            if (valid) {
                const float4 r2 = *reinterpret_cast<const float4*>(rec_j + 32);
                const float w = alpha * T;
                C0 += r2.x * w;
                C1 += r2.y * w;
                C2 += r2.z * w;
                D += r0.z * w;
                T = test_T;
                last_contributor = (uint32_t)(i * kRound + j + 1);
            }
        }
    }
    if (threadIdx.x == 0) tile_walked[tile] = (uint32_t)min(list_len, walked_rounds * kRound);
    if (status_post && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(status_post, ((unsigned long long)status_serial << 32) | order_hdr[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (inside) {
        const int pix = py * W + px;
        if (final_T) final_T[pix] = T;
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
        float* s_sum = reinterpret_cast<float*>(s_mask);      // the masks are dead
        __syncthreads();
        if (lane == 0) { s_sum[2 * wv] = a1; s_sum[2 * wv + 1] = a2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const float t1 = (s_sum[0] + s_sum[2]) + (s_sum[4] + s_sum[6]), t2 = (s_sum[1] + s_sum[3]) + (s_sum[5] + s_sum[7]);
            if (l1.partials) {
                reinterpret_cast<float2*>(l1.partials)[tile] = make_float2(t1, t2);
            } else {
                atomicAdd(&l1.sums[0], t1);
                atomicAdd(&l1.sums[1], t2);
            }
        }
    }
}

template <bool DEPTH>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MOM_BWD_MIN, MOM_BWD_WAVES)))
render_bwd_rows_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, int W, int H, int gx, int nt, int t0, int run,
                       const uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ order_hdr, const float4* __restrict__ rec, const float* __restrict__ bg, const float* __restrict__ final_Ts,
                       const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpixels,
                       const float* __restrict__ dL_dpixel_depths, float* __restrict__ gacc, uint32_t capacity)
{
    __shared__ float4 s_rec[(kRoundR + 1) * 3];              // slot kRoundR: the record of a row that has run out of entries
    __shared__ uint16_t s_mask[kRoundR];
    __shared__ uint16_t s_lists[16][kRoundR];
    const bool ordered = MOM_TILE_ORDER && order_hdr[3] == (uint32_t)t0 && order_hdr[4] == (uint32_t)nt;
    const int tile = ordered ? (int)tile_order[blockIdx.x] : t0 + remap_tile(blockIdx.x, nt, run);
    const int tx = tile % gx, ty = tile / gx;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, row = lane >> 4;
    const int lx = 4 * row + (lane & 3), ly = 4 * wv + ((lane >> 2) & 3);       // row r of wave w: block (r, w) of the tile
    const int px = tx * MOM_TILE + lx, py = ty * MOM_TILE + ly;
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;

    uint2 range = ranges[tile];
    if (range.y > capacity) range.y = capacity;
    if (range.x > range.y) range.x = range.y;
    int toDo = (int)(range.y - range.x);

    const int pix = inside ? py * W + px : 0;
    const size_t HW = (size_t)H * W;
    const float T_final = inside ? final_Ts[pix] : 0.f;
    float T = T_final;
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
    const uint64_t kBit0 = 0xAAAAAAAAAAAAAAAAull, kSel2 = 0x4444444444444444ull;
    const int vslot = mom_row_slot(lane);
    const uint64_t m_inside = __builtin_amdgcn_ballot_w64(inside);
    const float bg_dot_dpixel = bg[0] * dp0 + bg[1] * dp1 + bg[2] * dp2;
    // where each ROW's list ends: the last contributor of its sixteen pixels
    int row_last = last_contributor;
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) row_last = max(row_last, __shfl_xor(row_last, d));
    int rl[4];
#pragma unroll
    for (int r = 0; r < 4; r++) rl[r] = __builtin_amdgcn_readlane(row_last, 16 * r);
    {
        // ... and the splats behind the last contributor of the whole tile are not even staged
        int* s_last = reinterpret_cast<int*>(s_mask);
        if (lane == 0) s_last[wv] = max(max(rl[0], rl[1]), max(rl[2], rl[3]));
        if (threadIdx.x == 0) {
            s_rec[kRoundR * 3 + 0] = make_float4(0.f, 0.f, 0.f, __builtin_inff());     // bound +inf: no pixel reaches it
            s_rec[kRoundR * 3 + 1] = make_float4(0.f, 0.f, 0.f, 0.f);                 // opacity 0
            s_rec[kRoundR * 3 + 2] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        const int tile_last = max(max(s_last[0], s_last[1]), max(s_last[2], s_last[3]));
        __syncthreads();
        if (tile_last < toDo) {
            toDo = tile_last;
            range.y = range.x + (uint32_t)tile_last;
        }
    }
    const int rounds = (toDo + kRoundR - 1) / kRoundR;
    const uint16_t* my_list = s_lists[4 * wv + row];

    for (int i = 0; i < rounds; i++, toDo -= kRoundR) {
        __syncthreads();
#pragma unroll
        for (int sl = 0; sl < kRoundR / 256; sl++) {
            const int slot = threadIdx.x + 256 * sl;
            uint32_t reach = 0;
            const int progress = i * kRoundR + slot;
            if (range.x + progress < range.y) {
                const uint32_t id = point_list[range.y - progress - 1];
                float4 q0 = rec[3 * (size_t)id + 0];
                const float4 q1 = rec[3 * (size_t)id + 1];
                float4 q2 = rec[3 * (size_t)id + 2];
                q0.w = power_bound(q1.w);
                q2.w = __uint_as_float(id);                  // the Gaussian's index rides in the record (the radius is not used here)
                reach = block_reach_mask(q0, q1, (float)(tx * MOM_TILE), (float)(ty * MOM_TILE));
                s_rec[slot * 3 + 0] = q0;
                s_rec[slot * 3 + 1] = q1;
                s_rec[slot * 3 + 2] = q2;
            }
            s_mask[slot] = (uint16_t)reach;
        }
        __syncthreads();
        // the four lists of this wave's rows, back to front, each cut at its row's last contributor
        int n[4] = {0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < kRoundChunksR; c++) {
            const int j = 64 * c + lane;
            const uint32_t m = (uint32_t)s_mask[j] >> (4 * wv);
            const int pos = toDo - j - 1;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const bool bit = ((m >> r) & 1u) && pos < rl[r];
                const uint64_t bal = __ballot(bit);
                const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                if (bit) s_lists[4 * wv + r][n[r] + rank] = (uint16_t)j;
                n[r] += __popcll(bal);
            }
        }
        const int my_n = row == 0 ? n[0] : row == 1 ? n[1] : row == 2 ? n[2] : n[3];
#ifdef MOM_ROWS_NOLOOP
        const int n_it = __builtin_amdgcn_readfirstlane(max(max(n[0], n[1]), max(n[2], n[3]))) > 100000 ? 1 : 0;
#else
        const int n_it = __builtin_amdgcn_readfirstlane(max(max(n[0], n[1]), max(n[2], n[3])));
#endif
        for (int k = 0; k < n_it; k++) {
            const bool act = k < my_n;
            const int j = act ? (int)my_list[k] : kRoundR;
            const int contributor = toDo - j - 1;          // position of splat j in the tile's list, counted from 0
            uint32_t rec_off = (uint32_t)j * 48u;
            asm volatile("" : "+v"(rec_off));
            const char* rec_j = reinterpret_cast<const char*>(s_rec) + rec_off;
            const float4 r0 = *reinterpret_cast<const float4*>(rec_j);
            const float4 r1 = *reinterpret_cast<const float4*>(rec_j + 16);
            float dx, dy;
            const float power = splat_power(r0, r1, pxf, pyf, dx, dy);
            if (!__any(!(power < r0.w))) continue;
            const float G = mom_exp(power);
            const float alpha = fminf(0.99f, r1.w * G);
            const uint64_t vmask = m_inside & __builtin_amdgcn_ballot_w64(contributor < last_contributor) &
                                   __builtin_amdgcn_ballot_w64(!(power > 0.0f)) & __builtin_amdgcn_ballot_w64(!(alpha < 1.0f / 255.0f));
            if (vmask == 0) continue;
            const float alpha_in = alpha, G_in = G;
            float w, a, g_op;
            const float4 r2 = *reinterpret_cast<const float4*>(rec_j + 32);
            {
                float alpha, G;
                asm("v_cndmask_b32_e64 %0, 0, %2, %4\n\tv_cndmask_b32_e64 %1, 0, %3, %4" : "=&v"(alpha), "=v"(G) : "v"(alpha_in), "v"(G_in), "s"(vmask));
                const float one_m_alpha = 1.f - alpha;
                const float inv_1ma = __builtin_amdgcn_rcpf(one_m_alpha);
                T = T * inv_1ma;
                w = alpha * T;
                float dL_dalpha = (r2.x - accum0) * dp0;
                dL_dalpha += (r2.y - accum1) * dp1;
                dL_dalpha += (r2.z - accum2) * dp2;
                accum0 = __builtin_fmaf(alpha, r2.x, one_m_alpha * accum0);
                accum1 = __builtin_fmaf(alpha, r2.y, one_m_alpha * accum1);
                accum2 = __builtin_fmaf(alpha, r2.z, one_m_alpha * accum2);
                if (DEPTH) {
                    dL_dalpha += (r0.z - accum_d) * dpd;
                    accum_d = __builtin_fmaf(alpha, r0.z, one_m_alpha * accum_d);
                }
                dL_dalpha *= T;
                dL_dalpha += (-T_final * inv_1ma) * bg_dot_dpixel;
                g_op = G * dL_dalpha;
                a = r1.w * g_op;
            }
            const float g_c0 = w * dp0, g_c1 = w * dp1, g_c2 = w * dp2, g_d = DEPTH ? w * dpd : 0.f;
            const float ax = a * dx, ay = a * dy;
            const float g_cx = ax * dx, g_cy = ax * dy, g_cw = ay * dy;
            float v;
            if (DEPTH) {
                const float gv[10] = {ax, ay, g_cx, g_cy, g_cw, g_op, g_c0, g_c1, g_c2, g_d};
                v = row_totals<10>(gv, kBit0, kSel2);
            } else {
                const float gv[9] = {ax, ay, g_cx, g_cy, g_cw, g_op, g_c0, g_c1, g_c2};
                v = row_totals<9>(gv, kBit0, kSel2);
            }
            // lane k of a row holds the row's total of value k: one accumulator row per (row, splat) pair
#ifdef MOM_ROWS_NOATOMIC
            if (v == 123.456f) gacc[0] = v;
#else
            if (vslot < kVals && act) atomicAdd(&gacc[(size_t)__float_as_uint(r2.w) * MOM_GACC_FLOATS + vslot], v);
#endif
        }
    }
}


