// Deformation field in one pass: HexPlane lookup fused in front of the deformation MLP (forward), gfx950.
//
// Replaces deform_network.forward -> Deformation.query_time + forward_dynamic for the render() case -- ONE timestamp for all
// points (reference scene/deformation.py:97-153 calling scene/hexplane.py:73-106,160-183: 12 x F.grid_sample + product,
// then the 7 nn.Linear of the trunk and the pos / scales / rotations heads).  The separate kernels (hexplane.hip,
// deform_mlp.hip) pass feat[P,64] through HBM and are bound by vector-instruction issue (HexPlane) and by the phases around
// the MFMAs (MLP); here a wave gathers the features of its 32 Gaussians straight into the MFMA's B-operand tile in LDS, and
// while it runs its 256 MFMAs the SIMD's other wave gathers.
//
// Two things make the gather cheap enough to hide:
//  * One timestamp per frame collapses the three space-time planes to LINES: the bilinear sample of plane (x,t) at (x, t) is
//    ax * L[x0] + bx * L[x0+1] with L[r] = ay * plane[t0][r] + by * plane[t1][r] the same for every point.  A 73 KB table of
//    lines per frame (hexplane_lines_kernel) replaces 12 of the 24 texel rows per (point, level) by 6.
//  * EIGHT lanes own one (point, level): lane c of the eight holds channels 4c..4c+3, one texel row is one 128-byte line
//    fetched by eight 16-byte loads, a wave instruction covers eight rows.  The 32-lanes-per-row form of hexplane.hip issues
//    four times as many loads and address computations for the same FMAs.
//
// Arithmetic follows ATen's grid_sampler_2d (weights nw, ne, sw, se from bx = ix - x0 and 1 - bx) and the reference's product
// order over the planes; only the two time rows are combined first (a reassociation of the same four products).
//
// BUILD CONSTRAINT (correctness, not speed): no packed fp32 arithmetic in this file.  With v_pk_fma_f32 / v_pk_mul_f32 in the
// gather -- the SLP vectoriser's output -- about one launch in ten left wrong values in lanes 48-63 while other waves of the SIMD
// issued bf16 MFMAs (never without the MFMAs, never with scalar fp32).  The Makefile builds every file with
// -fno-slp-vectorize; tests/test_isa.py disassembles the shipped code object and fails on any v_pk_*_f32 in a kernel of this file
// or in any kernel that issues MFMAs, and tests/test_deform_field_gpu.py::test_soak_500_launches_are_bit_identical is the
// behavioural guard.  Do not hand-write float2 arithmetic here either: clang lowers it to the same packed forms.
#include "deform_mlp_dev.h"
#include "hexplane_dev.h"
#include "deform_b3_dev.h"
#include <stdlib.h>

namespace {

constexpr int kTileStride = 68;                      // B-operand tile rows: [32 gaussians][64 features + 4]: 16-byte aligned, conflict-free
constexpr int kTileFloats = 32 * kTileStride;
constexpr int kRecDw = 8;                            // dwords of one (point, level) record
constexpr int kBufFloats = kTileFloats + 64 * kRecDw;   // one gather wave's buffer: the tile + its records
#define MOM_FIELD_PRIO 2
#define MOM_FIELD_NG 2
#define MOM_FIELD_NM 1
constexpr int kNG = MOM_FIELD_NG;                    // gather waves per SIMD
constexpr int kNM = MOM_FIELD_NM;                    // MFMA waves per SIMD: they take the SIMD's tiles in turn
constexpr int kMfmaGroup = (kNM + kNG) >= 4 ? 8 : 16;   // A-fragment prefetch depth (registers)
constexpr int kNB = kNG > kNM ? kNG : kNM;           // tile buffers per SIMD
constexpr int kFieldWaves = 4 * (kNM + kNG);
constexpr int kLFlags = kLFwdTotal;                  // [4 SIMDs][kNG][2] ints: tiles published / tiles consumed
constexpr int kLBufs = kLFlags + 32;
constexpr int kLFieldTotal = kLBufs + 4 * kNB * kBufFloats;


// lines[level][axis][r][32] = ay * plane_(axis,t)[t0][r][:] + by * plane_(axis,t)[t1][r][:]
__global__ void __launch_bounds__(256) hexplane_lines_kernel(HexArgs a, LineTab lt, float* __restrict__ lines, int total)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    int lvl = 0, ax = 0;
    for (int l = 0; l < a.levels; l++)
        for (int k = 0; k < 3; k++)
            if ((unsigned)i >= lt.off[l][k]) { lvl = l; ax = k; }
    const int p = ax == 0 ? 2 : (ax == 1 ? 4 : 5);   // planes (x,t), (y,t), (z,t)
    const int Wd = a.res[lvl][ax], Td = a.res[lvl][3];
    const int rel = i - (int)lt.off[lvl][ax];        // r * 32 + ch
    int t0, t1;
    float w0, w1;
    time_sample(a.time, Td, t0, t1, w0, w1);
    const float* __restrict__ pl = a.planes[lvl][p];
    float v = 0.f;
    if (t0 >= 0) v += pl[(size_t)t0 * Wd * 32 + rel] * w0;
    if (t1 >= 0) v += pl[(size_t)t1 * Wd * 32 + rel] * w1;
    lines[i] = v;
}

struct WeightRegs {
    float4 v[16];
    float bias;
};
// load_weights of deform_mlp_dev.h in two halves, so that the gather of a wave's first tile runs while the 70 KB of weights
// are in flight (256 CUs pull the same lines out of L2 at once: 16 k cycles when waited for on the spot)
__device__ __forceinline__ void weights_issue(const MlpDev& m, WeightRegs& w, int tid, int nth)
{
    const float* Ws[4] = {m.W0, m.W1[0], m.W1[1], m.W1[2]};
    const float* bs[4] = {m.b0, m.b1[0], m.b1[1], m.b1[2]};
    constexpr int kQuads = 4 * kHid * kHid / 4;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const int q = tid + j * nth;
        if (q < kQuads) w.v[j] = reinterpret_cast<const float4*>(Ws[q >> 10])[q & 1023];
    }
    w.bias = 0.f;
    if (tid < 4 * kHid) w.bias = bs[tid >> 6][tid & 63];
}
__device__ __forceinline__ void weights_commit(const MlpDev& m, const WeightRegs& w, float* __restrict__ lds, int tid, int nth)
{
    constexpr int kQuads = 4 * kHid * kHid / 4;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const int q = tid + j * nth;
        if (q < kQuads) {
            const int L = q >> 10, i = 4 * (q & 1023), o = i >> 6, k = i & 63;
            float* d = lds + kLW + L * kWFloats + k * kWStride + o;
            d[0] = w.v[j].x; d[kWStride] = w.v[j].y; d[2 * kWStride] = w.v[j].z; d[3 * kWStride] = w.v[j].w;
        }
    }
    if (tid < 4 * kHid) lds[kLB + tid] = w.bias;
    for (int i = tid; i < 3 * 4 * kHid; i += nth) {
        const int head = i >> 8, n = (i >> 6) & 3, f = i & 63;
        const int nout = head == 2 ? 4 : 3;
        lds[kLW2 + i] = n < nout ? m.W2[head][n * kHid + f] : 0.f;
    }
    if (tid < 12) {
        const int head = tid >> 2, n = tid & 3;
        const int nout = head == 2 ? 4 : 3;
        lds[kLB2 + tid] = n < nout ? m.b2[head][n] : 0.f;
    }
}

__device__ __forceinline__ float4 ld4(const float* __restrict__ base, unsigned byte_off)
{
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ float4 mul4(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 fma4(float4 a, float s, float4 c)
{
    return make_float4(__builtin_fmaf(a.x, s, c.x), __builtin_fmaf(a.y, s, c.y), __builtin_fmaf(a.z, s, c.z), __builtin_fmaf(a.w, s, c.w));
}
__device__ __forceinline__ float4 mul44(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }

// Record of one (point, level), written by ONE lane per tile (phase A) and read by the eight lanes that gather it:
//   R0 = {byte offset of texel (y0,x0) of plane (x,y) | hx | hy << 1,  same of (z0,x0) of (x,z) | hz << 1,  same of (z0,y0) of
//         (y,z),  x0 | y0 << 10 | z0 << 20}      hx / hy / hz: the cell's upper neighbour along that axis is inside
//   R1 = {bx, by, bz, -}                        fractions; 1 - b is ATen's lower weight bit for bit (Sterbenz)
// a.res[lvl][k] for a level that differs between the lanes of a wave, picked from the four scalar values: indexed with a vector
// register the argument block is read from memory, one load per axis, each issued behind the previous axis' clip branches and
// waited for on the spot (three more round trips on a chain that has two to begin with: order -> position).
__device__ __forceinline__ int res_of(const HexArgs& a, int lvl, int k)
{
    int r0 = a.res[0][k], r1 = a.res[1][k], r2 = a.res[2][k], r3 = a.res[3][k];
    asm volatile("" : "+s"(r0), "+s"(r1), "+s"(r2), "+s"(r3));       // (opaque: the compiler turns a select of loads into a load of the selected address)
    int r = r0;
    r = lvl == 1 ? r1 : r;
    r = lvl == 2 ? r2 : r;
    r = lvl == 3 ? r3 : r;
    return r;
}
__device__ __forceinline__ void make_record(const HexArgs& a, const float* __restrict__ xyz, int g, int lvl, uint4& R0, float4& R1)
{
    float c[4];
    norm_coords(a, xyz, g, c);
    int i0[3];
    float b[3];
    bool hn[3];
    const int res[3] = {res_of(a, lvl, 0), res_of(a, lvl, 1), res_of(a, lvl, 2)};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float gm;
        const float ix = unnorm_clip(c[k], res[k], gm);
        i0[k] = (int)floorf(ix);
        b[k] = ix - (float)i0[k];
        hn[k] = i0[k] + 1 < res[k];
    }
    const unsigned Wx = (unsigned)res[0], Wy = (unsigned)res[1];
    R0.x = ((unsigned)i0[1] * Wx + (unsigned)i0[0]) * 128u | (hn[0] ? 1u : 0u) | (hn[1] ? 2u : 0u);
    R0.y = ((unsigned)i0[2] * Wx + (unsigned)i0[0]) * 128u | (hn[2] ? 2u : 0u);
    R0.z = ((unsigned)i0[2] * Wy + (unsigned)i0[1]) * 128u;
    R0.w = (unsigned)i0[0] | ((unsigned)i0[1] << 10) | ((unsigned)i0[2] << 20);
    R1 = make_float4(b[0], b[1], b[2], 0.f);
}

// The 18 texel rows (16 bytes per lane each) of one (point, level) in flight, with what finishing them needs.
struct UnitLoads {
    float4 t[18];          // (x,y) nw ne sw se | (x,z) | (y,z) | line x: r0 r1 | line y | line z
    float bx, by, bz;
    int gl;
};

// Gather the 64 features of the tile's 32 Gaussians into `tile` ([gaussian][kTileStride]) and, optionally, into feat[P][64].
// g_mine: the Gaussian of lane (lane & 31) (or -1 past the end).  Eight lanes work on one (point, level); the eight units of a
// pass are software pipelined two deep: the loads of unit u + 2 are issued as soon as unit u's registers are free, so two
// memory round trips are always in flight (issued one unit at a time and waited for on the spot, a tile took 34-41 k cycles).
__device__ __forceinline__ void gather_tile(const HexArgs& a, const LineTab& lt, const float* __restrict__ lines,
                                            const float* __restrict__ xyz, int g_mine, float* __restrict__ tile,
                                            uint4* __restrict__ rec, float* __restrict__ feat_save, int lane)
{
    // phase A: lane = unit (gaussian lane & 31, level lane >> 5)
    {
        const int lvl = lane >> 5;
        uint4 R0 = make_uint4(0, 0, 0, 0);
        float4 R1 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g_mine >= 0 && lvl < a.levels) make_record(a, xyz, g_mine, lvl, R0, R1);
        rec[2 * lane] = R0;
        rec[2 * lane + 1] = make_uint4(__float_as_uint(R1.x), __float_as_uint(R1.y), __float_as_uint(R1.z), 0u);
    }
    __builtin_amdgcn_wave_barrier();
    // phase B: eight lanes per unit; pass u = 4 lvl + i covers the gaussians 8 i .. 8 i + 7 of the tile at level lvl
    const int g8 = lane >> 3, c = lane & 7;
    const unsigned cb = (unsigned)c * 16u;
    // what a pass needs of its level: read from the argument block ONCE per level (indexed by the pass number inside the pass loop,
    // every pass began with three dependent scalar loads -- level -> plane pointer / line offset -> ... -- ahead of its 18 rows)
    struct LevelTab {
        unsigned rowx, rowy;
        const float *pxy, *pxz, *pyz, *lx, *ly, *lz;
    };
    auto issue = [&](int lvl, int i, const LevelTab& T, UnitLoads& L) {
        const int gl = 8 * i + g8;
        const uint4 R0 = rec[2 * (32 * lvl + gl)];
        const uint4 R1u = rec[2 * (32 * lvl + gl) + 1];
        L.bx = __uint_as_float(R1u.x); L.by = __uint_as_float(R1u.y); L.bz = __uint_as_float(R1u.z);
        L.gl = gl;
        const unsigned rowx = T.rowx, rowy = T.rowy;
        const unsigned sx = (R0.x & 1u) ? 128u : 0u, sy = (R0.x & 2u) ? 128u : 0u, sz = (R0.y & 2u) ? 128u : 0u;
        const unsigned ry_x = (R0.x & 2u) ? rowx : 0u, rz_x = (R0.y & 2u) ? rowx : 0u, rz_y = (R0.y & 2u) ? rowy : 0u;
        const unsigned x0 = R0.w & 1023u, y0 = (R0.w >> 10) & 1023u, z0 = R0.w >> 20;
        const float* __restrict__ pxy = T.pxy;
        const float* __restrict__ pxz = T.pxz;
        const float* __restrict__ pyz = T.pyz;
        const unsigned oxy = (R0.x & ~127u) + cb, oxz = (R0.y & ~127u) + cb, oyz = (R0.z & ~127u) + cb;
        L.t[0] = ld4(pxy, oxy); L.t[1] = ld4(pxy, oxy + sx); L.t[2] = ld4(pxy, oxy + ry_x); L.t[3] = ld4(pxy, oxy + sx + ry_x);
        L.t[4] = ld4(pxz, oxz); L.t[5] = ld4(pxz, oxz + sx); L.t[6] = ld4(pxz, oxz + rz_x); L.t[7] = ld4(pxz, oxz + sx + rz_x);
        L.t[8] = ld4(pyz, oyz); L.t[9] = ld4(pyz, oyz + sy); L.t[10] = ld4(pyz, oyz + rz_y); L.t[11] = ld4(pyz, oyz + sy + rz_y);
        const float* __restrict__ lx = T.lx;
        const float* __restrict__ ly = T.ly;
        const float* __restrict__ lz = T.lz;
        L.t[12] = ld4(lx, x0 * 128u + cb); L.t[13] = ld4(lx, x0 * 128u + cb + sx);
        L.t[14] = ld4(ly, y0 * 128u + cb); L.t[15] = ld4(ly, y0 * 128u + cb + sy);
        L.t[16] = ld4(lz, z0 * 128u + cb); L.t[17] = ld4(lz, z0 * 128u + cb + sz);
    };
    auto space = [&](const float4* t, float a0, float b0, float a1, float b1) {      // ATen's order nw, ne, sw, se
        float4 v = mul4(t[0], a0 * a1);
        v = fma4(t[1], b0 * a1, v);
        v = fma4(t[2], a0 * b1, v);
        v = fma4(t[3], b0 * b1, v);
        return v;
    };
    auto finish = [&](int lvl, const UnitLoads& L) {
        const float bx = L.bx, by = L.by, bz = L.bz, ax = 1.f - bx, ay = 1.f - by, az = 1.f - bz;
        // reference order of the product: (x,y) (x,z) (x,t) (y,z) (y,t) (z,t)
        float4 f = mul44(space(L.t, ax, bx, ay, by), space(L.t + 4, ax, bx, az, bz));
        f = mul44(f, fma4(L.t[13], bx, mul4(L.t[12], ax)));
        f = mul44(f, space(L.t + 8, ay, by, az, bz));
        f = mul44(f, fma4(L.t[15], by, mul4(L.t[14], ay)));
        f = mul44(f, fma4(L.t[17], bz, mul4(L.t[16], az)));
        if (tile) *reinterpret_cast<float4*>(tile + L.gl * kTileStride + 32 * lvl + 4 * c) = f;
        if (feat_save) {
            const int g = __shfl(g_mine, L.gl);
            if (g >= 0) *reinterpret_cast<float4*>(feat_save + (size_t)g * kHid + 32 * lvl + 4 * c) = f;
        }
    };
#pragma unroll 1
    for (int lvl = 0; lvl < 2; lvl++) {
        const LevelTab T = {(unsigned)a.res[lvl][0] * 128u, (unsigned)a.res[lvl][1] * 128u, a.planes[lvl][0], a.planes[lvl][1], a.planes[lvl][3],
                            lines + lt.off[lvl][0], lines + lt.off[lvl][1], lines + lt.off[lvl][2]};
#pragma unroll 1
        for (int i = 0; i < 4; i++) {    // one pass in flight per wave
            UnitLoads LA;
            issue(lvl, i, T, LA);
            finish(lvl, LA);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// B operand of the trunk layer from the tile: register r of lane half h is feature fmap(r, h) = (r & 3) + 8 (r >> 2) + 4 h
__device__ __forceinline__ void load_tile(const float* __restrict__ tile, int col, int h, f32x16 (&t)[2])
{
#pragma unroll
    for (int kt = 0; kt < 2; kt++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 v = *reinterpret_cast<const float4*>(tile + col * kTileStride + 32 * kt + 8 * q + 4 * h);
            t[kt][4 * q + 0] = v.x;
            t[kt][4 * q + 1] = v.y;
            t[kt][4 * q + 2] = v.z;
            t[kt][4 * q + 3] = v.w;
        }
}

#ifdef MOM_FIELD_STAMPS
__device__ unsigned long long g_field_dbg[256 * 32 * 4];
#define STAMP() __builtin_amdgcn_s_memtime()
#else
#define STAMP() 0ull
#endif
__device__ __forceinline__ void flag_wait(const int* f, int need)
{
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(1);
}
__device__ __forceinline__ void flag_set(int* f, int v) { __hip_atomic_store(f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

// Waves specialise.  Waves 0..3 (one per SIMD) run the MLP of their SIMD's tiles on the matrix pipe and never wait for memory;
// waves 4.. (kNG per SIMD) gather: each owns one tile buffer in LDS, fills it for its next tile while the MFMA wave works through
// the previous one, and hands it over through two counters in LDS (tiles published / tiles consumed).  The MFMA wave frees a
// buffer as soon as the tile sits in its registers, so the gather has a whole tile's MFMA time (16 k cycles) for the next one.
// Run as ONE program per wave (gather a tile, then multiply it) the two waves of a SIMD moved in lockstep -- all gathering, then
// all queueing for the matrix pipe -- and the kernel took the sum of the two phases (120 us); so did the two separate kernels.
__global__ void __launch_bounds__(64 * kFieldWaves)
deform_field_fwd_kernel(HexArgs a, LineTab lt, MlpDev m, int tiles, const float* __restrict__ lines, const float* __restrict__ xyz,
                        const float* __restrict__ scaling, const float* __restrict__ rotation, const float* __restrict__ flow,
                        float flow_coef, float* __restrict__ pts, float* __restrict__ scales, float* __restrict__ rots,
                        float* __restrict__ feat_save, float* __restrict__ a0_save, ActOut act)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, col = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6;
    const int P = a.P;
    const int t_begin = (int)((long long)tiles * blockIdx.x / gridDim.x), t_end = (int)((long long)tiles * (blockIdx.x + 1) / gridDim.x);
    int* flags = reinterpret_cast<int*>(lds + kLFlags);
    if (threadIdx.x < 32) flags[threadIdx.x] = 0;
    auto gaussian_of = [&](int t) {
        const int gi = t * 32 + col;
        return gi < P ? (a.order ? (int)a.order[gi] : gi) : -1;
    };
    const bool mfma_wave = wv < 4 * kNM;
    const int simd = wv & 3, gj = mfma_wave ? wv >> 2 : (wv - 4 * kNM) >> 2;      // waves w, w + 4, w + 8 share a SIMD (speed only)
    WeightRegs wr;
    if (mfma_wave) weights_issue(m, wr, (int)threadIdx.x, 256 * kNM);
    __syncthreads();                                                   // the flags are zero
    if (!mfma_wave) {
        // the SIMD's k-th tile (tile t_begin + simd + 4 k) goes through buffer k % kNB; gather wave gj fills k = gj, gj + kNG, ...
        unsigned long long tw = 0, tg = 0, t00 = STAMP();
        int n_done = 0;
        for (int k = gj; t_begin + simd + 4 * k < t_end; k += kNG, n_done++) {
            const int t = t_begin + simd + 4 * k, j = k % kNB, n = k / kNB;
            float* tile = lds + kLBufs + (simd * kNB + j) * kBufFloats;
            uint4* rec = reinterpret_cast<uint4*>(tile + kTileFloats);
            int* f_ready = flags + (simd * kNB + j) * 2, *f_free = f_ready + 1;
            const int g = gaussian_of(t);
            const unsigned long long s0 = STAMP();
            flag_wait(f_free, n);                                      // the buffer's previous tile sits in an MFMA wave's registers
            const unsigned long long s1 = STAMP();
            gather_tile(a, lt, lines, xyz, g, tile, rec, feat_save, lane);
            flag_set(f_ready, n + 1);
            const unsigned long long s2 = STAMP();
            tw += s1 - s0; tg += s2 - s1;
        }
#ifdef MOM_FIELD_STAMPS
        if (lane == 0) {
            unsigned long long* d = g_field_dbg + ((size_t)blockIdx.x * 32 + wv) * 4;
            d[0] = tw; d[1] = tg; d[2] = STAMP() - t00; d[3] = n_done;
        }
#endif
        return;
    }
    __builtin_amdgcn_s_setprio(MOM_FIELD_PRIO);     // the matrix wave's few vector / LDS instructions go ahead of the gather waves' many
    weights_commit(m, wr, lds, (int)threadIdx.x, 256 * kNM);
    // only the four MFMA waves read the weights: a barrier among them (the gather waves never come here)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_fetch_add(flags + 31, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    flag_wait(flags + 31, 4 * kNM);
    int k = 0;
    unsigned long long tw = 0, t00 = STAMP(), tpro = 0, tl = 0, tm = 0, to = 0;
    // MFMA wave gj of the SIMD takes the SIMD's tiles k = gj, gj + kNM, ...: while one wave is between its MFMA phases (bias,
    // ReLU, output layers, stores) the other's MFMAs keep the pipe busy
    for (k = gj; t_begin + simd + 4 * k < t_end; k += kNM) {
        const int t = t_begin + simd + 4 * k;
        const int j = k % kNB, n = k / kNB;
        const float* tile = lds + kLBufs + (simd * kNB + j) * kBufFloats;
        int* f_ready = flags + (simd * kNB + j) * 2, *f_free = f_ready + 1;
        const int g = gaussian_of(t);
        const bool ok = g >= 0;
        // this Gaussian's inputs of the residual adds: requested now, used after the heads (a load issued where it is used costs
        // this wave -- alone on its SIMD -- the whole memory latency, three times per tile)
        float in_xyz[3] = {0.f, 0.f, 0.f}, in_flow[3] = {0.f, 0.f, 0.f}, in_scal[3] = {0.f, 0.f, 0.f}, in_opac = 0.f;
        float4 in_rot = make_float4(0.f, 0.f, 0.f, 0.f);
        constexpr bool kPrefetchIn = kNM == 1;     // with a second MFMA wave on the SIMD the loads are issued where they are used
        if (kPrefetchIn && h == 0 && ok) {
#pragma unroll
            for (int q = 0; q < 3; q++) {
                in_xyz[q] = xyz[3 * g + q];
                in_flow[q] = flow[3 * g + q];
                in_scal[q] = scaling[3 * g + q];
            }
            in_rot = *reinterpret_cast<const float4*>(rotation + 4 * g);
            if (act.opacity) in_opac = act.opacity_raw[g];
        }
        f32x16 a0[2];
        {
            f32x16 x[2];
            const unsigned long long s0 = STAMP();
            flag_wait(f_ready, n + 1);
            tw += STAMP() - s0;
            if (k == gj) tpro = STAMP() - t00;
            load_tile(tile, col, h, x);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // the tile is in registers: its buffer is free again
            flag_set(f_free, n + 1);
            init_bias(lds + kLB, a0, h);
            const unsigned long long s1 = STAMP();
            layer64p<false, kMfmaGroup>(lds + kLW, x, a0, col, h);
            tl += STAMP() - s1;
        }
        relu_tile(a0);
        if (a0_save) store_feat(a0_save, g, ok, h, a0);
        // Heads.  (A software-pipelined form -- the thin output layer of head k riding between the MFMAs of head k + 1's hidden
        // layer -- pays for a wave that is ALONE on its SIMD and needs two hidden tiles in registers; with two MFMA waves per SIMD the
        // other wave's MFMAs fill those gaps and the plain order is used: round 3, in the git history.)
        f32x16 hcur[2];
#pragma nounroll
        for (int head = 0; head < 3; head++) {
            float o[4];
            const unsigned long long s3 = STAMP();
            init_bias(lds + kLB + (1 + head) * kHid, hcur, h);
            layer64p<false, kMfmaGroup>(lds + kLW + (1 + head) * kWFloats, a0, hcur, col, h);
            tm += STAMP() - s3;
            relu_tile(hcur);
            out_layer(lds + kLW2 + head * 4 * kHid, lds + kLB2 + head * 4, hcur, h, o);
            to += STAMP() - s3;
            if (h == 0 && ok) {
                if (!kPrefetchIn) {
                    if (head == 0) {
#pragma unroll
                        for (int q = 0; q < 3; q++) { in_xyz[q] = xyz[3 * g + q]; in_flow[q] = flow[3 * g + q]; }
                    } else if (head == 1) {
#pragma unroll
                        for (int q = 0; q < 3; q++) in_scal[q] = scaling[3 * g + q];
                    } else {
                        in_rot = *reinterpret_cast<const float4*>(rotation + 4 * g);
                        if (act.opacity) in_opac = act.opacity_raw[g];
                    }
                }
                if (head == 0) {
#pragma unroll
                    for (int q = 0; q < 3; q++) pts[3 * g + q] = in_xyz[q] + (o[q] + flow_coef * in_flow[q]);
                } else if (head == 1) {
                    float s3v[3];
#pragma unroll
                    for (int q = 0; q < 3; q++) {
                        s3v[q] = in_scal[q] + o[q];
                        scales[3 * g + q] = s3v[q];
                    }
                    if (act.scales) {
#pragma unroll
                        for (int q = 0; q < 3; q++) act.scales[3 * g + q] = expf(s3v[q]);
                    }
                } else {
                    const float4 q4 = make_float4(in_rot.x + o[0], in_rot.y + o[1], in_rot.z + o[2], in_rot.w + o[3]);
                    *reinterpret_cast<float4*>(rots + 4 * g) = q4;
                    if (act.rots) {
                        const float nq = mom_quat_norm(q4.x, q4.y, q4.z, q4.w);
                        *reinterpret_cast<float4*>(act.rots + 4 * g) = make_float4(q4.x / nq, q4.y / nq, q4.z / nq, q4.w / nq);
                    }
                    if (act.opacity) act.opacity[g] = mom_sigmoid(in_opac);
                }
            }
        }
    }
#ifdef MOM_FIELD_STAMPS
    if (lane == 0) {
        unsigned long long* d = g_field_dbg + ((size_t)blockIdx.x * 32 + wv) * 4;
        d[0] = tw; d[1] = tpro; d[2] = STAMP() - t00; d[3] = k;
        unsigned long long* e = g_field_dbg + ((size_t)blockIdx.x * 32 + 16 + wv) * 4;
        e[0] = tl; e[1] = tm; e[2] = to; e[3] = 0;
    }
#endif
}

// ---- the MLP on the bf16 matrix cores, fp32-exact operands --------------------------------------------------------------
// v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate and occupies the SIMD's vector ALU while it does (tools/probe/
// mfma_valu_coissue.hip: MFMA-only and FMA-only waves on one SIMD take the SUM of their times), so an f32 MLP can neither beat
// 157 TFLOP/s nor hide the gather's vector work behind its matrix work.  The bf16 pipe is 16 times faster and does co-issue.
// Every fp32 operand is split EXACTLY into three bf16 pieces (x = x1 + x2 + x3: the top 8, the next 8 and the last 8 significant
// bits, by masking -- no rounding anywhere) and a product is formed from the six piece products whose weight is at least
// 2^-16 of the full one (x1 y1, x1 y2, x2 y1, x1 y3, x3 y1, x2 y2; the three dropped ones are below 2^-23 |x||y|, the size of
// fp32's own rounding of the product).  Accumulation is fp32 inside the MFMA.  Six bf16 MFMAs of K = 16 replace eight f32
// MFMAs of K = 2: 0.375 of the matrix cycles, and the vector ALU is free meanwhile.
//
// Layout: D = A B with A = weights (32 outputs x 16 inputs per MFMA), B = activations (16 inputs x 32 Gaussians).  K-step s of a
// layer covers input features 16 s .. 16 s + 15 in the order k(h, j) = 16 s + 4 h + (j & 3) + 8 (j >> 2) for lane half h and
// j = 0..7: exactly the eight accumulator registers 8 (s & 1) .. + 7 of output tile s >> 1 of the previous layer, so a layer's
// B operand is its predecessor's accumulator, split in place, with no shuffle.  The weight fragments are prepared once per
// workgroup in that order: wfrag[layer][piece][mt][s][lane] = 16 bytes = a lane's A operand, 96 KB of LDS in all.

#define MOM_B3_WAVES 12
constexpr int kB3Waves = MOM_B3_WAVES;
constexpr int kL3Frag = 0;                              // uint4 [4 layers][3 pieces][2 mt][4 s][64 lanes]
constexpr int kL3FragU4 = 4 * 3 * 2 * 4 * 64;
constexpr int kL3Small = kL3FragU4 * 4;                 // floats from here: biases [4][64], W2 [3][4][64], b2 [16]
constexpr int kL3B = kL3Small, kL3W2 = kL3B + 4 * kHid, kL3B2 = kL3W2 + 3 * 4 * kHid;
constexpr int kL3Recs = kL3B2 + 16;
constexpr int kL3Total = kL3Recs + kB3Waves * 64 * kRecDw;

// acc[mt] += W_layer x: 48 MFMAs, the smallest products first
__device__ __forceinline__ void layer_b3(const uint4* __restrict__ wf /* [3][2][4][64] of this layer */, const Frag3 (&B)[4], f32x16 (&acc)[2], int lane)
{
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const uint4 a1 = wf[((0 * 2 + mt) * 4 + s) * 64 + lane], a2 = wf[((1 * 2 + mt) * 4 + s) * 64 + lane],
                        a3 = wf[((2 * 2 + mt) * 4 + s) * 64 + lane];
            f32x16 c = acc[mt];
            c = mfma16(a3, B[s].p[0], c);
            c = mfma16(a1, B[s].p[2], c);
            c = mfma16(a2, B[s].p[1], c);
            c = mfma16(a2, B[s].p[0], c);
            c = mfma16(a1, B[s].p[1], c);
            c = mfma16(a1, B[s].p[0], c);
            acc[mt] = c;
        }
}

__global__ void __launch_bounds__(64 * kB3Waves)
deform_field_fwd_b3_kernel(HexArgs a, LineTab lt, MlpDev m, int tiles, const float* __restrict__ lines, const float* __restrict__ xyz,
                           const float* __restrict__ scaling, const float* __restrict__ rotation, const float* __restrict__ flow,
                           float flow_coef, float* __restrict__ pts, float* __restrict__ scales, float* __restrict__ rots,
                           float* __restrict__ feat, float* __restrict__ a0_save, ActOut act)
{
    extern __shared__ float lds[];
    uint4* wfrag = reinterpret_cast<uint4*>(lds + kL3Frag);
    const int lane = threadIdx.x & 63, col = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6;
    uint4* rec = reinterpret_cast<uint4*>(lds + kL3Recs + wv * 64 * kRecDw);
    const int P = a.P;
    const unsigned long long t_start = STAMP();
    // ---- prologue: weight fragments (exact three-way split), biases, output layers
    {
        const float* Ws[4] = {m.W0, m.W1[0], m.W1[1], m.W1[2]};
        const float* bs[4] = {m.b0, m.b1[0], m.b1[1], m.b1[2]};
        for (int slot = threadIdx.x; slot < 4 * 2 * 4 * 64; slot += blockDim.x) {
            const int L = slot >> 9, mt = (slot >> 8) & 1, s = (slot >> 6) & 3, ln = slot & 63;
            const int row = 32 * mt + (ln & 31), k0 = 16 * s + 4 * (ln >> 5);
            const float4 lo = *reinterpret_cast<const float4*>(Ws[L] + row * kHid + k0);
            const float4 hi = *reinterpret_cast<const float4*>(Ws[L] + row * kHid + k0 + 8);
            const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            const Frag3 f = split8(v);
#pragma unroll
            for (int p = 0; p < 3; p++) wfrag[(((L * 3 + p) * 2 + mt) * 4 + s) * 64 + ln] = f.p[p];
        }
        for (int i = threadIdx.x; i < 4 * kHid; i += blockDim.x) lds[kL3B + i] = bs[i >> 6][i & 63];
        for (int i = threadIdx.x; i < 3 * 4 * kHid; i += blockDim.x) {
            const int head = i >> 8, n = (i >> 6) & 3, f = i & 63;
            const int nout = head == 2 ? 4 : 3;
            lds[kL3W2 + i] = n < nout ? m.W2[head][n * kHid + f] : 0.f;
        }
        if (threadIdx.x < 12) {
            const int head = threadIdx.x >> 2, n = threadIdx.x & 3;
            const int nout = head == 2 ? 4 : 3;
            lds[kL3B2 + threadIdx.x] = n < nout ? m.b2[head][n] : 0.f;
        }
    }
    __syncthreads();
    const int t_begin = (int)((long long)tiles * blockIdx.x / gridDim.x), t_end = (int)((long long)tiles * (blockIdx.x + 1) / gridDim.x);
    unsigned long long tg = 0, tb = 0, tl = 0, to = 0;
    int n_t = 0;
    const unsigned long long t_pro = STAMP();
    const int n_waves = (int)(blockDim.x >> 6);
    for (int t = t_begin + wv; t < t_end; t += n_waves, n_t++) {
        const int gi = t * 32 + col;
        const int g = gi < P ? (a.order ? (int)a.order[gi] : gi) : -1;
        const bool ok = g >= 0;
        const unsigned long long s0 = STAMP();
        // this Gaussian's inputs of the residual adds, requested now and used after the heads: loaded where they are used, each
        // head paid a memory round trip (stamps: 4.4 k cycles per head for 150 vector instructions)
        float in_xyz[3] = {0.f, 0.f, 0.f}, in_flow[3] = {0.f, 0.f, 0.f}, in_scal[3] = {0.f, 0.f, 0.f}, in_opac = 0.f;
        float4 in_rot = make_float4(0.f, 0.f, 0.f, 0.f);
        if (h == 0 && ok) {
#pragma unroll
            for (int q = 0; q < 3; q++) {
                in_xyz[q] = xyz[3 * g + q];
                in_flow[q] = flow[3 * g + q];
                in_scal[q] = scaling[3 * g + q];
            }
            in_rot = *reinterpret_cast<const float4*>(rotation + 4 * g);
            if (act.opacity) in_opac = act.opacity_raw[g];
        }
        // features: gathered eight lanes per (point, level), bounced through this wave's own rows of feat[P,64] (kept for the
        // backward anyway) into the MFMA layout; the rows were written by this wave: visible to it after the wait
        gather_tile(a, lt, lines, xyz, g, nullptr, rec, feat, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long s1 = STAMP();
        Frag3 B[4];
        {
            f32x16 x[2];
            load_feat(feat, g, ok, h, x);
            split_tile<false>(x, B);
        }
#ifdef MOM_FIELD_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" :: "v"(B[0].p[0].x), "v"(B[3].p[2].w) : "memory");
#endif
        const unsigned long long s2 = STAMP();
        f32x16 acc[2];
        init_bias(lds + kL3B, acc, h);
        layer_b3(wfrag, B, acc, lane);
        relu_tile(acc);
        if (a0_save) store_feat(a0_save, g, ok, h, acc);
        Frag3 A0[4];
        split_tile<false>(acc, A0);                       // already through the ReLU
        tg += s1 - s0; tb += s2 - s1;
#pragma nounroll
        for (int head = 0; head < 3; head++) {
            const unsigned long long s3 = STAMP();
            f32x16 h1[2];
            init_bias(lds + kL3B + (1 + head) * kHid, h1, h);
            layer_b3(wfrag + (1 + head) * (3 * 2 * 4 * 64), A0, h1, lane);
            relu_tile(h1);
#ifdef MOM_FIELD_STAMPS
            asm volatile("" :: "v"(h1[0][0]), "v"(h1[1][15]));
#endif
            const unsigned long long s4 = STAMP();
            tl += s4 - s3;
            float o[4];
            out_layer(lds + kL3W2 + head * 4 * kHid, lds + kL3B2 + head * 4, h1, h, o);
            if (h == 0 && ok) {
                if (head == 0) {
#pragma unroll
                    for (int q = 0; q < 3; q++) pts[3 * g + q] = in_xyz[q] + (o[q] + flow_coef * in_flow[q]);
                } else if (head == 1) {
                    float s3[3];
#pragma unroll
                    for (int q = 0; q < 3; q++) {
                        s3[q] = in_scal[q] + o[q];
                        scales[3 * g + q] = s3[q];
                    }
                    if (act.scales) {
#pragma unroll
                        for (int q = 0; q < 3; q++) act.scales[3 * g + q] = expf(s3[q]);
                    }
                } else {
                    const float4 q4 = make_float4(in_rot.x + o[0], in_rot.y + o[1], in_rot.z + o[2], in_rot.w + o[3]);
                    *reinterpret_cast<float4*>(rots + 4 * g) = q4;
                    if (act.rots) {
                        const float nq = mom_quat_norm(q4.x, q4.y, q4.z, q4.w);
                        *reinterpret_cast<float4*>(act.rots + 4 * g) = make_float4(q4.x / nq, q4.y / nq, q4.z / nq, q4.w / nq);
                    }
                    if (act.opacity) act.opacity[g] = mom_sigmoid(in_opac);
                }
            }
            to += STAMP() - s4;
        }
    }
#ifdef MOM_FIELD_STAMPS
    if (lane == 0) {
        unsigned long long* d = g_field_dbg + ((size_t)blockIdx.x * 32 + wv) * 4;
        d[0] = tg; d[1] = tb; d[2] = STAMP() - t_start; d[3] = n_t;
        unsigned long long* e = g_field_dbg + ((size_t)blockIdx.x * 32 + 16 + wv) * 4;
        e[0] = tl; e[1] = to; e[2] = t_pro - t_start; e[3] = 0;
    }
#endif
}


// ---- HexPlane backward, pass 1 (gather) in the forward's layout ------------------------------------------------------------
// Same contract as hexplane_bwd5_gather_kernel (hexplane.hip): per (point, level) the six samples, their product, the six
// gv = dfeat * (product of the other five) rows STORED at the point's position in the order of the space plane each row is
// scattered with, and the position gradient reduced over the channels.  What changes is the shape: eight lanes own one
// (point, level) with four channels each, so the 18 texel rows (the space-time planes are this frame's lines: two rows instead
// of four) are 18 sixteen-byte loads per lane and one wave instruction serves eight (point, level) units; the lane-per-channel
// kernel issued 24 four-byte loads per two units and spent half its time issuing vector instructions (37 M per launch).  A wave
// takes both levels of its 32 points, so a point's gradient is complete in registers and is added to dxyz without atomics.
constexpr int kRec6Dw = 16;                           // dwords of one (point, level) record of the backward

__device__ __forceinline__ float4 sub4(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 fma44(float s, float4 a, float4 c)      // s * a + c
{
    return make_float4(__builtin_fmaf(s, a.x, c.x), __builtin_fmaf(s, a.y, c.y), __builtin_fmaf(s, a.z, c.z), __builtin_fmaf(s, a.w, c.w));
}
__device__ __forceinline__ float dot4(float4 a, float4 b, float acc)
{
    acc = __builtin_fmaf(a.x, b.x, acc);
    acc = __builtin_fmaf(a.y, b.y, acc);
    acc = __builtin_fmaf(a.z, b.z, acc);
    return __builtin_fmaf(a.w, b.w, acc);
}
// sum over the eight lanes of a unit (lanes 8k .. 8k+7): row_shr-free butterfly inside a DPP row
__device__ __forceinline__ float unit_sum(float v)
{
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    return v;
}

#define HX6_WAVES 3
template <bool CROWS>
__global__ void __launch_bounds__(256, HX6_WAVES)
hexplane_bwd6_gather_kernel(HexArgs a, LineTab lt, int nchunks, const float* __restrict__ lines, const float* __restrict__ xyz,
                            const float* __restrict__ dfeat, float* __restrict__ dxyz, const uint32_t* __restrict__ inv /* [3][levels][P] */,
                            float* __restrict__ gvbuf /* [3 slots][levels][P][2][32]; CROWS: common-factor rows [3 slots][levels][P][32] */)
{
    __shared__ uint4 s_rec[4][64 * (kRec6Dw / 4)];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (gridDim.x * 256) >> 6;
    uint4* __restrict__ rec = s_rec[wv];
    const int g8 = lane >> 3, c = lane & 7;
    const unsigned cb = (unsigned)c * 16u;
    const size_t plane_floats = (size_t)a.P * 32;

    auto point_of = [&](int chunk) {
        const int gi = chunk * 32 + (lane & 31);
        return chunk < nchunks && gi < a.P ? (a.order ? (int)a.order[gi] : gi) : -1;
    };
    int g_next = point_of(wave);
    asm volatile("" :: "v"(g_next));      // (waited for HERE: with it pending at the loop's head the compiler waits for the prefetch below too)
    for (int chunk = wave; chunk < nchunks; chunk += nwaves) {
        // phase A: lane = unit (point lane & 31 of the chunk, level lane >> 5).  Its chain of dependent round trips is order ->
        // {position, order slots} -> 19 rows; the next chunk's first link is asked for here, a chunk ahead.
        const int g_mine = g_next;
        g_next = point_of(chunk + nwaves);
        __builtin_amdgcn_wave_barrier();
        {
            int lvl = lane >> 5;
            asm volatile("" : "+v"(lvl));        // (opaque: the per-lane addresses built from it are NOT kept in registers across phase B)
            uint4 R0 = make_uint4(0, 0, 0, 0), R2 = make_uint4(0, 0, 0, 0), R3 = make_uint4(0, 0, 0, 0xffffffffu);
            float4 R1 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g_mine >= 0) {
                // the three order slots are asked for together with the position (make_record), not behind its clip branches
                uint32_t iv[3];
#pragma unroll
                for (int k = 0; k < 3; k++) iv[k] = inv[((size_t)k * a.levels + lvl) * a.P + g_mine];
                make_record(a, xyz, g_mine, lvl, R0, R1);
                float cc[4];
                norm_coords(a, xyz, g_mine, cc);
                asm volatile("" :: "v"(iv[0]), "v"(iv[1]), "v"(iv[2]), "v"(cc[0]), "v"(cc[1]), "v"(cc[2]));
                float gm[3];
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    float m;
                    const int res = res_of(a, lvl, k);
                    (void)unnorm_clip(cc[k], res, m);
                    // d(ix)/d(world coordinate): (size-1)/2 * 2/(aabb1 - aabb0), 0 where the coordinate was clipped at the border
                    gm[k] = m != 0.f ? ((float)(res - 1) / 2.f) * (2.0f / (a.a1[k] - a.a0[k])) : 0.f;
                }
                R2 = make_uint4(__float_as_uint(gm[0]), __float_as_uint(gm[1]), __float_as_uint(gm[2]), 0u);
#pragma unroll
                for (int k = 0; k < 3; k++) (&R3.x)[k] = iv[k] * (CROWS ? 128u : 256u);   // CROWS: [slot][level][position][32], else [..][position][space | time][32]
                R3.w = (unsigned)g_mine;
            }
            rec[4 * lane] = R0;
            rec[4 * lane + 1] = make_uint4(__float_as_uint(R1.x), __float_as_uint(R1.y), __float_as_uint(R1.z), 0u);
            rec[4 * lane + 2] = R2;
            rec[4 * lane + 3] = R3;
        }
        __builtin_amdgcn_wave_barrier();
        // phase B: pass u = 4 lvl + i covers the points 8 i .. 8 i + 7 of the chunk at level lvl; the gradient of level 0 waits in
        // registers for level 1's (same lanes, four passes later)
        float gsum[4][3];
#pragma unroll 1
        for (int lvl = 0; lvl < 2; lvl++) {
            const unsigned rowx = (unsigned)a.res[lvl][0] * 128u, rowy = (unsigned)a.res[lvl][1] * 128u;
            const float* __restrict__ pxy = a.planes[lvl][0];
            const float* __restrict__ pxz = a.planes[lvl][1];
            const float* __restrict__ pyz = a.planes[lvl][3];
            const float* __restrict__ lx = lines + lt.off[lvl][0];
            const float* __restrict__ ly = lines + lt.off[lvl][1];
            const float* __restrict__ lz = lines + lt.off[lvl][2];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int unit = 32 * lvl + 8 * i + g8;
                const uint4 R0 = rec[4 * unit], R1u = rec[4 * unit + 1], R2 = rec[4 * unit + 2], R3 = rec[4 * unit + 3];
                const bool live = R3.w != 0xffffffffu;
                const float bx = __uint_as_float(R1u.x), by = __uint_as_float(R1u.y), bz = __uint_as_float(R1u.z);
                const unsigned sx = (R0.x & 1u) ? 128u : 0u, sy = (R0.x & 2u) ? 128u : 0u, sz = (R0.y & 2u) ? 128u : 0u;
                const unsigned ry_x = (R0.x & 2u) ? rowx : 0u, rz_x = (R0.y & 2u) ? rowx : 0u, rz_y = (R0.y & 2u) ? rowy : 0u;
                const unsigned x0 = R0.w & 1023u, y0 = (R0.w >> 10) & 1023u, z0 = R0.w >> 20;
                const unsigned oxy = (R0.x & ~127u) + cb, oxz = (R0.y & ~127u) + cb, oyz = (R0.z & ~127u) + cb;
                float4 t[18];
                t[0] = ld4(pxy, oxy); t[1] = ld4(pxy, oxy + sx); t[2] = ld4(pxy, oxy + ry_x); t[3] = ld4(pxy, oxy + sx + ry_x);
                t[4] = ld4(pxz, oxz); t[5] = ld4(pxz, oxz + sx); t[6] = ld4(pxz, oxz + rz_x); t[7] = ld4(pxz, oxz + sx + rz_x);
                t[8] = ld4(pyz, oyz); t[9] = ld4(pyz, oyz + sy); t[10] = ld4(pyz, oyz + rz_y); t[11] = ld4(pyz, oyz + sy + rz_y);
                t[12] = ld4(lx, x0 * 128u + cb); t[13] = ld4(lx, x0 * 128u + cb + sx);
                t[14] = ld4(ly, y0 * 128u + cb); t[15] = ld4(ly, y0 * 128u + cb + sy);
                t[16] = ld4(lz, z0 * 128u + cb); t[17] = ld4(lz, z0 * 128u + cb + sz);
                float4 go = make_float4(0.f, 0.f, 0.f, 0.f);
                if (live) go = *reinterpret_cast<const float4*>(dfeat + (size_t)R3.w * (a.levels * 32) + 32 * lvl + 4 * c);
                // (the row of dxyz this pass adds to, with the same batch: read at the point of the addition it was one more round
                // trip per pass, waited for on the spot)
                const bool adds = lvl == 1 && dxyz && live && c == 0;
                float dold[3] = {0.f, 0.f, 0.f};
                if (adds) {
#pragma unroll
                    for (int k = 0; k < 3; k++) dold[k] = dxyz[3 * (size_t)R3.w + k];
                }
                // bilinear sample and the two raw derivatives of a space plane (corners nw ne sw se; b0 along the row, b1 across rows)
                float4 v[6], da[6], db[3];
                auto space = [&](const float4* q, float b0, float b1, float4& val, float4& d_first, float4& d_second) {
                    const float4 d0 = sub4(q[1], q[0]), d1 = sub4(q[3], q[2]);
                    const float4 tx0 = fma44(b0, d0, q[0]), tx1 = fma44(b0, d1, q[2]);
                    const float4 dy = sub4(tx1, tx0);
                    val = fma44(b1, dy, tx0);
                    d_first = fma44(b1, sub4(d1, d0), d0);
                    d_second = dy;
                };
                auto line = [&](const float4* q, float b, float4& val, float4& d) {
                    d = sub4(q[1], q[0]);
                    val = fma44(b, d, q[0]);
                };
                // planes in the reference's order: 0 (x,y)  1 (x,z)  2 (x,t)  3 (y,z)  4 (y,t)  5 (z,t)
                space(t, bx, by, v[0], da[0], db[0]);
                space(t + 4, bx, bz, v[1], da[1], db[1]);
                line(t + 12, bx, v[2], da[2]);
                space(t + 8, by, bz, v[3], da[3], db[2]);
                line(t + 14, by, v[4], da[4]);
                line(t + 16, bz, v[5], da[5]);
                float4 pre[6], suf[6];
                pre[0] = go;                                   // dfeat rides in the prefix products
#pragma unroll
                for (int p = 1; p < 6; p++) pre[p] = mul44(pre[p - 1], v[p - 1]);
                suf[5] = make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
                for (int p = 4; p >= 0; p--) suf[p] = mul44(suf[p + 1], v[p + 1]);
                float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
                for (int p = 0; p < 6; p++) {
                    const float4 gv = p == 5 ? pre[5] : mul44(pre[p], suf[p]);
                    if (!CROWS && live) {
                        const unsigned pos = p == 0 || p == 2 ? R3.x : (p == 1 || p == 5 ? R3.y : R3.z);     // order slot of the plane
                        *reinterpret_cast<float4*>(reinterpret_cast<char*>(gvbuf + ((size_t)(p == 0 || p == 2 ? 0 : (p == 1 || p == 5 ? 1 : 2)) * a.levels + lvl) * 2 * plane_floats) +
                                                   pos + (p == 2 || p == 4 || p == 5 ? 128u : 0u) + cb) = gv;
                    }
                    // first coordinate of the plane: x for 0 1 2, y for 3 4, z for 5; second: y for 0, z for 1 and 3
                    if (p < 3) gx = dot4(gv, da[p], gx);
                    else if (p < 5) gy = dot4(gv, da[p], gy);
                    else gz = dot4(gv, da[p], gz);
                    if (p == 0) gy = dot4(gv, db[0], gy);
                    if (p == 1) gz = dot4(gv, db[1], gz);
                    if (p == 3) gz = dot4(gv, db[2], gz);
                }
                // CROWS: what goes to memory is ONE row per order slot, not the two gv rows of the slot's planes: the factor they share,
                // c = dfeat * (product of the four OTHER planes' samples).  The scatter pass multiplies it by the time line's sample
                // (gv of the space plane) and by the space plane's own sample (gv of the time plane) -- both of which it has at hand:
                // the line values sit in its LDS and the space plane's four texel rows are the rows it is accumulating into.
                // 154 MB of rows written and read back per step instead of 307 MB each way, on a stretch bound by memory bandwidth.
                //   slot 0 = (x,y) + (x,t): planes 0, 2     slot 1 = (x,z) + (z,t): planes 1, 5     slot 2 = (y,z) + (y,t): planes 3, 4
                if (CROWS && live) {
                    const float4 c0 = mul44(mul44(go, v[1]), suf[2]);                    // go v1 (v3 v4 v5)
                    const float4 c1 = mul44(pre[1], mul44(mul44(v[2], v[3]), v[4]));     // (go v0) v2 v3 v4
                    const float4 c2 = mul44(pre[3], v[5]);                               // (go v0 v1 v2) v5
                    char* crow = reinterpret_cast<char*>(gvbuf + (size_t)lvl * plane_floats) + cb;
                    *reinterpret_cast<float4*>(crow + R3.x) = c0;
                    *reinterpret_cast<float4*>(crow + (size_t)a.levels * plane_floats * 4 + R3.y) = c1;
                    *reinterpret_cast<float4*>(crow + (size_t)2 * a.levels * plane_floats * 4 + R3.z) = c2;
                }
                gx = unit_sum(gx) * __uint_as_float(R2.x);
                gy = unit_sum(gy) * __uint_as_float(R2.y);
                gz = unit_sum(gz) * __uint_as_float(R2.z);
                if (lvl == 0) { gsum[i][0] = gx; gsum[i][1] = gy; gsum[i][2] = gz; }
                else if (adds) {
                    float* d = dxyz + 3 * (size_t)R3.w;
                    d[0] = dold[0] + (gsum[i][0] + gx);
                    d[1] = dold[1] + (gsum[i][1] + gy);
                    d[2] = dold[2] + (gsum[i][2] + gz);
                }
            }
        }
    }
}

}  // namespace

// bytes of the time-line table (what mom_hexplane_backward_scratch_bytes adds for the gather below)
size_t mom_hexplane_lines_bytes(const MomHexPlane* hp)
{
    LineTab lt;
    return mom_align_up((size_t)line_table(hp, &lt) * sizeof(float)) + MOM_ALIGN;
}

// pass 1 of the two-pass HexPlane backward for a field mom_deform_field_supported() accepts (called by mom_hexplane_backward,
// hexplane.hip): the frame's lines into `lines`, then the gather
int mom_launch_hexplane_gather6(const MomHexPlane* hp, int P, const float* xyz, float time, const uint32_t* order, const float* dfeat,
                                float* dxyz, const uint32_t* plane_inverse, float* gvbuf, float* lines, bool lines_ready, bool crows, hipStream_t s)
{
    HexArgs a;
    fill_args(hp, P, nullptr, time, order, true, &a);
    LineTab lt;
    const int nline = line_table(hp, &lt);
    if (!lines_ready) hipLaunchKernelGGL(hexplane_lines_kernel, dim3((nline + 255) / 256), dim3(256), 0, s, a, lt, lines, nline);
    const int nchunks = (P + 31) / 32;
    static int cap = 0;
    // workgroups (MOM_HEX6_BLOCKS overrides); measured, gather + scatter beside dW: 512 293 us, 1024 281, 1536 280, 4096 280
    if (!cap) { const char* e = getenv("MOM_HEX6_BLOCKS"); cap = e ? atoi(e) : 1536; if (cap < 1) cap = 1536; }
    int blocks = (nchunks + 3) / 4;
    if (blocks > cap) blocks = cap;
    if (crows)
        hipLaunchKernelGGL(hexplane_bwd6_gather_kernel<true>, dim3(blocks), dim3(256), 0, s, a, lt, nchunks, lines, xyz, dfeat, dxyz, plane_inverse, gvbuf);
    else
        hipLaunchKernelGGL(hexplane_bwd6_gather_kernel<false>, dim3(blocks), dim3(256), 0, s, a, lt, nchunks, lines, xyz, dfeat, dxyz, plane_inverse, gvbuf);
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}

extern "C" size_t mom_deform_field_scratch_bytes(const MomHexPlane* hp, int P)
{
    if (!hp) return MOM_ALIGN;
    LineTab lt;
    // the table of time lines, then a [P,64] feature buffer for calls that keep no copy of the features (feat_save == null)
    return mom_align_up((size_t)line_table(hp, &lt) * sizeof(float)) + (size_t)(P > 0 ? P : 0) * kHid * sizeof(float) + 2 * MOM_ALIGN;
}

extern "C" int mom_deform_field_supported(const MomHexPlane* hp)
{
    if (!hp || hp->channels != 32 || hp->levels != 2) return 0;
    for (int l = 0; l < 2; l++)
        for (int k = 0; k < 4; k++)
            if (hp->res[l][k] < 1 || hp->res[l][k] > 1024) return 0;
    // texel offsets are 32-bit byte offsets with the low 7 bits free
    for (int l = 0; l < 2; l++)
        if ((unsigned long long)hp->res[l][0] * hp->res[l][1] * 128ull >= (1ull << 31) ||
            (unsigned long long)hp->res[l][0] * hp->res[l][2] * 128ull >= (1ull << 31) ||
            (unsigned long long)hp->res[l][1] * hp->res[l][2] * 128ull >= (1ull << 31))
            return 0;
    return 1;
}

extern "C" int mom_deform_field_forward(const MomHexPlane* hp, const MomDeformMLP* w, int P, const float* xyz, float time,
                                        const uint32_t* order, const float* scaling, const float* rotation, const float* scene_flow,
                                        float flow_coef, float* pts, float* scales, float* rots, float* feat_save, float* a0_save,
                                        const float* opacity_raw, float* scales_act, float* rots_act, float* opacity_act,
                                        void* scratch, mom_stream_t stream)
{
    if (P < 0 || !mom_deform_field_supported(hp)) return MOM_EINVAL;
    if (P == 0) return MOM_OK;
    if (!xyz || !scaling || !rotation || !scene_flow || !pts || !scales || !rots || !scratch) return MOM_EINVAL;
    if ((opacity_act != nullptr) != (opacity_raw != nullptr)) return MOM_EINVAL;
    for (int l = 0; l < 2; l++)
        for (int p = 0; p < 6; p++)
            if (!hp->planes[l][p]) return MOM_EINVAL;
    MlpDev d;
    int rc = fill_dev(w, &d);
    if (rc) return rc;
    HexArgs a;
    fill_args(hp, P, nullptr, time, order, false, &a);
    LineTab lt;
    const int nline = line_table(hp, &lt);
    float* lines = (float*)mom_align_ptr(scratch);
    hipStream_t s = (hipStream_t)stream;
    MomProfScope ps(MOM_P_HEX_FWD, s);
    hipLaunchKernelGGL(hexplane_lines_kernel, dim3((nline + 255) / 256), dim3(256), 0, s, a, lt, lines, nline);
    const int tiles = (P + 31) / 32;
    const ActOut act = {scales_act, rots_act, opacity_act, opacity_raw};
    static int mode = -1;                 // MOM_FIELD_MODE=f32: the f32-MFMA kernel (gather waves feeding MFMA waves; measurement)
    if (mode < 0) {
        const char* e = getenv("MOM_FIELD_MODE");
        mode = (e && e[0] == 'f') ? 1 : 0;
    }
    if (mode == 0) {
        float* feat = feat_save ? feat_save : (float*)mom_align_ptr((char*)lines + mom_align_up((size_t)nline * sizeof(float)));
        // Waves per workgroup (MOM_B3_RUN_WAVES overrides): measured at 200 k Gaussians, 24.4 tiles per CU: 5 waves 116 us, 6: 109,
        // 8: 98, 9: 92, 10: 91, 11: 86, 12: 87 -- the gather wants bytes in flight more than the tile rounds want an even split
        // (12 waves take 3 rounds for 2.03 tiles each, 10 waves 3 rounds for 2.44)
        int waves = kB3Waves;
        {
            static int forced = -1;
            if (forced < 0) { const char* e = getenv("MOM_B3_RUN_WAVES"); forced = e ? atoi(e) : 0; }
            if (forced >= 4 && forced <= kB3Waves) waves = forced;
        }
        const int blocks = tiles < 256 * waves ? (tiles + waves - 1) / waves : 256;
        static bool attr_b3 = false;
        const size_t lds_b3 = sizeof(float) * kL3Total;
        if (!attr_b3) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(deform_field_fwd_b3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds_b3) != hipSuccess)
                return MOM_ELAUNCH;
            attr_b3 = true;
        }
        hipLaunchKernelGGL(deform_field_fwd_b3_kernel, dim3(blocks), dim3(64 * waves), lds_b3, s, a, lt, d, tiles, lines, xyz, scaling,
                           rotation, scene_flow, flow_coef, pts, scales, rots, feat, a0_save, act);
        return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
    }
    // one workgroup per CU (MFMA waves fed by gather waves); a small problem is spread over the CUs
    const int blocks = tiles < 256 * 4 ? (tiles + 3) / 4 : 256;
    static bool attr_set = false;
    const size_t lds_bytes = sizeof(float) * kLFieldTotal;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(deform_field_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes) != hipSuccess)
            return MOM_ELAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL(deform_field_fwd_kernel, dim3(blocks), dim3(64 * kFieldWaves), lds_bytes, s, a, lt, d, tiles, lines, xyz, scaling,
                       rotation, scene_flow, flow_coef, pts, scales, rots, feat_save, a0_save, act);
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}

#ifdef MOM_FIELD_STAMPS
// diagnostic build only: per-wave cycle stamps of the last forward launch ([256 workgroups][16 waves][4])
extern "C" int mom_debug_field_stamps(unsigned long long* host_out)
{
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_field_dbg), sizeof(unsigned long long) * 256 * 32 * 4) == hipSuccess ? 0 : -2;
}
#endif
