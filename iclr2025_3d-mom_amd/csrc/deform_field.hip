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
#include "deform_mlp_dev.h"
#include "hexplane_dev.h"
#include <stdlib.h>

namespace {

constexpr int kTileStride = 68;                      // B-operand tile rows: [32 gaussians][64 features + 4]: 16-byte aligned, conflict-free
constexpr int kTileFloats = 32 * kTileStride;
constexpr int kRecDw = 8;                            // dwords of one (point, level) record
constexpr int kWaveFloats = kTileFloats + 64 * kRecDw;
constexpr int kFieldWaves = 8;                       // waves per workgroup: two per SIMD (one gathers while the other multiplies)
constexpr int kLFieldTotal = kLFwdTotal + kFieldWaves * kWaveFloats;

struct LineTab {
    unsigned off[4][3];                              // float offset of line (level, axis) inside the table
};

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
__device__ __forceinline__ void weights_issue(const MlpDev& m, WeightRegs& w)
{
    const int nth = (int)blockDim.x;
    const float* Ws[4] = {m.W0, m.W1[0], m.W1[1], m.W1[2]};
    const float* bs[4] = {m.b0, m.b1[0], m.b1[1], m.b1[2]};
    constexpr int kQuads = 4 * kHid * kHid / 4;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const int q = threadIdx.x + j * nth;
        if (q < kQuads) w.v[j] = reinterpret_cast<const float4*>(Ws[q >> 10])[q & 1023];
    }
    w.bias = 0.f;
    if (threadIdx.x < 4 * kHid) w.bias = bs[threadIdx.x >> 6][threadIdx.x & 63];
}
__device__ __forceinline__ void weights_commit(const MlpDev& m, const WeightRegs& w, float* __restrict__ lds)
{
    const int nth = (int)blockDim.x;
    constexpr int kQuads = 4 * kHid * kHid / 4;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const int q = threadIdx.x + j * nth;
        if (q < kQuads) {
            const int L = q >> 10, i = 4 * (q & 1023), o = i >> 6, k = i & 63;
            float* d = lds + kLW + L * kWFloats + k * kWStride + o;
            d[0] = w.v[j].x; d[kWStride] = w.v[j].y; d[2 * kWStride] = w.v[j].z; d[3 * kWStride] = w.v[j].w;
        }
    }
    if (threadIdx.x < 4 * kHid) lds[kLB + threadIdx.x] = w.bias;
    for (int i = threadIdx.x; i < 3 * 4 * kHid; i += nth) {
        const int head = i >> 8, n = (i >> 6) & 3, f = i & 63;
        const int nout = head == 2 ? 4 : 3;
        lds[kLW2 + i] = n < nout ? m.W2[head][n * kHid + f] : 0.f;
    }
    if (threadIdx.x < 12) {
        const int head = threadIdx.x >> 2, n = threadIdx.x & 3;
        const int nout = head == 2 ? 4 : 3;
        lds[kLB2 + threadIdx.x] = n < nout ? m.b2[head][n] : 0.f;
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
__device__ __forceinline__ void make_record(const HexArgs& a, const float* __restrict__ xyz, int g, int lvl, uint4& R0, float4& R1)
{
    float c[4];
    norm_coords(a, xyz, g, c);
    int i0[3];
    float b[3];
    bool hn[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float gm;
        const float ix = unnorm_clip(c[k], a.res[lvl][k], gm);
        i0[k] = (int)floorf(ix);
        b[k] = ix - (float)i0[k];
        hn[k] = i0[k] + 1 < a.res[lvl][k];
    }
    const unsigned Wx = (unsigned)a.res[lvl][0], Wy = (unsigned)a.res[lvl][1];
    R0.x = ((unsigned)i0[1] * Wx + (unsigned)i0[0]) * 128u | (hn[0] ? 1u : 0u) | (hn[1] ? 2u : 0u);
    R0.y = ((unsigned)i0[2] * Wx + (unsigned)i0[0]) * 128u | (hn[2] ? 2u : 0u);
    R0.z = ((unsigned)i0[2] * Wy + (unsigned)i0[1]) * 128u;
    R0.w = (unsigned)i0[0] | ((unsigned)i0[1] << 10) | ((unsigned)i0[2] << 20);
    R1 = make_float4(b[0], b[1], b[2], 0.f);
}

// one space plane: four corner rows of 16 bytes per lane, ATen's accumulation order nw, ne, sw, se
__device__ __forceinline__ float4 sample_space(const float* __restrict__ pl, unsigned o, unsigned sx, unsigned sy, float ax, float bx,
                                               float ay, float by)
{
    const float4 t00 = ld4(pl, o), t01 = ld4(pl, o + sx), t10 = ld4(pl, o + sy), t11 = ld4(pl, o + sx + sy);
    float4 v = mul4(t00, ax * ay);
    v = fma4(t01, bx * ay, v);
    v = fma4(t10, ax * by, v);
    v = fma4(t11, bx * by, v);
    return v;
}
__device__ __forceinline__ float4 sample_line(const float* __restrict__ ln, unsigned o, unsigned s, float a, float b)
{
    const float4 l0 = ld4(ln, o), l1 = ld4(ln, o + s);
    return fma4(l1, b, mul4(l0, a));
}

// Gather the 64 features of the tile's 32 Gaussians into `tile` ([gaussian][kTileStride]) and, optionally, into feat[P][64].
// g_mine: the Gaussian of lane (lane & 31) (or -1 past the end).
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
    // phase B: eight lanes per unit; four iterations per level
    const int g8 = lane >> 3, c = lane & 7;
    const unsigned cb = (unsigned)c * 16u;
#pragma unroll 1
    for (int lvl = 0; lvl < 2; lvl++) {
        const unsigned rowx = (unsigned)a.res[lvl][0] * 128u, rowy = (unsigned)a.res[lvl][1] * 128u;
        const float* __restrict__ pxy = a.planes[lvl][0];
        const float* __restrict__ pxz = a.planes[lvl][1];
        const float* __restrict__ pyz = a.planes[lvl][3];
        const float* __restrict__ lx = lines + lt.off[lvl][0];
        const float* __restrict__ ly = lines + lt.off[lvl][1];
        const float* __restrict__ lz = lines + lt.off[lvl][2];
#pragma unroll 2
        for (int i = 0; i < 4; i++) {
            const int gl = 8 * i + g8;                           // gaussian of the tile this lane works on
            const uint4 R0 = rec[2 * (32 * lvl + gl)];
            const uint4 R1u = rec[2 * (32 * lvl + gl) + 1];
            const float bx = __uint_as_float(R1u.x), by = __uint_as_float(R1u.y), bz = __uint_as_float(R1u.z);
            const float ax = 1.f - bx, ay = 1.f - by, az = 1.f - bz;
            const unsigned sx = (R0.x & 1u) ? 128u : 0u, sy = (R0.x & 2u) ? 128u : 0u, sz = (R0.y & 2u) ? 128u : 0u;
            const unsigned x0 = R0.w & 1023u, y0 = (R0.w >> 10) & 1023u, z0 = R0.w >> 20;
            // reference order of the product: (x,y) (x,z) (x,t) (y,z) (y,t) (z,t)
            const float4 vxy = sample_space(pxy, (R0.x & ~127u) + cb, sx, (R0.x & 2u) ? rowx : 0u, ax, bx, ay, by);
            const float4 vxz = sample_space(pxz, (R0.y & ~127u) + cb, sx, (R0.y & 2u) ? rowx : 0u, ax, bx, az, bz);
            const float4 vxt = sample_line(lx, x0 * 128u + cb, sx, ax, bx);
            const float4 vyz = sample_space(pyz, (R0.z & ~127u) + cb, sy, (R0.y & 2u) ? rowy : 0u, ay, by, az, bz);
            const float4 vyt = sample_line(ly, y0 * 128u + cb, sy, ay, by);
            const float4 vzt = sample_line(lz, z0 * 128u + cb, sz, az, bz);
            float4 f = mul44(vxy, vxz);                          // 1 * v0 * v1 ... in the reference's order
            f = mul44(f, vxt);
            f = mul44(f, vyz);
            f = mul44(f, vyt);
            f = mul44(f, vzt);
            *reinterpret_cast<float4*>(tile + gl * kTileStride + 32 * lvl + 4 * c) = f;
            if (feat_save) {
                const int g = __shfl(g_mine, gl);
                if (g >= 0) *reinterpret_cast<float4*>(feat_save + (size_t)g * kHid + 32 * lvl + 4 * c) = f;
            }
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

__global__ void __launch_bounds__(64 * kFieldWaves)
deform_field_fwd_kernel(HexArgs a, LineTab lt, MlpDev m, int tiles, const float* __restrict__ lines, const float* __restrict__ xyz,
                        const float* __restrict__ scaling, const float* __restrict__ rotation, const float* __restrict__ flow,
                        float flow_coef, float* __restrict__ pts, float* __restrict__ scales, float* __restrict__ rots,
                        float* __restrict__ feat_save, float* __restrict__ a0_save, ActOut act)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, col = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6;
    float* tile = lds + kLFieldTotal - (wv + 1) * kWaveFloats;
    uint4* rec = reinterpret_cast<uint4*>(tile + kTileFloats);
    const int P = a.P;
    const int t_begin = (int)((long long)tiles * blockIdx.x / gridDim.x), t_end = (int)((long long)tiles * (blockIdx.x + 1) / gridDim.x);
    const int t_first = t_begin + wv, t_step = kFieldWaves;
    auto gaussian_of = [&](int t) {
        const int gi = t * 32 + col;
        return gi < P ? (a.order ? (int)a.order[gi] : gi) : -1;
    };
    WeightRegs wr;
    weights_issue(m, wr);
    int g = -1;
    if (t_first < t_end) {
        g = gaussian_of(t_first);
        gather_tile(a, lt, lines, xyz, g, tile, rec, feat_save, lane);
    }
    weights_commit(m, wr, lds);
    __syncthreads();
    for (int t = t_first; t < t_end; t += t_step) {
        const bool ok = g >= 0;
        f32x16 a0[2];
        {
            f32x16 x[2];
            load_tile(tile, col, h, x);
            init_bias(lds + kLB, a0, h);
            layer64<false>(lds + kLW, x, a0, col, h);
        }
        relu_tile(a0);
        if (a0_save) store_feat(a0_save, g, ok, h, a0);
#pragma nounroll
        for (int head = 0; head < 3; head++) {
            f32x16 h1[2];
            init_bias(lds + kLB + (1 + head) * kHid, h1, h);
            layer64<false>(lds + kLW + (1 + head) * kWFloats, a0, h1, col, h);
            relu_tile(h1);
            float o[4];
            out_layer(lds + kLW2 + head * 4 * kHid, lds + kLB2 + head * 4, h1, h, o);
            if (h == 0 && ok) {
                if (head == 0) {
#pragma unroll
                    for (int k = 0; k < 3; k++) pts[3 * g + k] = xyz[3 * g + k] + (o[k] + flow_coef * flow[3 * g + k]);
                } else if (head == 1) {
                    float s3[3];
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        s3[k] = scaling[3 * g + k] + o[k];
                        scales[3 * g + k] = s3[k];
                    }
                    if (act.scales) {
#pragma unroll
                        for (int k = 0; k < 3; k++) act.scales[3 * g + k] = expf(s3[k]);
                    }
                } else {
                    float4 q = *reinterpret_cast<const float4*>(rotation + 4 * g);
                    q = make_float4(q.x + o[0], q.y + o[1], q.z + o[2], q.w + o[3]);
                    *reinterpret_cast<float4*>(rots + 4 * g) = q;
                    if (act.rots) {
                        const float n = mom_quat_norm(q.x, q.y, q.z, q.w);
                        *reinterpret_cast<float4*>(act.rots + 4 * g) = make_float4(q.x / n, q.y / n, q.z / n, q.w / n);
                    }
                    if (act.opacity) act.opacity[g] = mom_sigmoid(act.opacity_raw[g]);
                }
            }
        }
        // next tile's features (the other wave of this SIMD is somewhere in its MFMAs meanwhile)
        const int tn = t + t_step;
        if (tn < t_end) {
            g = gaussian_of(tn);
            gather_tile(a, lt, lines, xyz, g, tile, rec, feat_save, lane);
        }
    }
}

}  // namespace

static int line_table(const MomHexPlane* hp, LineTab* lt)
{
    unsigned off = 0;
    for (int l = 0; l < 4; l++)
        for (int k = 0; k < 3; k++) {
            lt->off[l][k] = off;
            if (l < hp->levels) off += (unsigned)hp->res[l][k] * 32u;
        }
    return (int)off;
}

extern "C" size_t mom_deform_field_scratch_bytes(const MomHexPlane* hp)
{
    if (!hp) return MOM_ALIGN;
    LineTab lt;
    return (size_t)line_table(hp, &lt) * sizeof(float) + MOM_ALIGN;
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
    // one workgroup per CU (the weights and eight B-operand tiles fill its LDS); a small problem is spread over the CUs
    const int blocks = tiles < 256 * kFieldWaves ? (tiles + kFieldWaves - 1) / kFieldWaves : 256;
    static bool attr_set = false;
    const size_t lds_bytes = sizeof(float) * kLFieldTotal;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(deform_field_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes) != hipSuccess)
            return MOM_ELAUNCH;
        attr_set = true;
    }
    const ActOut act = {scales_act, rots_act, opacity_act, opacity_raw};
    hipLaunchKernelGGL(deform_field_fwd_kernel, dim3(blocks), dim3(64 * kFieldWaves), lds_bytes, s, a, lt, d, tiles, lines, xyz, scaling,
                       rotation, scene_flow, flow_coef, pts, scales, rots, feat_save, a0_save, act);
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}
