// Internal definitions shared by the HIP translation units of libmom4d.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/mom4d.h"

#define MOM_WAVE 64
#define MOM_ALIGN 256

// ---- private scratch layouts -------------------------------------------------
// geom (per Gaussian):
//   rec[P][3] float4 : {x, y, depth, tiles_touched bits} {conic.x, conic.y, conic.z, opacity} {r, g, b, radius bits}
//   cov3D[P][6] float, clamped[P] uchar4, gacc[P][MOM_GACC_FLOATS] float (backward accumulators), reach[P] u64 (tile cull: bit i =
//   tile i of the splat's rectangle is binned; written by tile_hist, read by tile_scatter)
// image: hdr[64] u32, tile_counts[tiles] u32, tile_cursor[tiles] u32, ranges[tiles] uint2, n_contrib[H*W] u32,
//        final_T[H*W] f32, tile_order[tiles] u32
// binning: keys[cap] u64, point_list[cap] u32   (sized once the instance count is known)
struct GeomView {
    float4* rec;
    float* cov3D;
    uchar4* clamped;
    float* gacc;
    unsigned long long* reach;
};
struct ImageView {
    uint32_t* hdr;  // [0] = num_rendered, [1] = status bits
    uint32_t* tile_counts;
    uint32_t* tile_cursor;
    uint2* ranges;
    uint32_t* n_contrib;
    float* final_T;
    uint32_t* tile_order;   // [tiles]: the launch's tiles, heaviest first (tile_scan) -- the compositing kernels' workgroup i takes tile_order[i]
};
struct BinView {
    uint64_t* keys;
    uint32_t* point_list;
};

static inline size_t mom_align_up(size_t x) { return (x + MOM_ALIGN - 1) & ~(size_t)(MOM_ALIGN - 1); }

static inline size_t geom_view(char* base, int P, GeomView* v)
{
    size_t off = 0;
    size_t o_rec = off; off = mom_align_up(off + (size_t)P * 48);
    size_t o_cov = off; off = mom_align_up(off + (size_t)P * 24);
    size_t o_cl = off; off = mom_align_up(off + (size_t)P * 4);
    size_t o_ga = off; off = mom_align_up(off + (size_t)P * MOM_GACC_FLOATS * 4);
    size_t o_re = off; off = mom_align_up(off + (size_t)P * 8);
    if (v) {
        v->rec = (float4*)(base + o_rec);
        v->cov3D = (float*)(base + o_cov);
        v->clamped = (uchar4*)(base + o_cl);
        v->gacc = (float*)(base + o_ga);
        v->reach = (unsigned long long*)(base + o_re);
    }
    return off + MOM_ALIGN;
}
static inline size_t image_view(char* base, int W, int H, ImageView* v)
{
    const size_t tiles = (size_t)((W + MOM_TILE - 1) / MOM_TILE) * ((H + MOM_TILE - 1) / MOM_TILE);
    const size_t N = (size_t)W * H;
    size_t off = 0;
    size_t o_h = off; off = mom_align_up(off + 64 * 4);
    size_t o_c = off; off = mom_align_up(off + tiles * 4);
    size_t o_u = off; off = mom_align_up(off + tiles * 4);
    size_t o_r = off; off = mom_align_up(off + tiles * 8);
    size_t o_n = off; off = mom_align_up(off + N * 4);
    size_t o_t = off; off = mom_align_up(off + N * 4);
    size_t o_o = off; off = mom_align_up(off + tiles * 4);
    if (v) {
        v->tile_order = (uint32_t*)(base + o_o);
        v->hdr = (uint32_t*)(base + o_h);
        v->tile_counts = (uint32_t*)(base + o_c);
        v->tile_cursor = (uint32_t*)(base + o_u);
        v->ranges = (uint2*)(base + o_r);
        v->n_contrib = (uint32_t*)(base + o_n);
        v->final_T = (float*)(base + o_t);
    }
    return off + MOM_ALIGN;
}
static inline size_t bin_view(char* base, size_t cap, BinView* v)
{
    // point_list FIRST: it is all the backward reads of this buffer, and at offset 0 its position does not depend on the capacity
    // the forward was given -- the drop-in's exact mode hands the backward the true instance count as `capacity` while the buffer
    // was laid out for a larger one (diff_gaussian_rasterization/_C.py)
    size_t off = 0;
    size_t o_p = off; off = mom_align_up(off + cap * 4);
    size_t o_k = off; off = mom_align_up(off + cap * 8);
    if (v) {
        v->keys = (uint64_t*)(base + o_k);
        v->point_list = (uint32_t*)(base + o_p);
    }
    return off + MOM_ALIGN;
}
static inline char* mom_align_ptr(void* p)
{
    return (char*)(((uintptr_t)p + MOM_ALIGN - 1) & ~(uintptr_t)(MOM_ALIGN - 1));
}

// flags of the events that order two streams of one device (stream_order.hip)
unsigned mom_order_event_flags();

// Local tile rows of a (possibly tile-row sharded) launch: [r0, r1) within [0, gy).
static inline void mom_tile_rows(const MomRasterArgs* a, int gy, int* r0, int* r1)
{
    if (a->tile_row0 == 0 && a->tile_row1 == 0) { *r0 = 0; *r1 = gy; return; }
    *r0 = a->tile_row0 < 0 ? 0 : (a->tile_row0 > gy ? gy : a->tile_row0);
    *r1 = a->tile_row1 < *r0 ? *r0 : (a->tile_row1 > gy ? gy : a->tile_row1);
}

#define MOM_CHECK_LAUNCH(a, s)                                         \
    do {                                                               \
        hipError_t e__ = hipGetLastError();                            \
        if (e__ != hipSuccess) return MOM_ELAUNCH;                     \
        if ((a)->debug) {                                              \
            e__ = hipStreamSynchronize((hipStream_t)(s));              \
            if (e__ != hipSuccess) return MOM_ELAUNCH;                 \
        }                                                              \
    } while (0)

// per-kernel timing slots (profile.hip); no-ops unless enabled through mom_profile_enable
enum { MOM_P_PRE_FWD = 0, MOM_P_HIST, MOM_P_SCAN, MOM_P_SCATTER, MOM_P_SORT, MOM_P_RENDER_FWD, MOM_P_RENDER_BWD, MOM_P_PRE_BWD,
       MOM_P_HEX_FWD, MOM_P_HEX_BWD, MOM_P_ADAM, MOM_P_L1, MOM_P_REG, MOM_P_MLP_FWD, MOM_P_MLP_BWD };
void mom_prof_begin(int slot, hipStream_t s);
void mom_prof_end(int slot, hipStream_t s);
struct MomProfScope {
    int slot; hipStream_t s;
    MomProfScope(int slot_, hipStream_t s_) : slot(slot_), s(s_) { mom_prof_begin(slot, s); }
    ~MomProfScope() { mom_prof_end(slot, s); }
};

#ifdef __HIPCC__
// ---- device helpers ----------------------------------------------------------
// Tile rectangle of a splat: truncating casts then clamp (reference auxiliary.h:46-56).
__device__ __forceinline__ void mom_get_rect(float px, float py, int max_radius, int gx, int gy, int& x0, int& y0,
                                             int& x1, int& y1)
{
    x0 = min(gx, max(0, (int)((px - max_radius) / MOM_TILE)));
    y0 = min(gy, max(0, (int)((py - max_radius) / MOM_TILE)));
    x1 = min(gx, max(0, (int)((px + max_radius + MOM_TILE - 1) / MOM_TILE)));
    y1 = min(gy, max(0, (int)((py + max_radius + MOM_TILE - 1) / MOM_TILE)));
}

// exp(x) for x <= 0 on the transcendental unit: v_exp_f32(x*log2e) with the
// rounding error of the product fed back (about 1.5 ulp; both render passes use
// this one function so they take identical alpha branches).
__device__ __forceinline__ float mom_exp(float x)
{
    const float L2E = 1.4426950408889634f;
    const float L2E_LO = 1.9259629911266175e-8f;  // log2(e) - (float)log2(e)
    float t = x * L2E;
    float e = __builtin_fmaf(x, L2E, -t);
    e = __builtin_fmaf(x, L2E_LO, e);
    float r = __builtin_amdgcn_exp2f(t);
    return __builtin_fmaf(r * 0.6931471805599453f, e, r);
}

// ---- which pixels can a splat reach? ----------------------------------------------------------------------------------
// A splat is composited at a pixel only if alpha = min(0.99, opacity * exp(power)) >= 1/255, i.e. only if
// power >= ln(1 / (255 opacity)).  mom_power_bound is that bound lowered by a margin that dwarfs the rounding of logf, of
// the product and of mom_exp (all below 1e-6 here).  The exponent is concave, so its maximum over a rectangle of pixel
// centres is 0 if the splat's centre is inside and otherwise lies on an edge, at the clamped stationary point of the
// edge's 1-D quadratic: mom_rect_reach evaluates that on the continuous rectangle, with a second margin, and therefore
// keeps every (splat, pixel) pair the exact per-pixel tests of the compositing kernels could accept.  Those tests still
// run on whatever is kept, so culling by this predicate never changes a result.  Anything degenerate (non-positive conic
// diagonal, NaN anywhere) counts as reachable; written as !(best < bound) so that a NaN falls through.
__device__ __forceinline__ float mom_power_bound(float opacity) { return -logf(255.0f * opacity) - 1e-3f; }
__device__ __forceinline__ float mom_edge_max(float fixed, float lo, float hi, float q_fixed, float q_free, float inv_q_free, float b)
{
    // max over t in [lo, hi] of  -0.5 (q_fixed fixed^2 + q_free t^2) - b fixed t.  The stationary point uses a hardware
    // reciprocal (1 ulp): the quadratic is flat there, so its error is second order and far inside the caller's margin.
    // Contraction off: the tile histogram and the tile scatter must take the same decision for the same (splat, tile).
#pragma clang fp contract(off)
    const float t = fminf(fmaxf(-b * fixed * inv_q_free, lo), hi);
    // The three terms can be large and cancel (a long thin splat hundreds of pixels away: each ~1e5, sum ~ -5): the fp32 error
    // of the sum scales with their magnitude, not with the result, so the returned maximum is raised by a bound on that error
    // (a few ulp of the summed magnitudes) on top of the caller's absolute margin -- the cull stays on the safe side for any
    // anisotropy.
    const float p = q_fixed * fixed * fixed, q = q_free * t * t, r = b * fixed * t;
    return (-0.5f * (p + q) - r) + 1e-6f * (0.5f * (p + q) + fabsf(r));
}
// centre (cx, cy), conic (a, b, c), bound = mom_power_bound(opacity), inv_a = 1/a, inv_c = 1/c (hardware reciprocals are
// enough); pixel centres xa..xb by ya..yb.  The caller has checked a > 0 and c > 0.
// Only the edges FACING the centre are evaluated: with p a maximiser on the rectangle and q the centre, concavity gives
// f((1-e) p + e q) >= f(p), so if p were on an edge facing away a step towards q would stay inside and be no worse.  The
// facing x edge is x = clamp(cx, xa, xb) (the line through the centre when cx is inside the columns, also a valid
// candidate), and the same for y; both evaluate to 0 when the centre is inside the rectangle.
__device__ __forceinline__ bool mom_rect_reach(float cx, float cy, float a, float b, float c, float bound, float inv_a, float inv_c,
                                               float xa, float xb, float ya, float yb)
{
#pragma clang fp contract(off)
    const float dxl = cx - xb, dxh = cx - xa;        // dx = cx - px over the rectangle's columns
    const float dyl = cy - yb, dyh = cy - ya;
#ifdef MOM_FOUR_EDGES
    float best;
    if (cx >= xa && cx <= xb && cy >= ya && cy <= yb) {
        best = 0.f;
    } else {
        best = fmaxf(fmaxf(mom_edge_max(dxl, dyl, dyh, a, c, inv_c, b), mom_edge_max(dxh, dyl, dyh, a, c, inv_c, b)),
                     fmaxf(mom_edge_max(dyl, dxl, dxh, c, a, inv_a, b), mom_edge_max(dyh, dxl, dxh, c, a, inv_a, b)));
    }
#else
    const float dxn = cx - fminf(fmaxf(cx, xa), xb), dyn = cy - fminf(fmaxf(cy, ya), yb);
    const float best = fmaxf(mom_edge_max(dxn, dyl, dyh, a, c, inv_c, b), mom_edge_max(dyn, dxl, dxh, c, a, inv_a, b));
#endif
    return !(best < bound - 1e-3f);
}

// The activations of gaussian_renderer.render() (scene/gaussian_model.py:60-75: exp, normalize, sigmoid), in one place so that
// the stand-alone kernel (mom_activations_forward) and the MLP forward's epilogue (mom_deform_forward_activated) round alike.
__device__ __forceinline__ float mom_quat_norm(float x, float y, float z, float w)
{
#pragma clang fp contract(off)
    return fmaxf(sqrtf(x * x + y * y + z * z + w * w), 1e-12f);
}
__device__ __forceinline__ float mom_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ int mom_lane() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
#endif
