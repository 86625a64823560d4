// Per-Gaussian projection for the tile rasterizer (forward), gfx950.
//
// Replaces preprocessCUDA<3> + computeCov3D + computeCov2D + computeColorFromSH
// (reference forward.cu:20-256, auxiliary.h:41-164).  THIS FILE IS COMPILED WITH
// -ffp-contract=off: radii, tile rectangles and the depth sort keys must be
// bit-identical to an IEEE evaluation of the reference's expressions in source
// order (fp64 ndc2Pix, truncating casts, un-normalised quaternion), and they are.
#include "raster_bin_dev.h"

namespace {

struct M3 {  // column-major 3x3, m[c][r]; products summed left to right like glm (type_mat3x3.inl:486-519)
    float m[3][3];
};
__device__ __forceinline__ M3 mul(const M3& A, const M3& B)
{
    M3 R;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int r = 0; r < 3; r++)
            R.m[c][r] = A.m[0][r] * B.m[c][0] + A.m[1][r] * B.m[c][1] + A.m[2][r] * B.m[c][2];
    return R;
}
__device__ __forceinline__ M3 transpose(const M3& A)
{
    M3 R;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int r = 0; r < 3; r++)
            R.m[c][r] = A.m[r][c];
    return R;
}

__device__ __forceinline__ float ndc2pix(float v, int S) { return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5); }

__constant__ float kSH_C0 = 0.28209479177387814f;
__constant__ float kSH_C1 = 0.4886025119029199f;
__constant__ float kSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                                0.5462742152960396f};
__constant__ float kSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

struct PreArgs {
    int P, D, M, W, H, gx, gy;
    const float *means3D, *shs, *shs_rest, *colors_precomp, *opacities, *scales, *rotations, *cov3D_precomp;
    const float *view, *proj, *cam;  // device pointers, [16] [16] [3]
    float scale_modifier, tan_fovx, tan_fovy, focal_x, focal_y;
    // HIST: the tile histogram of the binning rides in this kernel (rows [ry0, ry1) of tiles, cull as MomRasterArgs.keep_all_tiles)
    int ry0, ry1, cull;
    uint32_t* tile_counts;
    unsigned long long* reach;
};

// STAGED: the higher-order SH coefficients of the workgroup's 256 Gaussians -- (M-1)*3 floats each, contiguous in memory --
// are copied to LDS with coalesced 16-byte loads and each thread then reads its own row from there (row stride odd: no
// bank conflicts).  Read in place, a thread's 180-byte row makes every load instruction touch 64 different cache lines.
// HIST: after the projection every wave decides which (splat, tile) instances of its 64 Gaussians are binned (decide_instances,
// raster_bin_dev.h: the reach test of the tile cull), counts them per tile in an LDS histogram behind the SH rows, leaves the
// decisions as a 64-bit mask per Gaussian and the workgroup flushes its histogram with one global atomic per non-empty tile --
// what tile_hist_kernel did in a launch of its own (27 us at 200 k Gaussians, most of it waiting for the records this kernel
// still holds in registers).  The caller clears the header and the counters with a fill command in front of this kernel.
template <bool STAGED, bool HIST>
__global__ void __launch_bounds__(256) preprocess_fwd_kernel(PreArgs a, int* __restrict__ radii, float4* __restrict__ rec,
                                                            float* __restrict__ cov3Ds, uchar4* __restrict__ clamped,
                                                            uint32_t* __restrict__ zero_words, int n_zero, int sh_floats)
{
    extern __shared__ float s_sh[];
    uint32_t* s_cnt = reinterpret_cast<uint32_t*>(s_sh + sh_floats);
    const int idx = blockIdx.x * 256 + threadIdx.x;
    // the image scratch's header and tile counters, which the binning kernels behind this one accumulate into, are cleared
    // here instead of by a fill launch of their own (not with HIST: this kernel then accumulates into them itself)
    if (!HIST)
        for (int i = idx; i < n_zero; i += gridDim.x * 256) zero_words[i] = 0u;
    const int n_tiles = a.gx * a.gy;
    if (HIST)
        for (int t = threadIdx.x; t < n_tiles; t += 256) s_cnt[t] = 0u;
    const int sh_stride = (a.M - 1) * 3;
    // Everything this thread reads of its Gaussian, asked for in ONE go and before the SH rows are staged.  Read where it is used,
    // each item sat behind the cull test before it -- position -> scale + rotation -> DC colour, one channel at a time -> opacity:
    // seven dependent round trips in a kernel whose workgroups live as long as their longest chain.
    const bool live = idx < a.P;
    const int gi = live ? idx : 0;                 // lanes past the end (HIST keeps them for the ballots) read Gaussian 0 and discard
    float px = a.means3D[3 * gi], py = a.means3D[3 * gi + 1], pz = a.means3D[3 * gi + 2];
    float in_c3[6], in_s[3], in_q[4], in_col[3];
    {
        // (no branch around either form of the covariance's inputs -- the values would be waited for where the branches join:
        // the form that is absent reads the view matrix, 16 floats that are always there, and is never looked at)
        const float* __restrict__ p6 = a.cov3D_precomp ? a.cov3D_precomp + 6 * (size_t)gi : a.view;
        const float* __restrict__ p3 = a.cov3D_precomp ? a.view : a.scales + 3 * (size_t)gi;
        const float* __restrict__ p4 = a.cov3D_precomp ? a.view : a.rotations + 4 * (size_t)gi;
#pragma unroll
        for (int i = 0; i < 6; i++) in_c3[i] = p6[i];
#pragma unroll
        for (int i = 0; i < 3; i++) in_s[i] = p3[i];
#pragma unroll
        for (int i = 0; i < 4; i++) in_q[i] = p4[i];
        // the DC coefficient (one [P,M,3] tensor, or DC and rest stored apart) or the caller's colour
        const float* __restrict__ col = a.colors_precomp ? a.colors_precomp + 3 * (size_t)gi : a.shs + (size_t)gi * (a.shs_rest ? 1 : a.M) * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) in_col[c] = col[c];
    }
    float in_opacity = a.opacities[gi];
    if (STAGED) {
        const int block0 = blockIdx.x * 256;
        const int n = min(256, a.P - block0) * sh_stride;
        const float* __restrict__ src = a.shs_rest + (size_t)block0 * sh_stride;
        const int n4 = n >> 2;
        // (all of a thread's pieces in flight at once -- at most 12 with rows of 45 floats; as a loop they were taken in three
        // batches, each waited for before the next was asked for.  The index is clamped, not tested: a load under a test is
        // waited for where the test ends)
        float4 piece[12];
#pragma unroll
        for (int k = 0; k < 12; k++) piece[k] = reinterpret_cast<const float4*>(src)[min((int)threadIdx.x + 256 * k, n4 - 1)];
        // (and all of them are values of the straight-line code: used only under the tests below, each load is moved into its test)
        asm("" : "+v"(piece[0].x), "+v"(piece[0].y), "+v"(piece[0].z), "+v"(piece[0].w), "+v"(piece[1].x), "+v"(piece[1].y), "+v"(piece[1].z), "+v"(piece[1].w), "+v"(piece[2].x), "+v"(piece[2].y), "+v"(piece[2].z), "+v"(piece[2].w), "+v"(piece[3].x), "+v"(piece[3].y), "+v"(piece[3].z), "+v"(piece[3].w), "+v"(piece[4].x), "+v"(piece[4].y), "+v"(piece[4].z), "+v"(piece[4].w), "+v"(piece[5].x), "+v"(piece[5].y), "+v"(piece[5].z), "+v"(piece[5].w));
        asm("" : "+v"(piece[6].x), "+v"(piece[6].y), "+v"(piece[6].z), "+v"(piece[6].w), "+v"(piece[7].x), "+v"(piece[7].y), "+v"(piece[7].z), "+v"(piece[7].w), "+v"(piece[8].x), "+v"(piece[8].y), "+v"(piece[8].z), "+v"(piece[8].w), "+v"(piece[9].x), "+v"(piece[9].y), "+v"(piece[9].z), "+v"(piece[9].w), "+v"(piece[10].x), "+v"(piece[10].y), "+v"(piece[10].z), "+v"(piece[10].w), "+v"(piece[11].x), "+v"(piece[11].y), "+v"(piece[11].z), "+v"(piece[11].w));
#pragma unroll
        for (int k = 0; k < 12; k++) {
            const int i = threadIdx.x + 256 * k;
            if (i < n4) reinterpret_cast<float4*>(s_sh)[i] = piece[k];
        }
        for (int i = 4 * n4 + threadIdx.x; i < n; i += 256) s_sh[i] = src[i];
        __syncthreads();
    } else if (HIST) {
        __syncthreads();
    }
    // All of it complete HERE, not one by one behind the tests below.  (Not `asm volatile`: with no memory operands that still counts
    // as a possible store, and the uniform loads of the view / projection matrices behind it become vector loads -- three more
    // round trips; an asm with outputs only makes the values opaque.)
    asm("" : "+v"(px), "+v"(py), "+v"(pz), "+v"(in_c3[0]), "+v"(in_c3[1]), "+v"(in_c3[2]), "+v"(in_c3[3]), "+v"(in_c3[4]), "+v"(in_c3[5]),
        "+v"(in_s[0]), "+v"(in_s[1]), "+v"(in_s[2]), "+v"(in_q[0]), "+v"(in_q[1]), "+v"(in_q[2]), "+v"(in_q[3]), "+v"(in_col[0]),
        "+v"(in_col[1]), "+v"(in_col[2]), "+v"(in_opacity));
    if (!HIST && idx >= a.P) return;
    const float* __restrict__ view = a.view;
    const float* __restrict__ proj = a.proj;
    const float* __restrict__ cam = a.cam;

    int radius = 0;
    uint32_t tiles = 0;
    float4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = {0.f, 0.f, 0.f, 0.f}, r2 = {0.f, 0.f, 0.f, 0.f};
    uchar4 cl = {0, 0, 0, 0};
    int hx0 = 0, hy0 = 0, hx1 = 0, hy1 = 0;        // the rectangle of tiles (HIST)
    // near cull: keep iff p_view.z > 0.2
    const float vx = view[0] * px + view[4] * py + view[8] * pz + view[12];
    const float vy = view[1] * px + view[5] * py + view[9] * pz + view[13];
    const float vz = view[2] * px + view[6] * py + view[10] * pz + view[14];
    do {
        if (!live || vz <= 0.2f) break;
        const float hx = proj[0] * px + proj[4] * py + proj[8] * pz + proj[12];
        const float hy = proj[1] * px + proj[5] * py + proj[9] * pz + proj[13];
        const float hw = proj[3] * px + proj[7] * py + proj[11] * pz + proj[15];
        const float p_w = 1.0f / (hw + 0.0000001f);
        const float projx = hx * p_w, projy = hy * p_w;

        float c3[6];
        if (a.cov3D_precomp != nullptr) {
#pragma unroll
            for (int i = 0; i < 6; i++) c3[i] = in_c3[i];
        } else {
            const float mod = a.scale_modifier;
            M3 S = {{{1.f, 0.f, 0.f}, {0.f, 1.f, 0.f}, {0.f, 0.f, 1.f}}};
            S.m[0][0] = mod * in_s[0];
            S.m[1][1] = mod * in_s[1];
            S.m[2][2] = mod * in_s[2];
            const float r = in_q[0], x = in_q[1], y = in_q[2], z = in_q[3];
            M3 Rm = {{{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
                      {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
                      {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}}};
            M3 Mm = mul(S, Rm);
            M3 Sg = mul(transpose(Mm), Mm);
            c3[0] = Sg.m[0][0]; c3[1] = Sg.m[0][1]; c3[2] = Sg.m[0][2];
            c3[3] = Sg.m[1][1]; c3[4] = Sg.m[1][2]; c3[5] = Sg.m[2][2];
            if (cov3Ds) {
#pragma unroll
                for (int i = 0; i < 6; i++) cov3Ds[6 * gi + i] = c3[i];
            }
        }

        // EWA 2D covariance
        float tx = vx, ty = vy;
        const float tz = vz;
        const float limx = 1.3f * a.tan_fovx, limy = 1.3f * a.tan_fovy;
        const float txtz = tx / tz, tytz = ty / tz;
        tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
        ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
        M3 J = {{{a.focal_x / tz, 0.0f, -(a.focal_x * tx) / (tz * tz)},
                 {0.0f, a.focal_y / tz, -(a.focal_y * ty) / (tz * tz)},
                 {0.f, 0.f, 0.f}}};
        M3 Wm = {{{view[0], view[4], view[8]}, {view[1], view[5], view[9]}, {view[2], view[6], view[10]}}};
        M3 T = mul(Wm, J);
        M3 Vrk = {{{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}}};
        M3 cov = mul(mul(transpose(T), transpose(Vrk)), T);
        const float cxx = cov.m[0][0] + 0.3f, cxy = cov.m[0][1], cyy = cov.m[1][1] + 0.3f;

        const float det = cxx * cyy - cxy * cxy;
        if (det == 0.0f) break;
        const float det_inv = 1.f / det;
        const float conx = cyy * det_inv, cony = -cxy * det_inv, conz = cxx * det_inv;

        const float mid = 0.5f * (cxx + cyy);
        const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
        const float pix = ndc2pix(projx, a.W), piy = ndc2pix(projy, a.H);
        int x0, y0, x1, y1;
        mom_get_rect(pix, piy, (int)my_radius, a.gx, a.gy, x0, y0, x1, y1);
        const uint32_t cnt = (uint32_t)(x1 - x0) * (uint32_t)(y1 - y0);
        if (cnt == 0) break;

        float cr, cg, cb;
        if (a.colors_precomp == nullptr) {
            float dx = px - cam[0], dy = py - cam[1], dz = pz - cam[2];
            const float len = sqrtf(dx * dx + dy * dy + dz * dz);
            dx = dx / len; dy = dy / len; dz = dz / len;
            // coefficient i >= 1 of this Gaussian: one [P,M,3] tensor, or DC and rest stored apart
            const float* sh = STAGED ? s_sh + threadIdx.x * sh_stride - 3
                                     : (a.shs_rest ? a.shs_rest + (size_t)gi * (a.M - 1) * 3 - 3 : a.shs + (size_t)gi * a.M * 3);   // sh[3*i+c] valid for i >= 1
            float res[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                float v = kSH_C0 * in_col[c];
                if (a.D > 0) {
                    const float x = dx, y = dy, z = dz;
                    v = v - kSH_C1 * y * sh[3 + c] + kSH_C1 * z * sh[6 + c] - kSH_C1 * x * sh[9 + c];
                    if (a.D > 1) {
                        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                        v = v + kSH_C2[0] * xy * sh[12 + c] + kSH_C2[1] * yz * sh[15 + c] +
                            kSH_C2[2] * (2.0f * zz - xx - yy) * sh[18 + c] + kSH_C2[3] * xz * sh[21 + c] +
                            kSH_C2[4] * (xx - yy) * sh[24 + c];
                        if (a.D > 2) {
                            v = v + kSH_C3[0] * y * (3.0f * xx - yy) * sh[27 + c] + kSH_C3[1] * xy * z * sh[30 + c] +
                                kSH_C3[2] * y * (4.0f * zz - xx - yy) * sh[33 + c] +
                                kSH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[36 + c] +
                                kSH_C3[4] * x * (4.0f * zz - xx - yy) * sh[39 + c] + kSH_C3[5] * z * (xx - yy) * sh[42 + c] +
                                kSH_C3[6] * x * (xx - 3.0f * yy) * sh[45 + c];
                        }
                    }
                }
                v += 0.5f;
                res[c] = v;
            }
            cl.x = res[0] < 0; cl.y = res[1] < 0; cl.z = res[2] < 0;
            cr = fmaxf(res[0], 0.0f); cg = fmaxf(res[1], 0.0f); cb = fmaxf(res[2], 0.0f);
        } else {
            cr = in_col[0]; cg = in_col[1]; cb = in_col[2];
        }
        radius = (int)my_radius;
        tiles = cnt;
        hx0 = x0; hy0 = y0; hx1 = x1; hy1 = y1;
        r0 = make_float4(pix, piy, vz, __uint_as_float(tiles));
        r1 = make_float4(conx, cony, conz, in_opacity);
        r2 = make_float4(cr, cg, cb, __int_as_float(radius));
    } while (0);

    if (live) {
        radii[idx] = radius;
        rec[3 * (size_t)idx + 0] = r0;
        rec[3 * (size_t)idx + 1] = r1;
        rec[3 * (size_t)idx + 2] = r2;
        if (clamped) clamped[idx] = cl;
    }
    if (HIST) {
        // exactly what load_rect + tile_hist_kernel made of the stored record (raster_binning.hip): the rectangle cut to this
        // launch's tile rows, the reach parameters from the conic and the opacity
        hy0 = max(hy0, a.ry0);
        hy1 = min(hy1, a.ry1);
        if (hy1 <= hy0 || radius <= 0) hx0 = hy0 = hx1 = hy1 = 0;
        Reach rc = Reach{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0};
        if (a.cull && radius > 0 && r1.x > 0.f && r1.z > 0.f)
            rc = Reach{r0.x, r0.y, r1.x, r1.y, r1.z, mom_power_bound(r1.w), __builtin_amdgcn_rcpf(r1.x), __builtin_amdgcn_rcpf(r1.z), 1};
        const uint64_t mask = decide_instances(hx0, hy0, hx1, hy1, a.gx, rc, [&](int tile) { atomicAdd(&s_cnt[tile], 1u); });
        if (live) a.reach[idx] = mask;
        __syncthreads();
        for (int t = threadIdx.x; t < n_tiles; t += 256) {
            const uint32_t n = s_cnt[t];
            if (n) atomicAdd(&a.tile_counts[t], n);
        }
    }
}

__global__ void mark_visible_kernel(int P, const float* __restrict__ means, const float* __restrict__ view,
                                    uint8_t* __restrict__ present)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    const float vz = view[2] * means[3 * idx] + view[6] * means[3 * idx + 1] + view[10] * means[3 * idx + 2] + view[14];
    present[idx] = vz <= 0.2f ? 0 : 1;
}

}  // namespace

// host launcher (called from raster_api.hip).  hist_counts / hist_reach non-null: the tile histogram rides in the projection kernel
// (the caller has cleared the counters); *did_hist tells the binning whether it still has to run tile_hist_kernel.
int mom_launch_preprocess_fwd(const MomRasterArgs* a, const GeomView& g, int* radii, uint32_t* zero_words, int n_zero,
                              uint32_t* hist_counts, bool* did_hist, hipStream_t s)
{
    PreArgs p;
    p.P = a->P; p.D = a->D; p.M = a->M; p.W = a->W; p.H = a->H;
    p.gx = (a->W + MOM_TILE - 1) / MOM_TILE;
    p.gy = (a->H + MOM_TILE - 1) / MOM_TILE;
    p.means3D = a->means3D; p.shs = a->shs; p.shs_rest = a->shs_rest; p.colors_precomp = a->colors_precomp; p.opacities = a->opacities;
    p.scales = a->scales; p.rotations = a->rotations; p.cov3D_precomp = a->cov3D_precomp;
    p.scale_modifier = a->scale_modifier; p.tan_fovx = a->tan_fovx; p.tan_fovy = a->tan_fovy;
    // rasterizer_impl.cu:223-224
    p.focal_y = a->H / (2.0f * a->tan_fovy);
    p.focal_x = a->W / (2.0f * a->tan_fovx);
    p.view = a->viewmatrix; p.proj = a->projmatrix; p.cam = a->campos;
    mom_tile_rows(a, p.gy, &p.ry0, &p.ry1);
    p.cull = a->keep_all_tiles ? 0 : 1;
    p.tile_counts = hist_counts;
    p.reach = g.reach;
    const int blocks = (a->P + 255) / 256;
    // staged SH rows: DC and rest stored apart, colours from SH above degree 0, an odd row length, 16-byte aligned rows
    const int sh_stride = (a->M - 1) * 3;
    const bool staged = a->shs_rest && !a->colors_precomp && a->D > 0 && (sh_stride & 1) && sh_stride <= 45 &&
                        ((uintptr_t)a->shs_rest & 15) == 0;
    const int tiles = p.gx * p.gy;
    const int sh_floats = staged ? 256 * sh_stride : 0;
    // the histogram needs its counters in LDS beside the SH rows (45 KB): images up to 16 k tiles (e.g. 2048 x 2048)
    const bool hist = hist_counts != nullptr && tiles <= kMaxLdsTiles && (size_t)(sh_floats + tiles) * 4 <= 64 * 1024 + 46 * 1024;
    if (did_hist) *did_hist = hist;
    const size_t lds = (size_t)(sh_floats + (hist ? tiles : 0)) * 4;
    MomProfScope ps(MOM_P_PRE_FWD, s);
    float* cov = a->forward_only ? nullptr : g.cov3D;
    uchar4* cl = a->forward_only ? nullptr : g.clamped;
    if (hist) {
        // the header and the tile counters: one fill command in front (this kernel adds into the counters itself)
        if (hipMemsetAsync(zero_words, 0, (size_t)n_zero * 4, s) != hipSuccess) return MOM_ELAUNCH;
        static bool attr_set = false;
        if (!attr_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(preprocess_fwd_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    112 * 1024) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(preprocess_fwd_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    112 * 1024) != hipSuccess)
                return MOM_ELAUNCH;
            attr_set = true;
        }
        if (staged)
            hipLaunchKernelGGL((preprocess_fwd_kernel<true, true>), dim3(blocks), dim3(256), lds, s, p, radii, g.rec, cov, cl, zero_words, n_zero, sh_floats);
        else
            hipLaunchKernelGGL((preprocess_fwd_kernel<false, true>), dim3(blocks), dim3(256), lds, s, p, radii, g.rec, cov, cl, zero_words, n_zero, sh_floats);
    } else if (staged) {
        hipLaunchKernelGGL((preprocess_fwd_kernel<true, false>), dim3(blocks), dim3(256), lds, s, p, radii, g.rec, cov, cl, zero_words, n_zero, sh_floats);
    } else {
        hipLaunchKernelGGL((preprocess_fwd_kernel<false, false>), dim3(blocks), dim3(256), 0, s, p, radii, g.rec, cov, cl, zero_words, n_zero, sh_floats);
    }
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}

int mom_launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t s)
{
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, view, present);
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}
