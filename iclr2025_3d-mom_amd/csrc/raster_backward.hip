// Per-Gaussian backward of the projection step, gfx950.
//
// Replaces computeCov2DCUDA + preprocessCUDA(backward) + computeColorFromSH(backward)
// + computeCov3D(backward) (reference backward.cu:20-412) as ONE kernel: it consumes
// the per-Gaussian accumulator record gacc[P][MOM_GACC_FLOATS] written by the render backward
// ({dmean2D.x, .y, dconic.x, .y, .w, dopacity, dcolor r, g, b, ddepth}) and writes
// every output gradient exactly once (zeros for culled Gaussians), so none of the
// ten gradient tensors needs a memset (the reference zero-fills them first:
// rasterize_points.cu:154-163).
#include "mom_common.h"

namespace {

struct M3 {
    float m[3][3];  // m[c][r]
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

__constant__ float bSH_C0 = 0.28209479177387814f;
__constant__ float bSH_C1 = 0.4886025119029199f;
__constant__ float bSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                                0.5462742152960396f};
__constant__ float bSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

struct BwdArgs {
    int P, D, M, W, H;
    const float *means3D, *shs, *shs_rest, *scales, *rotations, *cov3D;  // cov3D = precomputed input or geom cov3D
    const float *view, *proj, *cam;
    const int* radii;
    const uchar4* clamped;
    const float* gacc;
    const float4* rec;           // the forward's per-Gaussian record: rec[3 i + 1] = {conic.x, conic.y, conic.z, opacity}
    const float* rots_raw;       // MomRasterGrads.act_rotations_raw: scale / rotation / opacity gradients go out through their activations
    float scale_modifier, tan_fovx, tan_fovy, h_x, h_y;
    int colors_from_sh;
    float *dmeans2D, *dcolors, *dopacity, *dmeans3D, *dcov3D, *dsh, *dsh_rest, *dscales, *drot;
    float *dscales2, *drot2;     // MomRasterGrads.dL_dscales_copy / dL_drotations_copy: second destinations of the same values (or null)
};

// STAGED (DC and rest stored apart, an odd row length): the workgroup's higher-order SH rows are copied to LDS with
// coalesced loads, every thread reads its row there, writes the row's GRADIENT over it once it is done reading, and the
// workgroup stores the gradient rows with coalesced writes.  In place, a thread's 180-byte row makes every load and
// every store instruction touch 64 different cache lines.
template <bool STAGED>
__global__ void __launch_bounds__(256) preprocess_bwd_kernel(BwdArgs a)
{
    extern __shared__ float s_sh[];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int sh_stride = (a.M - 1) * 3;
    const int block0 = blockIdx.x * 256;
    const int n_stage = STAGED ? min(256, a.P - block0) * sh_stride : 0;
    // Everything this thread reads of its Gaussian, asked for in ONE go, before the SH rows are staged and whether or not the frame saw
    // it (read where used, the items queued behind one another: radius -> accumulator record -> conic -> position + covariance ->
    // clamp flags -> rotation + scale, behind a staging loop that waited for each of its eleven pieces in turn -- see the same change
    // in raster_preprocess.hip).  An absent optional input reads the view matrix, 16 floats that are always there.
    const int gi = min(idx, a.P - 1);
    int in_radius = a.radii[gi];
    float ga[10], in_m[3], in_c3[6], in_q[4], in_s[3];
    float4 in_co = a.rec[3 * (size_t)gi + 1];
#pragma unroll
    for (int i = 0; i < 10; i++) ga[i] = a.gacc[(size_t)gi * MOM_GACC_FLOATS + i];
#pragma unroll
    for (int i = 0; i < 3; i++) in_m[i] = a.means3D[3 * gi + i];
#pragma unroll
    for (int i = 0; i < 6; i++) in_c3[i] = a.cov3D[6 * (size_t)gi + i];
    {
        const float* __restrict__ p4 = a.scales ? a.rotations + 4 * (size_t)gi : a.view;
        const float* __restrict__ p3 = a.scales ? a.scales + 3 * (size_t)gi : a.view;
#pragma unroll
        for (int i = 0; i < 4; i++) in_q[i] = p4[i];
#pragma unroll
        for (int i = 0; i < 3; i++) in_s[i] = p3[i];
    }
    float4 in_qraw;
    {
        const float* __restrict__ pr = a.rots_raw ? a.rots_raw + 4 * (size_t)gi : a.view;
        in_qraw = make_float4(pr[0], pr[1], pr[2], pr[3]);
    }
    unsigned in_cl = *reinterpret_cast<const unsigned*>(a.colors_from_sh ? reinterpret_cast<const void*>(a.clamped + gi) : reinterpret_cast<const void*>(a.view));
    if (STAGED) {
        const float* __restrict__ src = a.shs_rest + (size_t)block0 * sh_stride;
        const int n4 = n_stage >> 2;
        // all of a thread's pieces in flight at once (at most 12 with rows of 45 floats); the index is clamped, not tested, and the
        // values are made opaque in the straight-line code: a load under a test, or used only under one, is waited for on the spot
        // (the first statement names one value of every other group of loads, so that all of them are asked for ahead of it)
        float4 piece[12];
#pragma unroll
        for (int k = 0; k < 12; k++) piece[k] = reinterpret_cast<const float4*>(src)[min((int)threadIdx.x + 256 * k, n4 - 1)];
        asm("" : "+v"(piece[0].x), "+v"(piece[0].y), "+v"(piece[0].z), "+v"(piece[0].w), "+v"(piece[1].x), "+v"(piece[1].y), "+v"(piece[1].z), "+v"(piece[1].w), "+v"(piece[2].x), "+v"(piece[2].y), "+v"(piece[2].z), "+v"(piece[2].w), "+v"(piece[3].x), "+v"(piece[3].y), "+v"(piece[3].z), "+v"(piece[3].w), "+v"(piece[4].x), "+v"(piece[4].y), "+v"(piece[4].z), "+v"(piece[4].w), "+v"(piece[5].x), "+v"(piece[5].y), "+v"(piece[5].z), "+v"(piece[5].w), "+v"(piece[11].w), "+v"(in_cl), "+v"(in_s[2]), "+v"(in_c3[5]), "+v"(ga[9]), "+v"(in_m[2]));
        asm("" : "+v"(piece[6].x), "+v"(piece[6].y), "+v"(piece[6].z), "+v"(piece[6].w), "+v"(piece[7].x), "+v"(piece[7].y), "+v"(piece[7].z), "+v"(piece[7].w), "+v"(piece[8].x), "+v"(piece[8].y), "+v"(piece[8].z), "+v"(piece[8].w), "+v"(piece[9].x), "+v"(piece[9].y), "+v"(piece[9].z), "+v"(piece[9].w), "+v"(piece[10].x), "+v"(piece[10].y), "+v"(piece[10].z), "+v"(piece[10].w), "+v"(piece[11].x), "+v"(piece[11].y), "+v"(piece[11].z), "+v"(piece[11].w));
#pragma unroll
        for (int k = 0; k < 12; k++) {
            const int i = threadIdx.x + 256 * k;
            if (i < n4) reinterpret_cast<float4*>(s_sh)[i] = piece[k];
        }
        for (int i = 4 * n4 + threadIdx.x; i < n_stage; i += 256) s_sh[i] = src[i];
        __syncthreads();
    }
    // (outputs only, not `asm volatile`: that counts as a possible store and turns the uniform matrix loads below into vector loads)
    asm("" : "+v"(in_radius), "+v"(ga[0]), "+v"(ga[1]), "+v"(ga[2]), "+v"(ga[3]), "+v"(ga[4]), "+v"(ga[5]), "+v"(ga[6]), "+v"(ga[7]), "+v"(ga[8]), "+v"(ga[9]), "+v"(in_co.x), "+v"(in_co.y), "+v"(in_co.z), "+v"(in_co.w));
    asm("" : "+v"(in_m[0]), "+v"(in_m[1]), "+v"(in_m[2]), "+v"(in_c3[0]), "+v"(in_c3[1]), "+v"(in_c3[2]), "+v"(in_c3[3]), "+v"(in_c3[4]), "+v"(in_c3[5]), "+v"(in_q[0]), "+v"(in_q[1]), "+v"(in_q[2]), "+v"(in_q[3]), "+v"(in_s[0]), "+v"(in_s[1]), "+v"(in_s[2]), "+v"(in_cl), "+v"(in_qraw.x), "+v"(in_qraw.y), "+v"(in_qraw.z), "+v"(in_qraw.w));
    if (idx < a.P) {
    const float* __restrict__ view = a.view;
    const float* __restrict__ proj = a.proj;
    const bool vis = in_radius > 0;

    if (!vis) {
#pragma unroll
        for (int i = 0; i < 10; i++) ga[i] = 0.f;
    }
    // the compositing backward leaves raw sums (raster_render.hip): for the mean, S_x = sum a dx and S_y = sum a dy -- the conic
    // matrix that turns them into dL/d mean (backward.cu:573-579: a (conic (dx, dy))) is the Gaussian's own, so it is applied here,
    // once per Gaussian, instead of there, once per (pixel, splat) pair -- then d(pixel)/d(ndc) with the sign; -1/2 for the conic
    if (vis) {
        const float4 co = in_co;
        const float sx = ga[0], sy = ga[1];
        ga[0] = sx * co.x + sy * co.y;
        ga[1] = sy * co.z + sx * co.y;
    }
    ga[0] *= -0.5f * (float)a.W;
    ga[1] *= -0.5f * (float)a.H;
    ga[2] *= -0.5f;
    ga[3] *= -0.5f;
    ga[4] *= -0.5f;

    a.dmeans2D[3 * idx + 0] = ga[0];
    a.dmeans2D[3 * idx + 1] = ga[1];
    a.dmeans2D[3 * idx + 2] = 0.f;
    // (act_rotations_raw: through the sigmoid, y (1 - y) of the opacity the forward kept in its record -- optim_loss.hip, act_bwd_kernel)
    a.dopacity[idx] = a.rots_raw ? ga[5] * ((1.0f - in_co.w) * in_co.w) : ga[5];
    a.dcolors[3 * idx + 0] = ga[6];
    a.dcolors[3 * idx + 1] = ga[7];
    a.dcolors[3 * idx + 2] = ga[8];

    float dmean[3] = {0.f, 0.f, 0.f};
    float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dscale[3] = {0.f, 0.f, 0.f};
    float drot[4] = {0.f, 0.f, 0.f, 0.f};
    // SH gradient rows: one [P,M,3] tensor, or DC and rest apart (dshr[3*i+c] valid for i >= 1)
    const bool split = a.dsh_rest != nullptr;
    float* dsh = a.dsh ? a.dsh + (size_t)idx * (split ? 1 : a.M) * 3 : nullptr;
    float* dshr = STAGED ? s_sh + threadIdx.x * sh_stride - 3 : (split ? a.dsh_rest + (size_t)idx * (a.M - 1) * 3 - 3 : dsh);

    if (vis) {
        const float mx = in_m[0], my = in_m[1], mz = in_m[2];
        const float* c3 = in_c3;
        // ---- cov2D backward (backward.cu:144-274) ----
        float tx = view[0] * mx + view[4] * my + view[8] * mz + view[12];
        float ty = view[1] * mx + view[5] * my + view[9] * mz + view[13];
        const float tz = view[2] * mx + view[6] * my + view[10] * mz + view[14];
        const float limx = 1.3f * a.tan_fovx, limy = 1.3f * a.tan_fovy;
        const float txtz = tx / tz, tytz = ty / tz;
        tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
        ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
        const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
        const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
        M3 J = {{{a.h_x / tz, 0.0f, -(a.h_x * tx) / (tz * tz)}, {0.0f, a.h_y / tz, -(a.h_y * ty) / (tz * tz)}, {0.f, 0.f, 0.f}}};
        M3 Wm = {{{view[0], view[4], view[8]}, {view[1], view[5], view[9]}, {view[2], view[6], view[10]}}};
        M3 Vrk = {{{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}}};
        M3 T = mul(Wm, J);
        M3 cov2D = mul(mul(transpose(T), transpose(Vrk)), T);
        const float ca = cov2D.m[0][0] + 0.3f, cb = cov2D.m[0][1], cc = cov2D.m[1][1] + 0.3f;
        const float denom = ca * cc - cb * cb;
        float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
        const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        const float dcx = ga[2], dcy = ga[3], dcz = ga[4];  // dL_dconic x, y, w
#define TT(c, r) T.m[c][r]
#define VV(c, r) Vrk.m[c][r]
#define WW(c, r) Wm.m[c][r]
        if (denom2inv != 0) {
            dL_da = denom2inv * (-cc * cc * dcx + 2 * cb * cc * dcy + (denom - ca * cc) * dcz);
            dL_dc = denom2inv * (-ca * ca * dcz + 2 * ca * cb * dcy + (denom - ca * cc) * dcx);
            dL_db = denom2inv * 2 * (cb * cc * dcx - (denom + 2 * cb * cb) * dcy + ca * cb * dcz);
            dcov[0] = (TT(0, 0) * TT(0, 0) * dL_da + TT(0, 0) * TT(1, 0) * dL_db + TT(1, 0) * TT(1, 0) * dL_dc);
            dcov[3] = (TT(0, 1) * TT(0, 1) * dL_da + TT(0, 1) * TT(1, 1) * dL_db + TT(1, 1) * TT(1, 1) * dL_dc);
            dcov[5] = (TT(0, 2) * TT(0, 2) * dL_da + TT(0, 2) * TT(1, 2) * dL_db + TT(1, 2) * TT(1, 2) * dL_dc);
            dcov[1] = 2 * TT(0, 0) * TT(0, 1) * dL_da + (TT(0, 0) * TT(1, 1) + TT(0, 1) * TT(1, 0)) * dL_db + 2 * TT(1, 0) * TT(1, 1) * dL_dc;
            dcov[2] = 2 * TT(0, 0) * TT(0, 2) * dL_da + (TT(0, 0) * TT(1, 2) + TT(0, 2) * TT(1, 0)) * dL_db + 2 * TT(1, 0) * TT(1, 2) * dL_dc;
            dcov[4] = 2 * TT(0, 2) * TT(0, 1) * dL_da + (TT(0, 1) * TT(1, 2) + TT(0, 2) * TT(1, 1)) * dL_db + 2 * TT(1, 1) * TT(1, 2) * dL_dc;
        }
        const float dL_dT00 = 2 * (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_da +
                              (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_db;
        const float dL_dT01 = 2 * (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_da +
                              (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_db;
        const float dL_dT02 = 2 * (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_da +
                              (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_db;
        const float dL_dT10 = 2 * (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_dc +
                              (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_db;
        const float dL_dT11 = 2 * (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_dc +
                              (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_db;
        const float dL_dT12 = 2 * (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_dc +
                              (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_db;
        const float dL_dJ00 = WW(0, 0) * dL_dT00 + WW(0, 1) * dL_dT01 + WW(0, 2) * dL_dT02;
        const float dL_dJ02 = WW(2, 0) * dL_dT00 + WW(2, 1) * dL_dT01 + WW(2, 2) * dL_dT02;
        const float dL_dJ11 = WW(1, 0) * dL_dT10 + WW(1, 1) * dL_dT11 + WW(1, 2) * dL_dT12;
        const float dL_dJ12 = WW(2, 0) * dL_dT10 + WW(2, 1) * dL_dT11 + WW(2, 2) * dL_dT12;
#undef TT
#undef VV
#undef WW
        const float itz = 1.f / tz, itz2 = itz * itz, itz3 = itz2 * itz;
        const float dL_dtx = x_grad_mul * -a.h_x * itz2 * dL_dJ02;
        const float dL_dty = y_grad_mul * -a.h_y * itz2 * dL_dJ12;
        const float dL_dtz = -a.h_x * itz2 * dL_dJ00 - a.h_y * itz2 * dL_dJ11 + (2 * a.h_x * tx) * itz3 * dL_dJ02 +
                             (2 * a.h_y * ty) * itz3 * dL_dJ12;
        dmean[0] = view[0] * dL_dtx + view[1] * dL_dty + view[2] * dL_dtz;
        dmean[1] = view[4] * dL_dtx + view[5] * dL_dty + view[6] * dL_dtz;
        dmean[2] = view[8] * dL_dtx + view[9] * dL_dty + view[10] * dL_dtz;

        // ---- projection backward (backward.cu:372-403) ----
        const float m_hw = proj[3] * mx + proj[7] * my + proj[11] * mz + proj[15];
        const float m_w = 1.0f / (m_hw + 0.0000001f);
        const float mul1 = (proj[0] * mx + proj[4] * my + proj[8] * mz + proj[12]) * m_w * m_w;
        const float mul2 = (proj[1] * mx + proj[5] * my + proj[9] * mz + proj[13]) * m_w * m_w;
        dmean[0] += (proj[0] * m_w - proj[3] * mul1) * ga[0] + (proj[1] * m_w - proj[3] * mul2) * ga[1];
        dmean[1] += (proj[4] * m_w - proj[7] * mul1) * ga[0] + (proj[5] * m_w - proj[7] * mul2) * ga[1];
        dmean[2] += (proj[8] * m_w - proj[11] * mul1) * ga[0] + (proj[9] * m_w - proj[11] * mul2) * ga[1];
        const float mul3 = view[2] * mx + view[6] * my + view[10] * mz + view[14];
        dmean[0] += (view[2] - view[3] * mul3) * ga[9];
        dmean[1] += (view[6] - view[7] * mul3) * ga[9];
        dmean[2] += (view[10] - view[11] * mul3) * ga[9];

        // ---- SH backward (backward.cu:20-139) ----
        if (a.colors_from_sh) {
            const float* __restrict__ cam = a.cam;
            const float dox = mx - cam[0], doy = my - cam[1], doz = mz - cam[2];
            const float len = sqrtf(dox * dox + doy * doy + doz * doz);
            const float x = dox / len, y = doy / len, z = doz / len;
            const float* sh0 = a.shs + (size_t)idx * (a.shs_rest ? 1 : a.M) * 3;
            const float* sh = STAGED ? s_sh + threadIdx.x * sh_stride - 3
                                     : (a.shs_rest ? a.shs_rest + (size_t)idx * (a.M - 1) * 3 - 3 : sh0);
            const uchar4 cl = make_uchar4(in_cl & 255u, (in_cl >> 8) & 255u, (in_cl >> 16) & 255u, in_cl >> 24);
            const float dRGB[3] = {cl.x ? 0.f : ga[6], cl.y ? 0.f : ga[7], cl.z ? 0.f : ga[8]};
            float ddir[3] = {0.f, 0.f, 0.f};
#define SH(i, c) sh[(i) * 3 + (c)]
#define DSH(i, s) { const float s__ = (s); dshr[(i) * 3 + 0] = s__ * dRGB[0]; dshr[(i) * 3 + 1] = s__ * dRGB[1]; dshr[(i) * 3 + 2] = s__ * dRGB[2]; }
            dsh[0] = bSH_C0 * dRGB[0]; dsh[1] = bSH_C0 * dRGB[1]; dsh[2] = bSH_C0 * dRGB[2];
            // the direction gradient READS the coefficients; the coefficient gradients below may overwrite them (staged rows)
            float xx = 0, yy = 0, zz = 0, xy = 0, yz = 0, xz = 0;
            if (a.D > 1) { xx = x * x; yy = y * y; zz = z * z; xy = x * y; yz = y * z; xz = x * z; }
            if (a.D > 0) {
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    float dRdx = -bSH_C1 * SH(3, c), dRdy = -bSH_C1 * SH(1, c), dRdz = bSH_C1 * SH(2, c);
                    if (a.D > 1) {
                        dRdx += bSH_C2[0] * y * SH(4, c) + bSH_C2[2] * 2.f * -x * SH(6, c) + bSH_C2[3] * z * SH(7, c) + bSH_C2[4] * 2.f * x * SH(8, c);
                        dRdy += bSH_C2[0] * x * SH(4, c) + bSH_C2[1] * z * SH(5, c) + bSH_C2[2] * 2.f * -y * SH(6, c) + bSH_C2[4] * 2.f * -y * SH(8, c);
                        dRdz += bSH_C2[1] * y * SH(5, c) + bSH_C2[2] * 2.f * 2.f * z * SH(6, c) + bSH_C2[3] * x * SH(7, c);
                        if (a.D > 2) {
                            dRdx += (bSH_C3[0] * SH(9, c) * 3.f * 2.f * xy + bSH_C3[1] * SH(10, c) * yz + bSH_C3[2] * SH(11, c) * -2.f * xy +
                                     bSH_C3[3] * SH(12, c) * -3.f * 2.f * xz + bSH_C3[4] * SH(13, c) * (-3.f * xx + 4.f * zz - yy) +
                                     bSH_C3[5] * SH(14, c) * 2.f * xz + bSH_C3[6] * SH(15, c) * 3.f * (xx - yy));
                            dRdy += (bSH_C3[0] * SH(9, c) * 3.f * (xx - yy) + bSH_C3[1] * SH(10, c) * xz +
                                     bSH_C3[2] * SH(11, c) * (-3.f * yy + 4.f * zz - xx) + bSH_C3[3] * SH(12, c) * -3.f * 2.f * yz +
                                     bSH_C3[4] * SH(13, c) * -2.f * xy + bSH_C3[5] * SH(14, c) * -2.f * yz +
                                     bSH_C3[6] * SH(15, c) * -3.f * 2.f * xy);
                            dRdz += (bSH_C3[1] * SH(10, c) * xy + bSH_C3[2] * SH(11, c) * 4.f * 2.f * yz +
                                     bSH_C3[3] * SH(12, c) * 3.f * (2.f * zz - xx - yy) + bSH_C3[4] * SH(13, c) * 4.f * 2.f * xz +
                                     bSH_C3[5] * SH(14, c) * (xx - yy));
                        }
                    }
                    ddir[0] += dRdx * dRGB[c];
                    ddir[1] += dRdy * dRGB[c];
                    ddir[2] += dRdz * dRGB[c];
                }
            }
            if (a.D > 0) {
                DSH(1, -bSH_C1 * y);
                DSH(2, bSH_C1 * z);
                DSH(3, -bSH_C1 * x);
                if (a.D > 1) {
                    DSH(4, bSH_C2[0] * xy);
                    DSH(5, bSH_C2[1] * yz);
                    DSH(6, bSH_C2[2] * (2.f * zz - xx - yy));
                    DSH(7, bSH_C2[3] * xz);
                    DSH(8, bSH_C2[4] * (xx - yy));
                    if (a.D > 2) {
                        DSH(9, bSH_C3[0] * y * (3.f * xx - yy));
                        DSH(10, bSH_C3[1] * xy * z);
                        DSH(11, bSH_C3[2] * y * (4.f * zz - xx - yy));
                        DSH(12, bSH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy));
                        DSH(13, bSH_C3[4] * x * (4.f * zz - xx - yy));
                        DSH(14, bSH_C3[5] * z * (xx - yy));
                        DSH(15, bSH_C3[6] * x * (xx - 3.f * yy));
                    }
                }
            }
            // coefficients above the active degree receive zero gradient
            const int used = (a.D + 1) * (a.D + 1);
            for (int i = used; i < a.M; i++) { dshr[i * 3 + 0] = 0.f; dshr[i * 3 + 1] = 0.f; dshr[i * 3 + 2] = 0.f; }
#undef SH
#undef DSH
            // through the normalisation of the view direction (auxiliary.h:107-117)
            const float sum2 = dox * dox + doy * doy + doz * doz;
            const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
            dmean[0] += ((+sum2 - dox * dox) * ddir[0] - doy * dox * ddir[1] - doz * dox * ddir[2]) * invsum32;
            dmean[1] += (-dox * doy * ddir[0] + (sum2 - doy * doy) * ddir[1] - doz * doy * ddir[2]) * invsum32;
            dmean[2] += (-dox * doz * ddir[0] - doy * doz * ddir[1] + (sum2 - doz * doz) * ddir[2]) * invsum32;
        }

        // ---- cov3D backward (backward.cu:278-341), no quaternion-normalisation Jacobian ----
        if (a.scales) {
            const float r = in_q[0], x = in_q[1], y = in_q[2], z = in_q[3];
            M3 Rm = {{{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
                      {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
                      {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}}};
            const float s[3] = {a.scale_modifier * in_s[0], a.scale_modifier * in_s[1], a.scale_modifier * in_s[2]};
            M3 S = {{{s[0], 0.f, 0.f}, {0.f, s[1], 0.f}, {0.f, 0.f, s[2]}}};
            M3 Mm = mul(S, Rm);
            M3 dSig = {{{dcov[0], 0.5f * dcov[1], 0.5f * dcov[2]}, {0.5f * dcov[1], dcov[3], 0.5f * dcov[4]},
                        {0.5f * dcov[2], 0.5f * dcov[4], dcov[5]}}};
            M3 M2;
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int rr = 0; rr < 3; rr++) M2.m[c][rr] = Mm.m[c][rr] * 2.0f;
            M3 dM = mul(M2, dSig);
            M3 Rt = transpose(Rm);
            M3 dMt = transpose(dM);
#pragma unroll
            for (int c = 0; c < 3; c++)
                dscale[c] = Rt.m[c][0] * dMt.m[c][0] + Rt.m[c][1] * dMt.m[c][1] + Rt.m[c][2] * dMt.m[c][2];
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int rr = 0; rr < 3; rr++) dMt.m[c][rr] *= s[c];
#define DM(c, rr) dMt.m[c][rr]
            drot[0] = 2 * z * (DM(0, 1) - DM(1, 0)) + 2 * y * (DM(2, 0) - DM(0, 2)) + 2 * x * (DM(1, 2) - DM(2, 1));
            drot[1] = 2 * y * (DM(1, 0) + DM(0, 1)) + 2 * z * (DM(2, 0) + DM(0, 2)) + 2 * r * (DM(1, 2) - DM(2, 1)) - 4 * x * (DM(2, 2) + DM(1, 1));
            drot[2] = 2 * x * (DM(1, 0) + DM(0, 1)) + 2 * r * (DM(2, 0) - DM(0, 2)) + 2 * z * (DM(1, 2) + DM(2, 1)) - 4 * y * (DM(2, 2) + DM(0, 0));
            drot[3] = 2 * r * (DM(0, 1) - DM(1, 0)) + 2 * x * (DM(2, 0) + DM(0, 2)) + 2 * y * (DM(1, 2) + DM(2, 1)) - 4 * z * (DM(1, 1) + DM(0, 0));
#undef DM
        }
    }
    if (dsh && !(vis && a.colors_from_sh)) {
        dsh[0] = dsh[1] = dsh[2] = 0.f;
        for (int i = 3; i < a.M * 3; i++) dshr[i] = 0.f;
    }

#pragma unroll
    for (int i = 0; i < 3; i++) a.dmeans3D[3 * idx + i] = dmean[i];
#pragma unroll
    for (int i = 0; i < 6; i++) a.dcov3D[6 * idx + i] = dcov[i];
    if (a.rots_raw) {
        // through exp (d = g * exp(raw) = g * scale) and through q / max(|q|, eps) (zero through the clamp, as ATen does): the
        // arithmetic of act_bwd_kernel (optim_loss.hip), operation for operation
#pragma unroll
        for (int i = 0; i < 3; i++) dscale[i] = dscale[i] * in_s[i];
        const float4 q = in_qraw;
        const float nrm = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
        const float n = fmaxf(nrm, 1e-12f);
        const float4 u = make_float4(q.x / n, q.y / n, q.z / n, q.w / n);
        const float dot = (nrm >= 1e-12f) ? (u.x * drot[0] + u.y * drot[1] + u.z * drot[2] + u.w * drot[3]) : 0.f;
        const float g4[4] = {drot[0], drot[1], drot[2], drot[3]};
        drot[0] = (g4[0] - u.x * dot) / n;
        drot[1] = (g4[1] - u.y * dot) / n;
        drot[2] = (g4[2] - u.z * dot) / n;
        drot[3] = (g4[3] - u.w * dot) / n;
    }
    if (a.dscales) {
#pragma unroll
        for (int i = 0; i < 3; i++) a.dscales[3 * idx + i] = dscale[i];
    }
    if (a.drot) {
#pragma unroll
        for (int i = 0; i < 4; i++) a.drot[4 * idx + i] = drot[i];
    }
    if (a.dscales2) {
#pragma unroll
        for (int i = 0; i < 3; i++) a.dscales2[3 * idx + i] = dscale[i];
    }
    if (a.drot2) {
#pragma unroll
        for (int i = 0; i < 4; i++) a.drot2[4 * idx + i] = drot[i];
    }
    }   // idx < P
    if (STAGED) {
        __syncthreads();
        float* __restrict__ dst = a.dsh_rest + (size_t)block0 * sh_stride;
        const int n4 = n_stage >> 2;
        for (int i = threadIdx.x; i < n4; i += 256) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(s_sh)[i];
        for (int i = 4 * n4 + threadIdx.x; i < n_stage; i += 256) dst[i] = s_sh[i];
    }
}

}  // namespace

int mom_launch_preprocess_bwd(const MomRasterArgs* a, const int* radii, const GeomView& g, const MomRasterGrads* gr, hipStream_t s)
{
    BwdArgs b;
    b.P = a->P; b.D = a->D; b.M = a->M; b.W = a->W; b.H = a->H;
    b.means3D = a->means3D; b.shs = a->shs; b.shs_rest = a->shs_rest; b.scales = a->scales; b.rotations = a->rotations;
    b.cov3D = a->cov3D_precomp ? a->cov3D_precomp : g.cov3D;
    b.view = a->viewmatrix; b.proj = a->projmatrix; b.cam = a->campos;
    b.radii = radii; b.clamped = g.clamped; b.gacc = g.gacc; b.rec = g.rec;
    b.scale_modifier = a->scale_modifier; b.tan_fovx = a->tan_fovx; b.tan_fovy = a->tan_fovy;
    b.h_y = a->H / (2.0f * a->tan_fovy);
    b.h_x = a->W / (2.0f * a->tan_fovx);
    b.colors_from_sh = (a->colors_precomp == nullptr && a->shs != nullptr && gr->dL_dsh != nullptr) ? 1 : 0;
    b.dmeans2D = gr->dL_dmeans2D; b.dcolors = gr->dL_dcolors; b.dopacity = gr->dL_dopacity; b.dmeans3D = gr->dL_dmeans3D;
    b.dcov3D = gr->dL_dcov3D; b.dsh = gr->dL_dsh; b.dsh_rest = a->shs_rest ? gr->dL_dsh_rest : nullptr; b.dscales = a->scales ? gr->dL_dscales : nullptr;
    b.drot = a->scales ? gr->dL_drotations : nullptr;
    b.dscales2 = a->scales ? gr->dL_dscales_copy : nullptr;
    b.drot2 = a->scales ? gr->dL_drotations_copy : nullptr;
    b.rots_raw = gr->act_rotations_raw;                      // (raster_api.hip, check_grads: only with scales and rotations present)
    MomProfScope ps(MOM_P_PRE_BWD, s);
    const int sh_stride = (a->M - 1) * 3;
    const bool staged = b.colors_from_sh && b.shs_rest && b.dsh_rest && (sh_stride & 1) && sh_stride <= 45 &&
                        ((uintptr_t)b.shs_rest & 15) == 0 && ((uintptr_t)b.dsh_rest & 15) == 0;
    if (staged)
        hipLaunchKernelGGL(preprocess_bwd_kernel<true>, dim3((a->P + 255) / 256), dim3(256), (size_t)256 * sh_stride * 4, s, b);
    else
        hipLaunchKernelGGL(preprocess_bwd_kernel<false>, dim3((a->P + 255) / 256), dim3(256), 0, s, b);
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}
