/*
 * raster_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle, not the product).
 *
 * Plain-C restatement of the reference's differentiable tile rasterizer
 * (submodules/depth-diff-gaussian-rasterization) and of simple-knn's
 * distCUDA2.  Every function cites the reference file:line it follows.  The
 * CUDA thread grid is replaced by loops; nothing else is re-designed, and the
 * reference's quirks are kept on purpose (fp64 ndc2Pix, truncating tile-rect
 * casts, un-normalised quaternion, no clamp derivative for the 0.99 alpha cap,
 * {0,0,0}-seeded min/max reduction in simple-knn ...).
 *
 * Parity status: the reference holds NO tests or golden vectors for this path
 *  and its CUDA sources cannot be built here (no nvcc, no
 * CUDA GPU) => "parity unpinned by the reference".  The oracle is pinned
 * instead by tests/test_oracle_*.py: hand-derived known answers, structural
 * invariants, finite differences of the fp64 build of this same file
 * (tests/test_oracle_raster.py), an independent pure-torch statement of the
 * forward whose torch.autograd gradients equal this file's backward to 1e-7
 * (oracle/torch_raster.py, tests/test_oracle_torch.py), and the reference's
 * own importable Python (eval_sh, build_covariance_from_scaling_rotation).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's shared object.
 *
 * Build: see oracle/Makefile.  -DORACLE_FP64 switches `real` to double (used
 * for finite-difference checks of the backward pass).  Floating-point
 * contraction must stay off (-ffp-contract=off) so that integer outputs
 * (radii, tile rects, sort keys) are reproducible bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifdef ORACLE_FP64
typedef double real;
#define R_SQRT sqrt
#define R_EXP exp
#define R_CEIL ceil
#define R_FABS fabs
#else
typedef float real;
#define R_SQRT sqrtf
#define R_EXP expf
#define R_CEIL ceilf
#define R_FABS fabsf
#endif

/* config.h:14-16, auxiliary.h:18 */
#define NUM_CHANNELS 3
#define BLOCK_X 16
#define BLOCK_Y 16
#define BLOCK_SIZE (BLOCK_X * BLOCK_Y)

/* auxiliary.h:22-39 */
static const real SH_C0 = (real)0.28209479177387814f;
static const real SH_C1 = (real)0.4886025119029199f;
static const real SH_C2[5] = {
    (real)1.0925484305920792f, (real)-1.0925484305920792f, (real)0.31539156525252005f,
    (real)-1.0925484305920792f, (real)0.5462742152960396f};
static const real SH_C3[7] = {
    (real)-0.5900435899266435f, (real)2.890611442640554f, (real)-0.4570457994644658f,
    (real)0.3731763325901154f, (real)-0.4570457994644658f, (real)1.445305721320277f,
    (real)-0.5900435899266435f};

static inline real rmin(real a, real b) { return a < b ? a : b; }
static inline real rmax(real a, real b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* ---- minimal column-major 3x3 helper with glm's arithmetic order ----------
 * glm (vendored third_party/glm, glm/detail/type_mat3x3.inl:486-519):
 * m[c][r]; (A*B)[c][r] = A[0][r]*B[c][0] + A[1][r]*B[c][1] + A[2][r]*B[c][2],
 * summed left to right. */
typedef struct { real m[3][3]; } mat3; /* m[col][row] */

static inline mat3 mat3_cols(real a, real b, real c, real d, real e, real f, real g, real h, real i)
{
    mat3 r;
    r.m[0][0] = a; r.m[0][1] = b; r.m[0][2] = c;
    r.m[1][0] = d; r.m[1][1] = e; r.m[1][2] = f;
    r.m[2][0] = g; r.m[2][1] = h; r.m[2][2] = i;
    return r;
}
static inline mat3 mat3_mul(const mat3* A, const mat3* B)
{
    mat3 R;
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++)
            R.m[c][r] = A->m[0][r] * B->m[c][0] + A->m[1][r] * B->m[c][1] + A->m[2][r] * B->m[c][2];
    return R;
}
static inline mat3 mat3_T(const mat3* A)
{
    mat3 R;
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++)
            R.m[c][r] = A->m[r][c];
    return R;
}

/* auxiliary.h:41-44 -- evaluated in double, returned as float */
static inline real ndc2Pix(real v, int S)
{
    return (real)((((double)v + 1.0) * (double)S - 1.0) * 0.5);
}

/* auxiliary.h:46-56 -- C truncation casts, then clamp to the tile grid */
static inline void getRect(real px, real py, int max_radius, int* rmin_x, int* rmin_y, int* rmax_x,
                           int* rmax_y, int grid_x, int grid_y)
{
    *rmin_x = imin(grid_x, imax(0, (int)((px - max_radius) / BLOCK_X)));
    *rmin_y = imin(grid_y, imax(0, (int)((py - max_radius) / BLOCK_Y)));
    *rmax_x = imin(grid_x, imax(0, (int)((px + max_radius + BLOCK_X - 1) / BLOCK_X)));
    *rmax_y = imin(grid_y, imax(0, (int)((py + max_radius + BLOCK_Y - 1) / BLOCK_Y)));
}

/* auxiliary.h:58-77 */
static inline void transformPoint4x3(const real* p, const real* m, real* o)
{
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static inline void transformPoint4x4(const real* p, const real* m, real* o)
{
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}
/* auxiliary.h:89-97 */
static inline void transformVec4x3Transpose(const real* p, const real* m, real* o)
{
    o[0] = m[0] * p[0] + m[1] * p[1] + m[2] * p[2];
    o[1] = m[4] * p[0] + m[5] * p[1] + m[6] * p[2];
    o[2] = m[8] * p[0] + m[9] * p[1] + m[10] * p[2];
}
/* auxiliary.h:107-117 */
static inline void dnormvdv3(const real* v, const real* dv, real* o)
{
    real sum2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    real invsum32 = (real)1.0 / R_SQRT(sum2 * sum2 * sum2);
    o[0] = ((+sum2 - v[0] * v[0]) * dv[0] - v[1] * v[0] * dv[1] - v[2] * v[0] * dv[2]) * invsum32;
    o[1] = (-v[0] * v[1] * dv[0] + (sum2 - v[1] * v[1]) * dv[1] - v[2] * v[1] * dv[2]) * invsum32;
    o[2] = (-v[0] * v[2] * dv[0] - v[1] * v[2] * dv[1] + (sum2 - v[2] * v[2]) * dv[2]) * invsum32;
}

/* auxiliary.h:139-164 (prefiltered trap omitted: it aborts the CUDA kernel) */
static inline int in_frustum(int idx, const real* orig_points, const real* viewmatrix, real* p_view)
{
    const real* p_orig = orig_points + 3 * idx;
    transformPoint4x3(p_orig, viewmatrix, p_view);
    if (p_view[2] <= (real)0.2f)
        return 0;
    return 1;
}

/* forward.cu:20-71 */
static void computeColorFromSH_fwd(int idx, int deg, int max_coeffs, const real* means, const real* campos,
                                   const real* shs, uint8_t* clamped, real* result)
{
    const real* pos = means + 3 * idx;
    real dir[3] = {pos[0] - campos[0], pos[1] - campos[1], pos[2] - campos[2]};
    real len = R_SQRT(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]); /* glm::length */
    dir[0] = dir[0] / len; dir[1] = dir[1] / len; dir[2] = dir[2] / len;

    const real* sh = shs + (size_t)idx * max_coeffs * 3; /* glm::vec3 sh[max_coeffs] */
#define SH(i, c) sh[(i) * 3 + (c)]
    for (int c = 0; c < 3; c++) {
        real res = SH_C0 * SH(0, c);
        if (deg > 0) {
            real x = dir[0], y = dir[1], z = dir[2];
            res = res - SH_C1 * y * SH(1, c) + SH_C1 * z * SH(2, c) - SH_C1 * x * SH(3, c);
            if (deg > 1) {
                real xx = x * x, yy = y * y, zz = z * z;
                real xy = x * y, yz = y * z, xz = x * z;
                res = res + SH_C2[0] * xy * SH(4, c) + SH_C2[1] * yz * SH(5, c) +
                      SH_C2[2] * ((real)2.0 * zz - xx - yy) * SH(6, c) + SH_C2[3] * xz * SH(7, c) +
                      SH_C2[4] * (xx - yy) * SH(8, c);
                if (deg > 2) {
                    res = res + SH_C3[0] * y * ((real)3.0 * xx - yy) * SH(9, c) +
                          SH_C3[1] * xy * z * SH(10, c) +
                          SH_C3[2] * y * ((real)4.0 * zz - xx - yy) * SH(11, c) +
                          SH_C3[3] * z * ((real)2.0 * zz - (real)3.0 * xx - (real)3.0 * yy) * SH(12, c) +
                          SH_C3[4] * x * ((real)4.0 * zz - xx - yy) * SH(13, c) +
                          SH_C3[5] * z * (xx - yy) * SH(14, c) +
                          SH_C3[6] * x * (xx - (real)3.0 * yy) * SH(15, c);
                }
            }
        }
        res += (real)0.5;
        clamped[3 * idx + c] = (res < 0);
        result[c] = rmax(res, (real)0.0);
    }
#undef SH
}

/* forward.cu:74-113 (shared with backward.cu:166-199) */
static void cov2D_TJ(const real* mean, real focal_x, real focal_y, real tan_fovx, real tan_fovy,
                     const real* cov3D, const real* viewmatrix, mat3* T_out, mat3* Vrk_out, mat3* cov_out,
                     real* t_out, real* txtz_out, real* tytz_out)
{
    real t[3];
    transformPoint4x3(mean, viewmatrix, t);
    const real limx = (real)1.3f * tan_fovx;
    const real limy = (real)1.3f * tan_fovy;
    const real txtz = t[0] / t[2];
    const real tytz = t[1] / t[2];
    t[0] = rmin(limx, rmax(-limx, txtz)) * t[2];
    t[1] = rmin(limy, rmax(-limy, tytz)) * t[2];

    mat3 J = mat3_cols(focal_x / t[2], (real)0.0, -(focal_x * t[0]) / (t[2] * t[2]),
                       (real)0.0, focal_y / t[2], -(focal_y * t[1]) / (t[2] * t[2]),
                       0, 0, 0);
    mat3 W = mat3_cols(viewmatrix[0], viewmatrix[4], viewmatrix[8],
                       viewmatrix[1], viewmatrix[5], viewmatrix[9],
                       viewmatrix[2], viewmatrix[6], viewmatrix[10]);
    mat3 T = mat3_mul(&W, &J);
    mat3 Vrk = mat3_cols(cov3D[0], cov3D[1], cov3D[2],
                         cov3D[1], cov3D[3], cov3D[4],
                         cov3D[2], cov3D[4], cov3D[5]);
    mat3 Tt = mat3_T(&T), Vt = mat3_T(&Vrk);
    mat3 tmp = mat3_mul(&Tt, &Vt);
    mat3 cov = mat3_mul(&tmp, &T);
    *T_out = T; *Vrk_out = Vrk; *cov_out = cov;
    t_out[0] = t[0]; t_out[1] = t[1]; t_out[2] = t[2];
    *txtz_out = txtz; *tytz_out = tytz;
}

/* forward.cu:118-152 -- quaternion (r,x,y,z) used AS GIVEN (normalisation commented out) */
static void computeCov3D_fwd(const real* scale, real mod, const real* rot, real* cov3D)
{
    mat3 S = mat3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
    S.m[0][0] = mod * scale[0];
    S.m[1][1] = mod * scale[1];
    S.m[2][2] = mod * scale[2];
    real r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    mat3 R = mat3_cols(
        (real)1.0 - (real)2.0 * (y * y + z * z), (real)2.0 * (x * y - r * z), (real)2.0 * (x * z + r * y),
        (real)2.0 * (x * y + r * z), (real)1.0 - (real)2.0 * (x * x + z * z), (real)2.0 * (y * z - r * x),
        (real)2.0 * (x * z - r * y), (real)2.0 * (y * z + r * x), (real)1.0 - (real)2.0 * (x * x + y * y));
    mat3 M = mat3_mul(&S, &R);
    mat3 Mt = mat3_T(&M);
    mat3 Sigma = mat3_mul(&Mt, &M);
    cov3D[0] = Sigma.m[0][0];
    cov3D[1] = Sigma.m[0][1];
    cov3D[2] = Sigma.m[0][2];
    cov3D[3] = Sigma.m[1][1];
    cov3D[4] = Sigma.m[1][2];
    cov3D[5] = Sigma.m[2][2];
}

/* rasterizer_impl.cu:35-50 */
static uint32_t getHigherMsb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4;
    uint32_t step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb)
            msb += step;
        else
            msb -= step;
    }
    if (n >> msb)
        msb++;
    return msb;
}
uint32_t oracle_get_higher_msb(uint32_t n) { return getHigherMsb(n); }

void oracle_set_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : 1);
#else
    (void)n;
#endif
}
int oracle_real_size(void) { return (int)sizeof(real); }

/* ---------------------------------------------------------------------------
 * Forward, part 1: preprocessCUDA (forward.cu:156-256) for every Gaussian,
 * then the inclusive scan of tiles_touched (rasterizer_impl.cu:278).
 * Returns num_rendered (rasterizer_impl.cu:281-282).
 * Any of shs / colors_precomp / scales+rotations / cov3D_precomp may be NULL
 * exactly as an empty tensor stands for "absent" in the reference.
 * ------------------------------------------------------------------------- */
int oracle_preprocess(int P, int D, int M, const real* orig_points, const real* scales, real scale_modifier,
                      const real* rotations, const real* opacities, const real* shs, uint8_t* clamped,
                      const real* cov3D_precomp, const real* colors_precomp, const real* viewmatrix,
                      const real* projmatrix, const real* cam_pos, int W, int H, real tan_fovx,
                      real tan_fovy, int* radii, real* points_xy_image, real* depths, real* cov3Ds,
                      real* rgb, real* conic_opacity, uint32_t* tiles_touched, uint32_t* point_offsets)
{
    /* rasterizer_impl.cu:223-224,235 */
    const real focal_y = H / ((real)2.0 * tan_fovy);
    const real focal_x = W / ((real)2.0 * tan_fovx);
    const int grid_x = (W + BLOCK_X - 1) / BLOCK_X, grid_y = (H + BLOCK_Y - 1) / BLOCK_Y;

#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        radii[idx] = 0;
        tiles_touched[idx] = 0;

        real p_view[3];
        if (!in_frustum(idx, orig_points, viewmatrix, p_view))
            continue;

        const real* p_orig = orig_points + 3 * idx;
        real p_hom[4];
        transformPoint4x4(p_orig, projmatrix, p_hom);
        real p_w = (real)1.0 / (p_hom[3] + (real)0.0000001f);
        real p_proj[3] = {p_hom[0] * p_w, p_hom[1] * p_w, p_hom[2] * p_w};

        const real* cov3D;
        if (cov3D_precomp != NULL) {
            cov3D = cov3D_precomp + idx * 6;
        } else {
            computeCov3D_fwd(scales + 3 * idx, scale_modifier, rotations + 4 * idx, cov3Ds + idx * 6);
            cov3D = cov3Ds + idx * 6;
        }

        mat3 T, Vrk, cov2;
        real t[3], txtz, tytz;
        cov2D_TJ(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, viewmatrix, &T, &Vrk, &cov2, t, &txtz, &tytz);
        /* forward.cu:110-112 */
        cov2.m[0][0] += (real)0.3f;
        cov2.m[1][1] += (real)0.3f;
        real cov_x = cov2.m[0][0], cov_y = cov2.m[0][1], cov_z = cov2.m[1][1];

        /* forward.cu:219-223 */
        real det = (cov_x * cov_z - cov_y * cov_y);
        if (det == (real)0.0)
            continue;
        real det_inv = (real)1.0 / det;
        real conic[3] = {cov_z * det_inv, -cov_y * det_inv, cov_x * det_inv};

        /* forward.cu:229-237 */
        real mid = (real)0.5 * (cov_x + cov_z);
        real lambda1 = mid + R_SQRT(rmax((real)0.1f, mid * mid - det));
        real lambda2 = mid - R_SQRT(rmax((real)0.1f, mid * mid - det));
        real my_radius = R_CEIL((real)3.0 * R_SQRT(rmax(lambda1, lambda2)));
        real point_image[2] = {ndc2Pix(p_proj[0], W), ndc2Pix(p_proj[1], H)};
        int rminx, rminy, rmaxx, rmaxy;
        getRect(point_image[0], point_image[1], (int)my_radius, &rminx, &rminy, &rmaxx, &rmaxy, grid_x, grid_y);
        if ((uint32_t)(rmaxx - rminx) * (uint32_t)(rmaxy - rminy) == 0)
            continue;

        /* forward.cu:241-247 */
        if (colors_precomp == NULL) {
            real result[3];
            computeColorFromSH_fwd(idx, D, M, orig_points, cam_pos, shs, clamped, result);
            rgb[idx * NUM_CHANNELS + 0] = result[0];
            rgb[idx * NUM_CHANNELS + 1] = result[1];
            rgb[idx * NUM_CHANNELS + 2] = result[2];
        }

        /* forward.cu:250-255 */
        depths[idx] = p_view[2];
        radii[idx] = (int)my_radius;
        points_xy_image[2 * idx + 0] = point_image[0];
        points_xy_image[2 * idx + 1] = point_image[1];
        conic_opacity[4 * idx + 0] = conic[0];
        conic_opacity[4 * idx + 1] = conic[1];
        conic_opacity[4 * idx + 2] = conic[2];
        conic_opacity[4 * idx + 3] = opacities[idx];
        tiles_touched[idx] = (uint32_t)(rmaxy - rminy) * (uint32_t)(rmaxx - rminx);
    }

    /* rasterizer_impl.cu:278 InclusiveSum */
    uint32_t acc = 0;
    for (int i = 0; i < P; i++) {
        acc += tiles_touched[i];
        point_offsets[i] = acc;
    }
    return (int)acc;
}

/* ---------------------------------------------------------------------------
 * Forward, part 2: duplicateWithKeys (rasterizer_impl.cu:70-111), the stable
 * LSD radix sort over bits [0, 32+getHigherMsb(tiles)) (:301-309),
 * identifyTileRanges (:116-138, ranges zeroed first :311) and renderCUDA
 * (forward.cu:261-379).
 * depth_bits: the fp32 bit patterns of depths (in the fp64 build the caller
 * passes float32-rounded depths' bits so that keys stay 64-bit).
 * ------------------------------------------------------------------------- */
static void radix_sort_pairs(uint64_t* keys_in, uint64_t* keys_out, uint32_t* vals_in, uint32_t* vals_out, int n,
                             int end_bit)
{
    /* stable LSD, 8-bit digits; result == cub::DeviceRadixSort::SortPairs(begin_bit=0,end_bit) */
    uint64_t* ka = keys_in; uint64_t* kb = keys_out;
    uint32_t* va = vals_in; uint32_t* vb = vals_out;
    int passes = 0;
    for (int shift = 0; shift < end_bit; shift += 8, passes++) {
        int bits = end_bit - shift < 8 ? end_bit - shift : 8;
        uint32_t mask = (1u << bits) - 1u;
        size_t count[257];
        memset(count, 0, sizeof(count));
        for (int i = 0; i < n; i++)
            count[((ka[i] >> shift) & mask) + 1]++;
        for (int d = 0; d < 256; d++)
            count[d + 1] += count[d];
        for (int i = 0; i < n; i++) {
            size_t pos = count[(ka[i] >> shift) & mask]++;
            kb[pos] = ka[i];
            vb[pos] = va[i];
        }
        uint64_t* tk = ka; ka = kb; kb = tk;
        uint32_t* tv = va; va = vb; vb = tv;
    }
    if (ka != keys_out) {
        memcpy(keys_out, ka, sizeof(uint64_t) * (size_t)n);
        memcpy(vals_out, va, sizeof(uint32_t) * (size_t)n);
    }
}

void oracle_bin(int P, int W, int H, int num_rendered, const real* points_xy, const uint32_t* depth_bits,
                const uint32_t* point_offsets, const int* radii, uint64_t* keys_unsorted,
                uint32_t* values_unsorted, uint64_t* keys_sorted, uint32_t* point_list, uint32_t* ranges /* [tiles][2] */)
{
    const int grid_x = (W + BLOCK_X - 1) / BLOCK_X, grid_y = (H + BLOCK_Y - 1) / BLOCK_Y;
    /* duplicateWithKeys */
    for (int idx = 0; idx < P; idx++) {
        if (radii[idx] > 0) {
            uint32_t off = (idx == 0) ? 0 : point_offsets[idx - 1];
            int rminx, rminy, rmaxx, rmaxy;
            getRect(points_xy[2 * idx], points_xy[2 * idx + 1], radii[idx], &rminx, &rminy, &rmaxx, &rmaxy, grid_x, grid_y);
            for (int y = rminy; y < rmaxy; y++)
                for (int x = rminx; x < rmaxx; x++) {
                    uint64_t key = (uint64_t)(y * grid_x + x);
                    key <<= 32;
                    key |= depth_bits[idx];
                    keys_unsorted[off] = key;
                    values_unsorted[off] = (uint32_t)idx;
                    off++;
                }
        }
    }
    int bit = (int)getHigherMsb((uint32_t)(grid_x * grid_y));
    uint64_t* tmpk = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(num_rendered > 0 ? num_rendered : 1));
    uint32_t* tmpv = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(num_rendered > 0 ? num_rendered : 1));
    memcpy(tmpk, keys_unsorted, sizeof(uint64_t) * (size_t)num_rendered);
    memcpy(tmpv, values_unsorted, sizeof(uint32_t) * (size_t)num_rendered);
    radix_sort_pairs(tmpk, keys_sorted, tmpv, point_list, num_rendered, 32 + bit);
    free(tmpk); free(tmpv);

    memset(ranges, 0, sizeof(uint32_t) * 2 * (size_t)(grid_x * grid_y));
    /* identifyTileRanges */
    for (int idx = 0; idx < num_rendered; idx++) {
        uint64_t key = keys_sorted[idx];
        uint32_t currtile = (uint32_t)(key >> 32);
        if (idx == 0)
            ranges[2 * currtile + 0] = 0;
        else {
            uint32_t prevtile = (uint32_t)(keys_sorted[idx - 1] >> 32);
            if (currtile != prevtile) {
                ranges[2 * prevtile + 1] = (uint32_t)idx;
                ranges[2 * currtile + 0] = (uint32_t)idx;
            }
        }
        if (idx == num_rendered - 1)
            ranges[2 * currtile + 1] = (uint32_t)num_rendered;
    }
}

/* renderCUDA forward (forward.cu:261-379).  The per-block batching through
 * shared memory and the block-wide early exit do not change any pixel's
 * result (a pixel only ever stops on its own `done`), so each pixel walks its
 * tile's range directly. */
void oracle_render_forward(int W, int H, const uint32_t* ranges, const uint32_t* point_list,
                           const real* points_xy_image, const real* features, const real* depths,
                           const real* conic_opacity, const real* bg_color, real* final_T, uint32_t* n_contrib,
                           real* out_color, real* out_depth, uint64_t* pairs_evaluated /* optional */,
                           uint32_t* tile_walked /* optional [tiles]: entries the tile's BLOCK walks before it exits */)
{
    const int grid_x = (W + BLOCK_X - 1) / BLOCK_X, grid_y = (H + BLOCK_Y - 1) / BLOCK_Y;
    uint64_t total_pairs = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : total_pairs)
    for (int tile = 0; tile < grid_x * grid_y; tile++) {
        const int tx = tile % grid_x, ty = tile / grid_x;
        const uint32_t rx = ranges[2 * tile], ry = ranges[2 * tile + 1];
        /* the block's own walk (forward.cu:305-312): the list is fetched in rounds of BLOCK_SIZE entries and a round is entered
         * unless every thread of the block is done (threads outside the image are done from the start).  So the block walks the
         * whole list if any pixel never saturates, else up to the end of the round in which the last pixel saturated. */
        uint32_t last_done_at = 0;
        int any_not_done = 0;
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
                if (!(px < W && py < H))
                    continue;
                const int pix_id = W * py + px;
                const real pixf[2] = {(real)px, (real)py};
                real T = (real)1.0;
                uint32_t contributor = 0, last_contributor = 0, done_at = 0;
                real C[NUM_CHANNELS] = {0};
                real Dp = 0;
                for (uint32_t k = rx; k < ry; k++) {
                    contributor++;
                    total_pairs++;
                    const int id = (int)point_list[k];
                    real dx = points_xy_image[2 * id] - pixf[0];
                    real dy = points_xy_image[2 * id + 1] - pixf[1];
                    const real* con_o = conic_opacity + 4 * id;
                    real power = (real)-0.5 * (con_o[0] * dx * dx + con_o[2] * dy * dy) - con_o[1] * dx * dy;
                    if (power > (real)0.0)
                        continue;
                    real alpha = rmin((real)0.99f, con_o[3] * R_EXP(power));
                    if (alpha < (real)1.0 / (real)255.0)
                        continue;
                    real test_T = T * (1 - alpha);
                    if (test_T < (real)0.0001f) {
                        done_at = contributor;
                        break; /* done = true */
                    }
                    for (int ch = 0; ch < NUM_CHANNELS; ch++)
                        C[ch] += features[id * NUM_CHANNELS + ch] * alpha * T;
                    Dp += depths[id] * alpha * T;
                    T = test_T;
                    last_contributor = contributor;
                }
                if (done_at == 0)
                    any_not_done = 1;
                else if (done_at > last_done_at)
                    last_done_at = done_at;
                final_T[pix_id] = T;
                n_contrib[pix_id] = last_contributor;
                for (int ch = 0; ch < NUM_CHANNELS; ch++)
                    out_color[ch * H * W + pix_id] = C[ch] + T * bg_color[ch];
                out_depth[pix_id] = Dp;
            }
        if (tile_walked) {
            const uint32_t len = ry - rx;
            const uint32_t upto = (last_done_at + BLOCK_SIZE - 1) / BLOCK_SIZE * BLOCK_SIZE;
            tile_walked[tile] = (any_not_done || upto > len) ? len : upto;
        }
    }
    if (pairs_evaluated)
        *pairs_evaluated = total_pairs;
}

/* ---------------------------------------------------------------------------
 * Backward: renderCUDA (backward.cu:415-590).  atomicAdd becomes a plain +=
 * in tile / pixel order (the reference's order is unspecified); under OpenMP
 * the adds are `omp atomic`.
 * dL_dmean2D is [P,3] (x,y used), dL_dconic2D is [P,4] (x,y,w used).
 * ------------------------------------------------------------------------- */
void oracle_render_backward(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const real* bg_color,
                            const real* points_xy_image, const real* conic_opacity, const real* colors,
                            const real* depths, const real* final_Ts, const uint32_t* n_contrib,
                            const real* dL_dpixels, const real* dL_dpixel_depths, real* dL_dmean2D,
                            real* dL_dconic2D, real* dL_dopacity, real* dL_dcolors, real* dL_ddepths)
{
    const int grid_x = (W + BLOCK_X - 1) / BLOCK_X, grid_y = (H + BLOCK_Y - 1) / BLOCK_Y;
    const int C = NUM_CHANNELS;
#define ATOMIC_ADD(ptr, v) do { real v__ = (v); _Pragma("omp atomic") (ptr) += v__; } while (0)
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < grid_x * grid_y; tile++) {
        const int tx = tile % grid_x, ty = tile / grid_x;
        const uint32_t rx = ranges[2 * tile], ry = ranges[2 * tile + 1];
        const int toDo = (int)(ry - rx);
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                const int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
                if (!(px < W && py < H))
                    continue;
                const int pix_id = W * py + px;
                const real pixf[2] = {(real)px, (real)py};
                const real T_final = final_Ts[pix_id];
                real T = T_final;
                uint32_t contributor = (uint32_t)toDo;
                const int last_contributor = (int)n_contrib[pix_id];
                real accum_rec[NUM_CHANNELS] = {0};
                real dL_dpixel[NUM_CHANNELS];
                real accum_depth_rec = 0;
                for (int i = 0; i < C; i++)
                    dL_dpixel[i] = dL_dpixels[i * H * W + pix_id];
                real dL_dpixel_depth = dL_dpixel_depths[pix_id];
                real last_alpha = 0;
                real last_color[NUM_CHANNELS] = {0};
                real last_depth = 0;
                /* backward.cu:485-486 */
                const real ddelx_dx = (real)(0.5 * W);
                const real ddely_dy = (real)(0.5 * H);

                for (int k = 0; k < toDo; k++) {
                    /* back to front: backward.cu:497 */
                    const int global_id = (int)point_list[ry - k - 1];
                    contributor--;
                    if ((int)contributor >= last_contributor)
                        continue;
                    const real dx = points_xy_image[2 * global_id] - pixf[0];
                    const real dy = points_xy_image[2 * global_id + 1] - pixf[1];
                    const real* con_o = conic_opacity + 4 * global_id;
                    const real power = (real)-0.5 * (con_o[0] * dx * dx + con_o[2] * dy * dy) - con_o[1] * dx * dy;
                    if (power > (real)0.0)
                        continue;
                    const real G = R_EXP(power);
                    const real alpha = rmin((real)0.99f, con_o[3] * G);
                    if (alpha < (real)1.0 / (real)255.0)
                        continue;

                    T = T / ((real)1.0 - alpha);
                    const real dchannel_dcolor = alpha * T;
                    const real dpixel_depth_ddepth = alpha * T;

                    real dL_dalpha = 0;
                    for (int ch = 0; ch < C; ch++) {
                        const real c = colors[global_id * C + ch];
                        accum_rec[ch] = last_alpha * last_color[ch] + ((real)1.0 - last_alpha) * accum_rec[ch];
                        last_color[ch] = c;
                        const real dL_dchannel = dL_dpixel[ch];
                        dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
                        ATOMIC_ADD(dL_dcolors[global_id * C + ch], dchannel_dcolor * dL_dchannel);
                    }
                    const real c_d = depths[global_id];
                    accum_depth_rec = last_alpha * last_depth + ((real)1.0 - last_alpha) * accum_depth_rec;
                    last_depth = c_d;
                    dL_dalpha += (c_d - accum_depth_rec) * dL_dpixel_depth;
                    ATOMIC_ADD(dL_ddepths[global_id], dpixel_depth_ddepth * dL_dpixel_depth);

                    dL_dalpha *= T;
                    last_alpha = alpha;

                    real bg_dot_dpixel = 0;
                    for (int i = 0; i < C; i++)
                        bg_dot_dpixel += bg_color[i] * dL_dpixel[i];
                    dL_dalpha += (-T_final / ((real)1.0 - alpha)) * bg_dot_dpixel;

                    const real dL_dG = con_o[3] * dL_dalpha;
                    const real gdx = G * dx;
                    const real gdy = G * dy;
                    const real dG_ddelx = -gdx * con_o[0] - gdy * con_o[1];
                    const real dG_ddely = -gdy * con_o[2] - gdx * con_o[1];

                    ATOMIC_ADD(dL_dmean2D[3 * global_id + 0], dL_dG * dG_ddelx * ddelx_dx);
                    ATOMIC_ADD(dL_dmean2D[3 * global_id + 1], dL_dG * dG_ddely * ddely_dy);
                    ATOMIC_ADD(dL_dconic2D[4 * global_id + 0], (real)-0.5 * gdx * dx * dL_dG);
                    ATOMIC_ADD(dL_dconic2D[4 * global_id + 1], (real)-0.5 * gdx * dy * dL_dG);
                    ATOMIC_ADD(dL_dconic2D[4 * global_id + 3], (real)-0.5 * gdy * dy * dL_dG);
                    ATOMIC_ADD(dL_dopacity[global_id], G * dL_dalpha);
                }
            }
    }
#undef ATOMIC_ADD
}

/* backward.cu:20-139 */
static void computeColorFromSH_bwd(int idx, int deg, int max_coeffs, const real* means, const real* campos,
                                   const real* shs, const uint8_t* clamped, const real* dL_dcolor,
                                   real* dL_dmeans, real* dL_dshs)
{
    const real* pos = means + 3 * idx;
    real dir_orig[3] = {pos[0] - campos[0], pos[1] - campos[1], pos[2] - campos[2]};
    real len = R_SQRT(dir_orig[0] * dir_orig[0] + dir_orig[1] * dir_orig[1] + dir_orig[2] * dir_orig[2]);
    real dir[3] = {dir_orig[0] / len, dir_orig[1] / len, dir_orig[2] / len};
    const real* sh = shs + (size_t)idx * max_coeffs * 3;
    real dL_dRGB[3] = {dL_dcolor[3 * idx + 0], dL_dcolor[3 * idx + 1], dL_dcolor[3 * idx + 2]};
    dL_dRGB[0] *= clamped[3 * idx + 0] ? 0 : 1;
    dL_dRGB[1] *= clamped[3 * idx + 1] ? 0 : 1;
    dL_dRGB[2] *= clamped[3 * idx + 2] ? 0 : 1;
    real dRGBdx[3] = {0, 0, 0}, dRGBdy[3] = {0, 0, 0}, dRGBdz[3] = {0, 0, 0};
    real x = dir[0], y = dir[1], z = dir[2];
    real* dL_dsh = dL_dshs + (size_t)idx * max_coeffs * 3;
#define SH(i, c) sh[(i) * 3 + (c)]
#define DSH(i, s) for (int c_ = 0; c_ < 3; c_++) dL_dsh[(i) * 3 + c_] = (s) * dL_dRGB[c_]
    real dRGBdsh0 = SH_C0;
    DSH(0, dRGBdsh0);
    if (deg > 0) {
        real dRGBdsh1 = -SH_C1 * y;
        real dRGBdsh2 = SH_C1 * z;
        real dRGBdsh3 = -SH_C1 * x;
        DSH(1, dRGBdsh1);
        DSH(2, dRGBdsh2);
        DSH(3, dRGBdsh3);
        for (int c = 0; c < 3; c++) {
            dRGBdx[c] = -SH_C1 * SH(3, c);
            dRGBdy[c] = -SH_C1 * SH(1, c);
            dRGBdz[c] = SH_C1 * SH(2, c);
        }
        if (deg > 1) {
            real xx = x * x, yy = y * y, zz = z * z;
            real xy = x * y, yz = y * z, xz = x * z;
            real dRGBdsh4 = SH_C2[0] * xy;
            real dRGBdsh5 = SH_C2[1] * yz;
            real dRGBdsh6 = SH_C2[2] * ((real)2.0 * zz - xx - yy);
            real dRGBdsh7 = SH_C2[3] * xz;
            real dRGBdsh8 = SH_C2[4] * (xx - yy);
            DSH(4, dRGBdsh4);
            DSH(5, dRGBdsh5);
            DSH(6, dRGBdsh6);
            DSH(7, dRGBdsh7);
            DSH(8, dRGBdsh8);
            for (int c = 0; c < 3; c++) {
                dRGBdx[c] += SH_C2[0] * y * SH(4, c) + SH_C2[2] * (real)2.0 * -x * SH(6, c) + SH_C2[3] * z * SH(7, c) + SH_C2[4] * (real)2.0 * x * SH(8, c);
                dRGBdy[c] += SH_C2[0] * x * SH(4, c) + SH_C2[1] * z * SH(5, c) + SH_C2[2] * (real)2.0 * -y * SH(6, c) + SH_C2[4] * (real)2.0 * -y * SH(8, c);
                dRGBdz[c] += SH_C2[1] * y * SH(5, c) + SH_C2[2] * (real)2.0 * (real)2.0 * z * SH(6, c) + SH_C2[3] * x * SH(7, c);
            }
            if (deg > 2) {
                real dRGBdsh9 = SH_C3[0] * y * ((real)3.0 * xx - yy);
                real dRGBdsh10 = SH_C3[1] * xy * z;
                real dRGBdsh11 = SH_C3[2] * y * ((real)4.0 * zz - xx - yy);
                real dRGBdsh12 = SH_C3[3] * z * ((real)2.0 * zz - (real)3.0 * xx - (real)3.0 * yy);
                real dRGBdsh13 = SH_C3[4] * x * ((real)4.0 * zz - xx - yy);
                real dRGBdsh14 = SH_C3[5] * z * (xx - yy);
                real dRGBdsh15 = SH_C3[6] * x * (xx - (real)3.0 * yy);
                DSH(9, dRGBdsh9);
                DSH(10, dRGBdsh10);
                DSH(11, dRGBdsh11);
                DSH(12, dRGBdsh12);
                DSH(13, dRGBdsh13);
                DSH(14, dRGBdsh14);
                DSH(15, dRGBdsh15);
                for (int c = 0; c < 3; c++) {
                    dRGBdx[c] += (SH_C3[0] * SH(9, c) * (real)3.0 * (real)2.0 * xy +
                                  SH_C3[1] * SH(10, c) * yz +
                                  SH_C3[2] * SH(11, c) * (real)-2.0 * xy +
                                  SH_C3[3] * SH(12, c) * (real)-3.0 * (real)2.0 * xz +
                                  SH_C3[4] * SH(13, c) * ((real)-3.0 * xx + (real)4.0 * zz - yy) +
                                  SH_C3[5] * SH(14, c) * (real)2.0 * xz +
                                  SH_C3[6] * SH(15, c) * (real)3.0 * (xx - yy));
                    dRGBdy[c] += (SH_C3[0] * SH(9, c) * (real)3.0 * (xx - yy) +
                                  SH_C3[1] * SH(10, c) * xz +
                                  SH_C3[2] * SH(11, c) * ((real)-3.0 * yy + (real)4.0 * zz - xx) +
                                  SH_C3[3] * SH(12, c) * (real)-3.0 * (real)2.0 * yz +
                                  SH_C3[4] * SH(13, c) * (real)-2.0 * xy +
                                  SH_C3[5] * SH(14, c) * (real)-2.0 * yz +
                                  SH_C3[6] * SH(15, c) * (real)-3.0 * (real)2.0 * xy);
                    dRGBdz[c] += (SH_C3[1] * SH(10, c) * xy +
                                  SH_C3[2] * SH(11, c) * (real)4.0 * (real)2.0 * yz +
                                  SH_C3[3] * SH(12, c) * (real)3.0 * ((real)2.0 * zz - xx - yy) +
                                  SH_C3[4] * SH(13, c) * (real)4.0 * (real)2.0 * xz +
                                  SH_C3[5] * SH(14, c) * (xx - yy));
                }
            }
        }
    }
#undef SH
#undef DSH
    /* glm::dot of vec3: x+y+z left to right */
    real dL_ddir[3] = {
        dRGBdx[0] * dL_dRGB[0] + dRGBdx[1] * dL_dRGB[1] + dRGBdx[2] * dL_dRGB[2],
        dRGBdy[0] * dL_dRGB[0] + dRGBdy[1] * dL_dRGB[1] + dRGBdy[2] * dL_dRGB[2],
        dRGBdz[0] * dL_dRGB[0] + dRGBdz[1] * dL_dRGB[1] + dRGBdz[2] * dL_dRGB[2]};
    real dL_dmean[3];
    dnormvdv3(dir_orig, dL_ddir, dL_dmean);
    dL_dmeans[3 * idx + 0] += dL_dmean[0];
    dL_dmeans[3 * idx + 1] += dL_dmean[1];
    dL_dmeans[3 * idx + 2] += dL_dmean[2];
}

/* backward.cu:278-341 */
static void computeCov3D_bwd(int idx, const real* scale, real mod, const real* rot, const real* dL_dcov3Ds,
                             real* dL_dscales, real* dL_drots)
{
    real r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    mat3 R = mat3_cols(
        (real)1.0 - (real)2.0 * (y * y + z * z), (real)2.0 * (x * y - r * z), (real)2.0 * (x * z + r * y),
        (real)2.0 * (x * y + r * z), (real)1.0 - (real)2.0 * (x * x + z * z), (real)2.0 * (y * z - r * x),
        (real)2.0 * (x * z - r * y), (real)2.0 * (y * z + r * x), (real)1.0 - (real)2.0 * (x * x + y * y));
    mat3 S = mat3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
    real s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    S.m[0][0] = s[0]; S.m[1][1] = s[1]; S.m[2][2] = s[2];
    mat3 M = mat3_mul(&S, &R);
    const real* d = dL_dcov3Ds + 6 * idx;
    mat3 dL_dSigma = mat3_cols(d[0], (real)0.5 * d[1], (real)0.5 * d[2],
                               (real)0.5 * d[1], d[3], (real)0.5 * d[4],
                               (real)0.5 * d[2], (real)0.5 * d[4], d[5]);
    /* 2.0f * M * dL_dSigma  ==  (2.0f * M) * dL_dSigma */
    mat3 M2;
    for (int c = 0; c < 3; c++)
        for (int rr = 0; rr < 3; rr++)
            M2.m[c][rr] = M.m[c][rr] * (real)2.0;
    mat3 dL_dM = mat3_mul(&M2, &dL_dSigma);
    mat3 Rt = mat3_T(&R);
    mat3 dL_dMt = mat3_T(&dL_dM);
    real* dL_dscale = dL_dscales + 3 * idx;
    for (int c = 0; c < 3; c++)
        dL_dscale[c] = Rt.m[c][0] * dL_dMt.m[c][0] + Rt.m[c][1] * dL_dMt.m[c][1] + Rt.m[c][2] * dL_dMt.m[c][2];
    for (int c = 0; c < 3; c++)
        for (int rr = 0; rr < 3; rr++)
            dL_dMt.m[c][rr] *= s[c];
#define DM(c, rr) dL_dMt.m[c][rr]
    real dq[4];
    dq[0] = 2 * z * (DM(0, 1) - DM(1, 0)) + 2 * y * (DM(2, 0) - DM(0, 2)) + 2 * x * (DM(1, 2) - DM(2, 1));
    dq[1] = 2 * y * (DM(1, 0) + DM(0, 1)) + 2 * z * (DM(2, 0) + DM(0, 2)) + 2 * r * (DM(1, 2) - DM(2, 1)) - 4 * x * (DM(2, 2) + DM(1, 1));
    dq[2] = 2 * x * (DM(1, 0) + DM(0, 1)) + 2 * r * (DM(2, 0) - DM(0, 2)) + 2 * z * (DM(1, 2) + DM(2, 1)) - 4 * y * (DM(2, 2) + DM(0, 0));
    dq[3] = 2 * r * (DM(0, 1) - DM(1, 0)) + 2 * x * (DM(2, 0) + DM(0, 2)) + 2 * y * (DM(1, 2) + DM(2, 1)) - 4 * z * (DM(1, 1) + DM(0, 0));
#undef DM
    real* dL_drot = dL_drots + 4 * idx;
    dL_drot[0] = dq[0]; dL_drot[1] = dq[1]; dL_drot[2] = dq[2]; dL_drot[3] = dq[3];
}

/* BACKWARD::preprocess (backward.cu:592-658) = computeCov2DCUDA (:144-274)
 * followed by preprocessCUDA (:346-412).  dL_dconic is [P,4] (indices 0,1,3
 * read, :165); dL_dmean2D is [P,3]. */
void oracle_preprocess_backward(int P, int D, int M, const real* means3D, const int* radii, const real* shs,
                                const uint8_t* clamped, const real* scales, const real* rotations,
                                real scale_modifier, const real* cov3Ds, const real* viewmatrix,
                                const real* projmatrix, int W, int H, real tan_fovx, real tan_fovy,
                                const real* campos, const real* dL_dmean2D, const real* dL_dconics,
                                real* dL_dmeans, real* dL_dcolor, real* dL_ddepth, real* dL_dcov,
                                real* dL_dsh, real* dL_dscale, real* dL_drot)
{
    const real h_y = H / ((real)2.0 * tan_fovy); /* rasterizer_impl.cu:385-386 */
    const real h_x = W / ((real)2.0 * tan_fovx);
    const real* view_matrix = viewmatrix;
    const real* proj = projmatrix;
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0))
            continue;
        /* ---- computeCov2DCUDA ---- */
        const real* cov3D = cov3Ds + 6 * idx;
        const real* mean = means3D + 3 * idx;
        real dL_dconic[3] = {dL_dconics[4 * idx], dL_dconics[4 * idx + 1], dL_dconics[4 * idx + 3]};
        mat3 T, Vrk, cov2D;
        real t[3], txtz, tytz;
        cov2D_TJ(mean, h_x, h_y, tan_fovx, tan_fovy, cov3D, view_matrix, &T, &Vrk, &cov2D, t, &txtz, &tytz);
        const real limx = (real)1.3f * tan_fovx;
        const real limy = (real)1.3f * tan_fovy;
        const real x_grad_mul = (txtz < -limx || txtz > limx) ? 0 : 1;
        const real y_grad_mul = (tytz < -limy || tytz > limy) ? 0 : 1;
        mat3 Wm = mat3_cols(view_matrix[0], view_matrix[4], view_matrix[8],
                            view_matrix[1], view_matrix[5], view_matrix[9],
                            view_matrix[2], view_matrix[6], view_matrix[10]);

        real a = cov2D.m[0][0] += (real)0.3f;
        real b = cov2D.m[0][1];
        real c = cov2D.m[1][1] += (real)0.3f;
        real denom = a * c - b * b;
        real dL_da = 0, dL_db = 0, dL_dc = 0;
        real denom2inv = (real)1.0 / ((denom * denom) + (real)0.0000001f);
#define TT(cc, rr) T.m[cc][rr]
#define VV(cc, rr) Vrk.m[cc][rr]
#define WW(cc, rr) Wm.m[cc][rr]
        if (denom2inv != 0) {
            dL_da = denom2inv * (-c * c * dL_dconic[0] + 2 * b * c * dL_dconic[1] + (denom - a * c) * dL_dconic[2]);
            dL_dc = denom2inv * (-a * a * dL_dconic[2] + 2 * a * b * dL_dconic[1] + (denom - a * c) * dL_dconic[0]);
            dL_db = denom2inv * 2 * (b * c * dL_dconic[0] - (denom + 2 * b * b) * dL_dconic[1] + a * b * dL_dconic[2]);
            dL_dcov[6 * idx + 0] = (TT(0, 0) * TT(0, 0) * dL_da + TT(0, 0) * TT(1, 0) * dL_db + TT(1, 0) * TT(1, 0) * dL_dc);
            dL_dcov[6 * idx + 3] = (TT(0, 1) * TT(0, 1) * dL_da + TT(0, 1) * TT(1, 1) * dL_db + TT(1, 1) * TT(1, 1) * dL_dc);
            dL_dcov[6 * idx + 5] = (TT(0, 2) * TT(0, 2) * dL_da + TT(0, 2) * TT(1, 2) * dL_db + TT(1, 2) * TT(1, 2) * dL_dc);
            dL_dcov[6 * idx + 1] = 2 * TT(0, 0) * TT(0, 1) * dL_da + (TT(0, 0) * TT(1, 1) + TT(0, 1) * TT(1, 0)) * dL_db + 2 * TT(1, 0) * TT(1, 1) * dL_dc;
            dL_dcov[6 * idx + 2] = 2 * TT(0, 0) * TT(0, 2) * dL_da + (TT(0, 0) * TT(1, 2) + TT(0, 2) * TT(1, 0)) * dL_db + 2 * TT(1, 0) * TT(1, 2) * dL_dc;
            dL_dcov[6 * idx + 4] = 2 * TT(0, 2) * TT(0, 1) * dL_da + (TT(0, 1) * TT(1, 2) + TT(0, 2) * TT(1, 1)) * dL_db + 2 * TT(1, 1) * TT(1, 2) * dL_dc;
        } else {
            for (int i = 0; i < 6; i++)
                dL_dcov[6 * idx + i] = 0;
        }
        real dL_dT00 = 2 * (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_da +
                       (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_db;
        real dL_dT01 = 2 * (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_da +
                       (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_db;
        real dL_dT02 = 2 * (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_da +
                       (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_db;
        real dL_dT10 = 2 * (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_dc +
                       (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_db;
        real dL_dT11 = 2 * (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_dc +
                       (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_db;
        real dL_dT12 = 2 * (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_dc +
                       (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_db;
        real dL_dJ00 = WW(0, 0) * dL_dT00 + WW(0, 1) * dL_dT01 + WW(0, 2) * dL_dT02;
        real dL_dJ02 = WW(2, 0) * dL_dT00 + WW(2, 1) * dL_dT01 + WW(2, 2) * dL_dT02;
        real dL_dJ11 = WW(1, 0) * dL_dT10 + WW(1, 1) * dL_dT11 + WW(1, 2) * dL_dT12;
        real dL_dJ12 = WW(2, 0) * dL_dT10 + WW(2, 1) * dL_dT11 + WW(2, 2) * dL_dT12;
#undef TT
#undef VV
#undef WW
        real tz = (real)1.0 / t[2];
        real tz2 = tz * tz;
        real tz3 = tz2 * tz;
        real dL_dt[3];
        dL_dt[0] = x_grad_mul * -h_x * tz2 * dL_dJ02;
        dL_dt[1] = y_grad_mul * -h_y * tz2 * dL_dJ12;
        dL_dt[2] = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t[0]) * tz3 * dL_dJ02 + (2 * h_y * t[1]) * tz3 * dL_dJ12;
        real dL_dmean_cov[3];
        transformVec4x3Transpose(dL_dt, view_matrix, dL_dmean_cov);
        /* backward.cu:273 -- ASSIGNS */
        dL_dmeans[3 * idx + 0] = dL_dmean_cov[0];
        dL_dmeans[3 * idx + 1] = dL_dmean_cov[1];
        dL_dmeans[3 * idx + 2] = dL_dmean_cov[2];

        /* ---- preprocessCUDA (backward.cu:346-412) ---- */
        const real* m = mean;
        real m_hom[4];
        transformPoint4x4(m, proj, m_hom);
        real m_w = (real)1.0 / (m_hom[3] + (real)0.0000001f);
        real mul1 = (proj[0] * m[0] + proj[4] * m[1] + proj[8] * m[2] + proj[12]) * m_w * m_w;
        real mul2 = (proj[1] * m[0] + proj[5] * m[1] + proj[9] * m[2] + proj[13]) * m_w * m_w;
        const real g2x = dL_dmean2D[3 * idx + 0], g2y = dL_dmean2D[3 * idx + 1];
        real dL_dmean[3];
        dL_dmean[0] = (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
        dL_dmean[1] = (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
        dL_dmean[2] = (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
        dL_dmeans[3 * idx + 0] += dL_dmean[0];
        dL_dmeans[3 * idx + 1] += dL_dmean[1];
        dL_dmeans[3 * idx + 2] += dL_dmean[2];

        const real* view = view_matrix;
        real mul3 = view[2] * m[0] + view[6] * m[1] + view[10] * m[2] + view[14];
        real dL_dmean2[3];
        dL_dmean2[0] = (view[2] - view[3] * mul3) * dL_ddepth[idx];
        dL_dmean2[1] = (view[6] - view[7] * mul3) * dL_ddepth[idx];
        dL_dmean2[2] = (view[10] - view[11] * mul3) * dL_ddepth[idx];
        dL_dmeans[3 * idx + 0] += dL_dmean2[0];
        dL_dmeans[3 * idx + 1] += dL_dmean2[1];
        dL_dmeans[3 * idx + 2] += dL_dmean2[2];

        if (shs)
            computeColorFromSH_bwd(idx, D, M, means3D, campos, shs, clamped, dL_dcolor, dL_dmeans, dL_dsh);
        if (scales)
            computeCov3D_bwd(idx, scales + 3 * idx, scale_modifier, rotations + 4 * idx, dL_dcov, dL_dscale, dL_drot);
    }
}

/* rasterizer_impl.cu:54-66 checkFrustum / markVisible */
void oracle_mark_visible(int P, const real* means3D, const real* viewmatrix, const real* projmatrix, uint8_t* present)
{
    (void)projmatrix;
    for (int idx = 0; idx < P; idx++) {
        real p_view[3];
        present[idx] = (uint8_t)in_frustum(idx, means3D, viewmatrix, p_view);
    }
}

/* ---------------------------------------------------------------------------
 * simple-knn (simple_knn.cu).  fp32 only.
 * ------------------------------------------------------------------------- */
#ifndef ORACLE_FP64
#define BOX_SIZE 1024
/* simple_knn.cu:45-52 */
static uint32_t prepMorton(uint32_t x)
{
    x = (x | (x << 16)) & 0x030000FF;
    x = (x | (x << 8)) & 0x0300F00F;
    x = (x | (x << 4)) & 0x030C30C3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}
/* simple_knn.cu:54-61 */
static uint32_t coord2Morton(const float* c, const float* minn, const float* maxx)
{
    uint32_t x = prepMorton((uint32_t)(((c[0] - minn[0]) / (maxx[0] - minn[0])) * ((1 << 10) - 1)));
    uint32_t y = prepMorton((uint32_t)(((c[1] - minn[1]) / (maxx[1] - minn[1])) * ((1 << 10) - 1)));
    uint32_t z = prepMorton((uint32_t)(((c[2] - minn[2]) / (maxx[2] - minn[2])) * ((1 << 10) - 1)));
    return x | (y << 1) | (z << 2);
}
typedef struct { float minn[3]; float maxx[3]; } MinMax;
/* simple_knn.cu:119-129 */
static float distBoxPoint(const MinMax* box, const float* p)
{
    float diff[3] = {0, 0, 0};
    for (int a = 0; a < 3; a++)
        if (p[a] < box->minn[a] || p[a] > box->maxx[a])
            diff[a] = fminf(fabsf(p[a] - box->minn[a]), fabsf(p[a] - box->maxx[a]));
    return diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2];
}
/* simple_knn.cu:131-145 */
static void updateKBest3(const float* ref, const float* point, float* knn)
{
    float d[3] = {point[0] - ref[0], point[1] - ref[1], point[2] - ref[2]};
    float dist = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    for (int j = 0; j < 3; j++) {
        if (knn[j] > dist) {
            float t = knn[j];
            knn[j] = dist;
            dist = t;
        }
    }
}
/* SimpleKNN::knn (simple_knn.cu:185-221) */
void oracle_knn(int P, const float* points, float* meanDists)
{
    /* cub::DeviceReduce::Reduce(..., CustomMin/Max, init = {0,0,0})  (:192-201) */
    float minn[3] = {0, 0, 0}, maxx[3] = {0, 0, 0};
    for (int i = 0; i < P; i++)
        for (int a = 0; a < 3; a++) {
            minn[a] = fminf(minn[a], points[3 * i + a]);
            maxx[a] = fmaxf(maxx[a], points[3 * i + a]);
        }
    uint64_t* keys = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(P > 0 ? P : 1));
    uint64_t* keys_s = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(P > 0 ? P : 1));
    uint32_t* idx = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(P > 0 ? P : 1));
    uint32_t* idx_s = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(P > 0 ? P : 1));
    for (int i = 0; i < P; i++) {
        keys[i] = coord2Morton(points + 3 * i, minn, maxx);
        idx[i] = (uint32_t)i; /* thrust::sequence */
    }
    radix_sort_pairs(keys, keys_s, idx, idx_s, P, 32); /* stable SortPairs on the 32-bit codes */
    uint32_t num_boxes = (uint32_t)((P + BOX_SIZE - 1) / BOX_SIZE);
    MinMax* boxes = (MinMax*)malloc(sizeof(MinMax) * (num_boxes > 0 ? num_boxes : 1));
    /* boxMinMax (:78-117) */
    for (uint32_t b = 0; b < num_boxes; b++) {
        MinMax me;
        for (int a = 0; a < 3; a++) { me.minn[a] = FLT_MAX; me.maxx[a] = -FLT_MAX; }
        for (int i = (int)b * BOX_SIZE; i < imin(P, (int)(b + 1) * BOX_SIZE); i++) {
            const float* p = points + 3 * idx_s[i];
            for (int a = 0; a < 3; a++) {
                me.minn[a] = fminf(me.minn[a], p[a]);
                me.maxx[a] = fmaxf(me.maxx[a], p[a]);
            }
        }
        boxes[b] = me;
    }
    /* boxMeanDist (:147-183) */
#pragma omp parallel for schedule(dynamic, 256)
    for (int i0 = 0; i0 < P; i0++) {
        const float* point = points + 3 * idx_s[i0];
        float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
        for (int i = imax(0, i0 - 3); i <= imin(P - 1, i0 + 3); i++) {
            if (i == i0)
                continue;
            updateKBest3(point, points + 3 * idx_s[i], best);
        }
        float reject = best[2];
        best[0] = FLT_MAX; best[1] = FLT_MAX; best[2] = FLT_MAX;
        for (int b = 0; b < (P + BOX_SIZE - 1) / BOX_SIZE; b++) {
            MinMax box = boxes[b];
            float dist = distBoxPoint(&box, point);
            if (dist > reject || dist > best[2])
                continue;
            for (int i = b * BOX_SIZE; i < imin(P, (b + 1) * BOX_SIZE); i++) {
                if (i == i0)
                    continue;
                updateKBest3(point, points + 3 * idx_s[i], best);
            }
        }
        meanDists[idx_s[i0]] = (best[0] + best[1] + best[2]) / 3.0f;
    }
    free(keys); free(keys_s); free(idx); free(idx_s); free(boxes);
}
#endif /* !ORACLE_FP64 */
