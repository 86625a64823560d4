// extern "C" entry points of the rasterizer (see include/mom4d.h).
#include "mom_common.h"
#include <stdlib.h>

int mom_launch_preprocess_fwd(const MomRasterArgs* a, const GeomView& g, int* radii, uint32_t* zero_words, int n_zero,
                              uint32_t* hist_counts, bool* did_hist, hipStream_t s);
int mom_launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t s);
int mom_launch_binning_count(const MomRasterArgs* a, const GeomView& g, const ImageView& im, uint32_t* num_rendered_dev,
                             uint32_t* num_rendered_host, bool hist_done, hipStream_t s);
int mom_launch_binning_sort(const MomRasterArgs* a, const GeomView& g, const BinView& b, const ImageView& im, size_t capacity,
                            uint32_t* status_dev, bool render_sorts_small, hipStream_t s);
int mom_launch_render_fwd(const MomRasterArgs* a, const GeomView& g, const BinView& b, const ImageView& im, size_t capacity,
                          float* out_color, float* out_depth, bool sort_small, hipStream_t s);
int mom_launch_render_bwd(const MomRasterArgs* a, const GeomView& g, const BinView& b, const ImageView& im, size_t capacity,
                          const float* dL_dpix, const float* dL_ddepth, hipStream_t s);
int mom_launch_preprocess_bwd(const MomRasterArgs* a, const int* radii, const GeomView& g, const MomRasterGrads* gr, hipStream_t s);

static int check_args(const MomRasterArgs* a)
{
    if (!a || a->struct_size != sizeof(MomRasterArgs)) return MOM_EINVAL;   // a binder built against another header: refuse
    if (a->P < 0 || a->W <= 0 || a->H <= 0) return MOM_EINVAL;
    if (a->tile_row0 < 0 || a->tile_row1 < a->tile_row0 || a->tile_row1 > (a->H + MOM_TILE - 1) / MOM_TILE) return MOM_EINVAL;
    if (a->P == 0) return MOM_OK;
    if (!a->means3D || !a->opacities || !a->viewmatrix || !a->projmatrix || !a->campos || !a->background) return MOM_EINVAL;
    if (!a->shs && !a->colors_precomp) return MOM_EINVAL;  // rasterizer_impl.cu:243-246
    if (!a->cov3D_precomp && (!a->scales || !a->rotations)) return MOM_EINVAL;
    if (a->shs && !a->colors_precomp && (a->D < 0 || a->D > 3 || a->M < (a->D + 1) * (a->D + 1))) return MOM_EINVAL;
    return MOM_OK;
}

static int check_grads(const MomRasterArgs* a, const MomRasterGrads* gr)
{
    if (!gr) return MOM_EINVAL;
    if (!gr->dL_dmeans2D || !gr->dL_dcolors || !gr->dL_dopacity || !gr->dL_dmeans3D || !gr->dL_dcov3D) return MOM_EINVAL;
    if (a->M > 0 && !a->colors_precomp && !gr->dL_dsh) return MOM_EINVAL;
    if (a->shs_rest && !gr->dL_dsh_rest) return MOM_EINVAL;
    if (a->scales && (!gr->dL_dscales || !gr->dL_drotations)) return MOM_EINVAL;
    if (gr->act_rotations_raw && (!a->scales || !a->rotations || a->cov3D_precomp)) return MOM_EINVAL;   // activations of inputs that are not there
    return MOM_OK;
}

extern "C" {

#ifndef MOM_SRC_HASH
#define MOM_SRC_HASH "unknown"
#endif
const char* mom_version(void) { return "mom4d 0.4 (gfx950) src " MOM_SRC_HASH; }
int mom_abi_version(void) { return MOM_ABI_VERSION; }
size_t mom_abi_sizeof(int which)
{
    switch (which) {
    case MOM_STRUCT_RASTER_ARGS: return sizeof(MomRasterArgs);
    case MOM_STRUCT_RASTER_GRADS: return sizeof(MomRasterGrads);
    case MOM_STRUCT_RASTER_LAYOUT: return sizeof(MomRasterLayout);
    case MOM_STRUCT_HEXPLANE: return sizeof(MomHexPlane);
    case MOM_STRUCT_ADAM_TENSOR: return sizeof(MomAdamTensor);
    case MOM_STRUCT_ROW_SELECT: return sizeof(MomRowSelect);
    case MOM_STRUCT_REG_PLANE: return sizeof(MomRegPlane);
    case MOM_STRUCT_DEFORM_MLP: return sizeof(MomDeformMLP);
    default: return 0;
    }
}

size_t mom_raster_geom_bytes(int P) { return geom_view(nullptr, P, nullptr) + MOM_ALIGN; }
size_t mom_raster_image_bytes(int W, int H) { return image_view(nullptr, W, H, nullptr) + MOM_ALIGN; }
size_t mom_raster_binning_bytes(int P, int W, int H, size_t capacity)
{
    (void)P; (void)W; (void)H;
    return bin_view(nullptr, capacity, nullptr) + MOM_ALIGN;
}

int mom_raster_layout(int P, int W, int H, size_t capacity, MomRasterLayout* out)
{
    if (!out) return MOM_EINVAL;
    // offsets are relative to the 256-byte aligned base of each buffer
    GeomView g; ImageView im; BinView b;
    char* z = (char*)0;
    geom_view(z, P, &g);
    image_view(z, W, H, &im);
    bin_view(z, capacity, &b);
    out->geom_rec = (size_t)((char*)g.rec - z);
    out->geom_cov3D = (size_t)((char*)g.cov3D - z);
    out->geom_clamped = (size_t)((char*)g.clamped - z);
    out->geom_gacc = (size_t)((char*)g.gacc - z);
    out->img_ranges = (size_t)((char*)im.ranges - z);
    out->img_n_contrib = (size_t)((char*)im.n_contrib - z);
    out->img_final_T = (size_t)((char*)im.final_T - z);
    out->bin_keys = (size_t)((char*)b.keys - z);
    out->bin_point_list = (size_t)((char*)b.point_list - z);
    out->img_tile_counts = (size_t)((char*)im.tile_counts - z);
    out->img_tile_walked = (size_t)((char*)im.tile_cursor - z);
    return MOM_OK;
}

int mom_raster_forward_geometry(const MomRasterArgs* a, void* geom, void* image, int* radii,
                                uint32_t* num_rendered_dev, uint32_t* num_rendered_host, mom_stream_t stream)
{
    int rc = check_args(a);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (!num_rendered_dev) return MOM_EINVAL;
    if (a->P == 0) {  // rasterize_points.cu:82: P == 0 short-circuits
        if (hipMemsetAsync(num_rendered_dev, 0, 4, s) != hipSuccess) return MOM_ELAUNCH;
        if (num_rendered_host) *num_rendered_host = 0;
        return MOM_OK;
    }
    if (!geom || !image || !radii) return MOM_EINVAL;
    GeomView g; ImageView im;
    geom_view(mom_align_ptr(geom), a->P, &g);
    image_view(mom_align_ptr(image), a->W, a->H, &im);
    // the header and the tile counters are adjacent in the image scratch (image_view): the projection kernel clears both
    const int tiles = ((a->W + MOM_TILE - 1) / MOM_TILE) * ((a->H + MOM_TILE - 1) / MOM_TILE);
    bool did_hist = false;
    static int fold = -1;                 // MOM_FOLD_HIST=1: the tile histogram inside the projection kernel (measured: slower, see raster_preprocess.hip)
    if (fold < 0) { const char* e = getenv("MOM_FOLD_HIST"); fold = (e && e[0] == '1') ? 1 : 0; }
    rc = mom_launch_preprocess_fwd(a, g, radii, im.hdr, (int)((im.tile_counts + tiles) - im.hdr), fold ? im.tile_counts : nullptr, &did_hist, s);
    if (rc) return rc;
    MOM_CHECK_LAUNCH(a, s);
    rc = mom_launch_binning_count(a, g, im, num_rendered_dev, num_rendered_host, did_hist, s);
    if (rc) return rc;
    MOM_CHECK_LAUNCH(a, s);
    return MOM_OK;
}

int mom_raster_forward_render(const MomRasterArgs* a, void* geom, void* binning, size_t capacity, void* image,
                              float* out_color, float* out_depth, uint32_t* status_dev, mom_stream_t stream)
{
    int rc = check_args(a);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (!out_color || !out_depth) return MOM_EINVAL;
    const size_t HW = (size_t)a->W * a->H;
    if (a->P == 0) {
        if (hipMemsetAsync(out_color, 0, HW * 12, s) != hipSuccess) return MOM_ELAUNCH;
        if (hipMemsetAsync(out_depth, 0, HW * 4, s) != hipSuccess) return MOM_ELAUNCH;
        return MOM_OK;
    }
    if (!geom || !binning || !image) return MOM_EINVAL;
    GeomView g; ImageView im; BinView b;
    geom_view(mom_align_ptr(geom), a->P, &g);
    image_view(mom_align_ptr(image), a->W, a->H, &im);
    bin_view(mom_align_ptr(binning), capacity, &b);
    // MOM_RENDER_SORT=0: every tile sorted by the binning's own launches (measurement, and the comparison in tests/test_raster_gpu.py;
    // read per call for that)
    const char* e_sort = getenv("MOM_RENDER_SORT");
    const int merged = (e_sort && e_sort[0] == '0') ? 0 : 1;
    rc = mom_launch_binning_sort(a, g, b, im, capacity, status_dev, merged != 0, s);
    if (rc) return rc;
    MOM_CHECK_LAUNCH(a, s);
    rc = mom_launch_render_fwd(a, g, b, im, capacity, out_color, out_depth, merged != 0, s);
    if (rc) return rc;
    MOM_CHECK_LAUNCH(a, s);
    return MOM_OK;
}

int mom_raster_backward_render(const MomRasterArgs* a, void* geom, void* binning, size_t capacity, void* image,
                               const float* dL_dout_color, const float* dL_dout_depth, mom_stream_t stream)
{
    int rc = check_args(a);
    if (rc) return rc;
    if (a->P == 0) return MOM_OK;
    hipStream_t s = (hipStream_t)stream;
    if (!geom || !binning || !image || !dL_dout_color) return MOM_EINVAL;
    GeomView g; ImageView im; BinView b;
    geom_view(mom_align_ptr(geom), a->P, &g);
    image_view(mom_align_ptr(image), a->W, a->H, &im);
    bin_view(mom_align_ptr(binning), capacity, &b);
    rc = mom_launch_render_bwd(a, g, b, im, capacity, dL_dout_color, dL_dout_depth, s);
    if (rc) return rc;
    MOM_CHECK_LAUNCH(a, s);
    return MOM_OK;
}

int mom_raster_backward_geometry(const MomRasterArgs* a, const int* radii, void* geom, const MomRasterGrads* gr, mom_stream_t stream)
{
    int rc = check_args(a);
    if (rc) return rc;
    if (a->P == 0) return MOM_OK;
    hipStream_t s = (hipStream_t)stream;
    if (!geom || !radii) return MOM_EINVAL;
    rc = check_grads(a, gr);
    if (rc) return rc;
    GeomView g;
    geom_view(mom_align_ptr(geom), a->P, &g);
    rc = mom_launch_preprocess_bwd(a, radii, g, gr, s);
    if (rc) return rc;
    MOM_CHECK_LAUNCH(a, s);
    return MOM_OK;
}

int mom_raster_backward(const MomRasterArgs* a, const int* radii, void* geom, void* binning, size_t capacity, void* image,
                        const float* dL_dout_color, const float* dL_dout_depth, const MomRasterGrads* gr, mom_stream_t stream)
{
    int rc = check_args(a);                     // every argument is checked before anything is launched
    if (rc) return rc;
    if (a->P == 0) return MOM_OK;
    if (!geom || !binning || !image || !radii || !dL_dout_color) return MOM_EINVAL;
    rc = check_grads(a, gr);
    if (rc) return rc;
    rc = mom_raster_backward_render(a, geom, binning, capacity, image, dL_dout_color, dL_dout_depth, stream);
    if (rc) return rc;
    return mom_raster_backward_geometry(a, radii, geom, gr, stream);
}

int mom_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                     mom_stream_t stream)
{
    (void)projmatrix;
    if (P < 0) return MOM_EINVAL;
    if (P == 0) return MOM_OK;
    if (!means3D || !viewmatrix || !present) return MOM_EINVAL;
    return mom_launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream);
}

}  // extern "C"
