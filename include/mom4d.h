/*
 * mom4d.h -- C ABI of libmom4d.so, the MI355X (gfx950) hot path of the 4D Gaussian
 * Splatting train/render loop of cvsp-lab/ICLR2025_3D-MOM.
 *
 * Every entry point takes plain device pointers and sizes plus a hipStream_t
 * (passed as void*), is stream-ordered on that stream, never allocates, never
 * synchronises the host unless its comment says so, and returns 0 on success
 * or a negative MOM_E* code (it never throws across the ABI).
 *
 * Reference interfaces replaced (paths relative to the reference tree):
 *   CudaRasterizer::Rasterizer::forward / backward / markVisible
 *       submodules/depth-diff-gaussian-rasterization/cuda_rasterizer/rasterizer.h:19-87
 *       (implementation rasterizer_impl.cu:141-153,198-444)
 *   RasterizeGaussiansCUDA / RasterizeGaussiansBackwardCUDA / markVisible (torch glue)
 *       submodules/depth-diff-gaussian-rasterization/rasterize_points.h:18-68
 *   distCUDA2 / SimpleKNN::knn
 *       submodules/simple-knn/spatial.h, simple_knn.h  (simple_knn.cu:185-221)
 * and, one level up (torch ops the reference issues from Python):
 *   deform_network.forward           scene/deformation.py:190-223 + scene/hexplane.py:160-183
 *   torch.optim.Adam.step            scene/gaussian_model.py:209, train_4DGS.py:295-297
 *   l1_loss / ssim                   utils/loss_utils.py:23-24,52-92
 *   compute_regulation               scene/gaussian_model.py:730-769
 *   boolean-mask row compaction      scene/gaussian_model.py:424-459
 */
#ifndef MOM4D_H_INCLUDED
#define MOM4D_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOM_OK 0
#define MOM_EINVAL (-1)   /* bad argument (shape, null pointer, unsupported channel count) */
#define MOM_ELAUNCH (-2)  /* a HIP launch / runtime call failed (hipGetLastError) */
#define MOM_ECAPACITY (-3)/* scratch buffer too small */
#define MOM_EUNAVAILABLE (-4) /* an optional run-time dependency is missing (librccl for mom_comm_*): mom_comm_last_error() says which */

#define MOM_TILE 16       /* config.h:14-16 BLOCK_X == BLOCK_Y == 16 */

typedef void* mom_stream_t; /* hipStream_t */

/* ---- ABI versioning ------------------------------------------------------------------------
 * The argument structs below have grown between releases.  A binder built against an older header must not be able to
 * pass a short struct silently, so:
 *   - mom_abi_version() returns MOM_ABI_VERSION of the library that is loaded; a binder compares it with the value of
 *     the header (or Python mirror) it was written against and refuses to continue on a mismatch;
 *   - mom_abi_sizeof(which) returns sizeof of each argument struct as the library was compiled, so a foreign-language
 *     mirror (ctypes, cffi) can verify its own layout field for field at load time;
 *   - MomRasterArgs, the struct that changes most often, additionally starts with `struct_size`: every entry point that
 *     takes it returns MOM_EINVAL unless struct_size == sizeof(MomRasterArgs) of the library.
 * (The reference's counterpart is a C++ static-method signature, rasterizer.h:19-87: there the compiler checks it.) */
#define MOM_ABI_VERSION 7
/* floats per Gaussian of the compositing backward's accumulator record (mom_raster_layout().geom_gacc); ten are used.  (A build with
 * -DMOM_GACC_FLOATS=16 pads the record to one 64-byte line: a device-scope atomic costs the device by the LINES an instruction
 * touches and 48-byte records straddle 1.5 on average -- but the shipped kernel's atomics hide under its arithmetic, and the larger
 * record measured 3 us SLOWER in render_bwd and 5 us per step, three alternating pairs in one call: DESIGN.md section 3.1.) */
#ifndef MOM_GACC_FLOATS
#define MOM_GACC_FLOATS 12
#endif
int mom_abi_version(void);
enum {
    MOM_STRUCT_RASTER_ARGS = 0, MOM_STRUCT_RASTER_GRADS, MOM_STRUCT_RASTER_LAYOUT, MOM_STRUCT_HEXPLANE, MOM_STRUCT_ADAM_TENSOR,
    MOM_STRUCT_ROW_SELECT, MOM_STRUCT_REG_PLANE, MOM_STRUCT_DEFORM_MLP, MOM_STRUCT_COUNT
};
size_t mom_abi_sizeof(int which);   /* 0 for an unknown id */

/* Arguments of one rasterizer call; the field meanings are those of
 * CudaRasterizer::Rasterizer::forward (rasterizer.h:29-55).  Null pointers stand
 * for "absent" exactly as empty tensors do in the reference (forward.cu:205,241).
 * Matrices are the transposed (column-major) 4x4s the reference passes. */
typedef struct MomRasterArgs {
    uint32_t struct_size;  /* = sizeof(MomRasterArgs) of the header the caller was built against; anything else: MOM_EINVAL */
    int P;                 /* number of Gaussians */
    int D;                 /* active SH degree (0..3) */
    int M;                 /* SH coefficients per Gaussian in `shs` (0 if absent) */
    int W, H;              /* image size in pixels */
    const float* background;     /* [3] */
    const float* means3D;        /* [P,3] */
    const float* shs;            /* [P,M,3] or null; if shs_rest != null: the DC coefficient only, [P,1,3] */
    const float* shs_rest;       /* null, or [P,M-1,3]: coefficients 1..M-1 stored apart (GaussianModel keeps
                                    _features_dc and _features_rest as two tensors; this avoids the torch.cat of
                                    get_features, scene/gaussian_model.py:137-140) */
    const float* colors_precomp; /* [P,3] or null */
    const float* opacities;      /* [P] (already activated) */
    const float* scales;         /* [P,3] or null */
    const float* rotations;      /* [P,4] (r,x,y,z), used as given, or null */
    const float* cov3D_precomp;  /* [P,6] or null */
    const float* viewmatrix;     /* [16] */
    const float* projmatrix;     /* [16] */
    const float* campos;         /* [3] */
    float scale_modifier;
    float tan_fovx, tan_fovy;
    int prefiltered;
    int debug;             /* !=0: hipStreamSynchronize + error check after every kernel (CHECK_CUDA, auxiliary.h:166-173) */
    /* Tile-row shard (one image split over several GPUs; the reference has no counterpart): bin, composite and
     * back-propagate only the 16-pixel tile rows [tile_row0, tile_row1).  0,0 = every row.  Projection is
     * unaffected (radii, depths, conics are those of the whole image); num_rendered counts the local instances;
     * pixels of other rows are neither read nor written. */
    int tile_row0, tile_row1;
    /* !=0: this forward will never be followed by a backward (no-grad render(), render_4DGS.py): the auxiliary state only the
     * backward reads -- cov3D[P][6], clamped[P], final_T[H*W], n_contrib[H*W] -- is not written (about 28 B per Gaussian and
     * 8 B per pixel).  mom_raster_backward on such a forward is an error the caller must not make. */
    int forward_only;
    /* What mom_raster_forward_render leaves in *status_dev when this call's binning overflows (0 = 1): a caller that runs
     * ahead of the GPU numbers its calls here and later reads WHICH call overflowed first. */
    uint32_t overflow_tag;
    /* 0 (default): a (splat, tile) instance is binned only if the splat can reach alpha >= 1/255 at some pixel of the tile --
     * a conservative bound on the compositing kernels' own per-pixel test (forward.cu:331-338 skips such pairs one pixel at a
     * time), so colour, depth and every gradient are bit-identical to binning the whole rectangle, while num_rendered, the
     * tile lists and n_contrib (positions in those lists) shrink: about 40 % of the instances at 960x540 / 200k.
     * !=0: every tile of the splat's rectangle is binned, as duplicateWithKeys does (rasterizer_impl.cu:70-111): num_rendered
     * and the lists are then the reference's, bit for bit.  The same value must be given to every stage of one frame. */
    int keep_all_tiles;
    /* Optional loss epilogue of the forward (train_4DGS.py:218 `Ll1 = l1_loss(image, gt_image)`): with l1_target [3,H,W] set,
     * mom_raster_forward_render also leaves l1_grad [3,H,W] = sign(image - target) / (3 H W) and ADDS sum |image - target| and
     * sum (image - target)^2 to l1_sums[0], [1] -- what mom_l1_loss_acc(3 H W, out_color, target, l1_grad, l1_sums) computes from
     * the stored image, without its launch and its re-read of the image.  All three null: no epilogue. */
    const float* l1_target;
    float* l1_grad;
    float* l1_sums;
    /* !=0: the caller has already cleared the backward's per-Gaussian accumulator record in the geometry scratch
     * (mom_raster_layout().geom_gacc, MOM_GACC_FLOATS floats = 48 bytes per Gaussian) since the last backward read it; mom_raster_backward /
     * _backward_render then skip their own fill command.  (The reference zeroes its ten gradient tensors in
     * RasterizeGaussiansBackwardCUDA, rasterize_points.cu:154-163; the fused training step does this fill on its second stream
     * during the forward.) */
    int accum_cleared;
    /* null, or [tiles of the whole image][2] floats (with l1_target set): every workgroup of the compositing forward then STORES its tile's
     * two sums at entry 2 * (ty * tiles_x + tx) instead of adding them to l1_sums (which may be null): 2040 workgroups adding into one
     * 64-byte line are serialised by the device (about 8 ns each; 6 us at the tail of the launch at 960x540), and nobody may ever read the
     * value.  Whoever wants it adds the entries up -- in a fixed order, so the loss value is reproducible to the bit, which the
     * atomic sums are not.  A tile-row shard writes the entries of its own rows only. */
    float* l1_partials;
    /* null, or one 64-bit word in PINNED HOST memory: the compositing forward then leaves (status_serial << 32) | status there with a
     * system-scope store -- status = this frame's status bits (bit 0: its binning overflowed `capacity`, the image is incomplete), as
     * of the end of the frame's binning.  A host that runs ahead of the device reads the frame's fate from the word once its upper half
     * equals the serial it gave, instead of copying the status back behind an event after every frame (a blit kernel and a marker: 12 us of
     * the stream per frame, 15 us of host time per call of the drop-in's async mode). */
    unsigned long long* status_post;
    uint32_t status_serial;
    /* 0 (= 1), or a factor on the L1 epilogue's gradient: l1_grad = sign(image - target) * ((1 / (3 H W)) * l1_grad_scale).  A camera-batch
     * shard's loss is the mean over the ranks' cameras (train_4DGS.py:189-229), so every rank's gradient image carries 1 / world: with
     * the factor here there is no scaling pass over the gradient image behind the forward.  The sums are not scaled. */
    float l1_grad_scale;
} MomRasterArgs;

/* Scratch sizing (bytes).  The three buffers play the roles of the reference's
 * geomBuffer / binningBuffer / imgBuffer (rasterize_points.cu:72-78) and are
 * handed back to the backward pass unchanged.  Their internal layout is
 * private; mom_raster_layout exposes it for tests. */
size_t mom_raster_geom_bytes(int P);
size_t mom_raster_image_bytes(int W, int H);
size_t mom_raster_binning_bytes(int P, int W, int H, size_t capacity /* max instances */);

/* Forward, stage 1: per-Gaussian projection (preprocessCUDA, forward.cu:156-256),
 * per-tile instance histogram and its scan.  Writes radii[P] (int32), the tile
 * ranges and *num_rendered_dev (uint32, device).  If num_rendered_host is not
 * null it must be host memory (pinned recommended) and receives the same value by
 * an async copy on `stream` (the caller synchronises if it wants to read it; this
 * replaces the blocking cudaMemcpy at rasterizer_impl.cu:282). */
int mom_raster_forward_geometry(const MomRasterArgs* a, void* geom, void* image, int* radii,
                                uint32_t* num_rendered_dev, uint32_t* num_rendered_host, mom_stream_t stream);

/* Forward, stage 2: scatter instances into their tiles, per-tile depth sort
 * (with keep_all_tiles: result identical to the reference's global sort of (tile<<32|depth) keys,
 * rasterizer_impl.cu:70-111,301-318; by default: the same lists less the instances that cannot contribute) and alpha compositing (renderCUDA,
 * forward.cu:261-379).  `capacity` is the instance capacity the binning buffer
 * was sized for; if the true count exceeds it nothing beyond capacity is written, the image
 * is incomplete, and if *status_dev (may be null) is still 0 it receives a->overflow_tag (1
 * when that is 0).  The word is STICKY: the call never clears it, the caller zeroes it (before
 * the first call, and after it has dealt with an overflow), so a host that reads it late cannot
 * miss an overflow and learns which call overflowed first.
 * out_color [3,H,W], out_depth [1,H,W]. */
int mom_raster_forward_render(const MomRasterArgs* a, void* geom, void* binning, size_t capacity, void* image,
                              float* out_color, float* out_depth, uint32_t* status_dev, mom_stream_t stream);

/* Gradients of one backward call (RasterizeGaussiansBackwardCUDA,
 * rasterize_points.cu:154-163,202).  All are fully written by the call (zeros
 * for Gaussians with radius 0), so they need no prior memset. */
typedef struct MomRasterGrads {
    float* dL_dmeans2D;   /* [P,3] (x,y in NDC-scaled pixels; z = 0) */
    float* dL_dcolors;    /* [P,3] */
    float* dL_dopacity;   /* [P,1] */
    float* dL_dmeans3D;   /* [P,3] */
    float* dL_dcov3D;     /* [P,6] */
    float* dL_dsh;        /* [P,M,3] or null when M == 0; with dL_dsh_rest != null: the DC part only, [P,1,3] */
    float* dL_dsh_rest;   /* null, or [P,M-1,3] */
    float* dL_dscales;    /* [P,3] or null when scales absent */
    float* dL_drotations; /* [P,4] or null when rotations absent */
    /* Optional (null: off).  The RAW rotations [P,4] the caller normalised into MomRasterArgs.rotations, for callers whose scales /
     * rotations / opacities are exp / normalize / sigmoid of raw parameters (gaussian_renderer/__init__.py:134-137): dL_dscales,
     * dL_drotations and dL_dopacity are then written THROUGH those activations (w.r.t. the raw values), exactly as
     * mom_activations_backward would make them from the plain gradients, without its launch and its pass over the arrays. */
    const float* act_rotations_raw;
    /* Optional second destinations of dL_dscales / dL_drotations (same values): a caller that reduces those two over ranks in place
     * while another kernel still reads its own copy (the camera-batch shard's deformation backward) gets the copies from the kernel
     * that makes the values, not from two copy launches behind it. */
    float* dL_dscales_copy;
    float* dL_drotations_copy;
} MomRasterGrads;

/* Backward (Rasterizer::backward, rasterizer_impl.cu:343-444): render backward
 * (backward.cu:415-590), cov2D backward (:144-274), preprocess backward (:346-412).
 * dL_dout_color [3,H,W]; dL_dout_depth [1,H,W] or null (treated as zeros).
 * `capacity` is the value given to mom_raster_forward_render, or any smaller value that is still >= the forward's num_rendered
 * (the tile lists sit at the start of the binning buffer whatever it was sized for).  dL_dscales /
 * dL_drotations are written only when scales/rotations are present. */
int mom_raster_backward(const MomRasterArgs* a, const int* radii, void* geom, void* binning, size_t capacity,
                        void* image, const float* dL_dout_color, const float* dL_dout_depth,
                        const MomRasterGrads* g, mom_stream_t stream);

/* The two halves of mom_raster_backward, for callers that must exchange between them (tile-row shard):
 * _render runs the compositing backward over this rank's tile rows and leaves, in the geometry scratch at
 * mom_raster_layout().geom_gacc, one record of MOM_GACC_FLOATS floats per Gaussian (ten used): the sums over the local pixels of the raw terms of
 * dL/d{mean2D.x, mean2D.y (before the Gaussian's conic matrix and the pixel-to-NDC factors are applied), conic.x, conic.y, conic.z
 * (before their -1/2), opacity, r, g, b, depth} and two zeros.  The
 * projection backward is linear in that record, so ranks sum it (an all-reduce of 48 bytes per Gaussian) and
 * then each runs _geometry, which yields identical parameter gradients everywhere. */
int mom_raster_backward_render(const MomRasterArgs* a, void* geom, void* binning, size_t capacity, void* image,
                               const float* dL_dout_color, const float* dL_dout_depth, mom_stream_t stream);
int mom_raster_backward_geometry(const MomRasterArgs* a, const int* radii, void* geom, const MomRasterGrads* grads,
                                 mom_stream_t stream);

/* checkFrustum (rasterizer_impl.cu:54-66): present[i] = p_view.z > 0.2 */
int mom_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     uint8_t* present, mom_stream_t stream);

/* Layout of the private scratch buffers, for tests: byte offsets of the named
 * arrays.  geom: rec[P][12] float = {x, y, depth, tiles_touched(bits) | conic.x,
 * conic.y, conic.z, opacity | r, g, b, radius(bits)}; cov3D[P][6]; clamped[P][4] u8.
 * image: ranges[tiles][2] u32, n_contrib[H*W] u32, final_T[H*W] f32, tile_counts[tiles] u32
 * (the per-tile histogram, bucket cursors and a small header also live here so that the
 * binning buffer can be sized AFTER the instance count is known).
 * binning: point_list[capacity] u32 at offset 0 (all the backward reads of this buffer: a backward may therefore be given
 * any capacity between the true instance count and the forward's), then keys[capacity] u64 (depth_bits<<32|idx, bucketed by tile). */
typedef struct MomRasterLayout {
    size_t geom_rec, geom_cov3D, geom_clamped, geom_gacc;
    size_t img_ranges, img_n_contrib, img_final_T, img_tile_counts;
    size_t bin_keys, bin_point_list;
    size_t img_tile_walked;   /* u32 per tile, valid after mom_raster_forward_render: list entries the tile's workgroup walked before
                               * every pixel was done, in rounds of 256 like forward.cu:305-327 (Q = 256 x their sum) */
} MomRasterLayout;
int mom_raster_layout(int P, int W, int H, size_t capacity, MomRasterLayout* out);

/* distCUDA2 (simple-knn/spatial.cu:15-25): mean squared distance to the 3 nearest
 * neighbours.  scratch must hold mom_knn_scratch_bytes(P).  Fully stream-ordered (the
 * reference does two blocking copies of the bounding box, simple_knn.cu:197,200). */
size_t mom_knn_scratch_bytes(int P);
int mom_knn_mean_dist2(int P, const float* points /* [P,3] */, float* mean_dist2 /* [P] */, void* scratch,
                       mom_stream_t stream);

/* ---- HexPlane feature field --------------------------------------------------------------
 * Replaces HexPlaneField.forward (scene/hexplane.py:160-183: normalize_aabb, 6 bilinear
 * grid_sample(align_corners=True, padding_mode='border') per level, product over planes,
 * concat over levels) and its autograd backward (plane scatter-adds + the gradient wrt the
 * sample positions).  Planes are CHANNEL-LAST: planes[l][p] points to [H][W][32] floats where,
 * for plane p = (a,b) in the order (0,1),(0,2),(0,3),(1,2),(1,3),(2,3) of (x,y,z,t),
 * W = res[l][a] and H = res[l][b].  aabb = the reference's 2x3 `aabb` parameter verbatim
 * (row 0 then row 1; the reference stores xyz_max in row 0, hexplane.py:152-157).  `time` is
 * the raw timestamp used as the 4th grid coordinate (gaussian_renderer/__init__.py:56). */
typedef struct MomHexPlane {
    int levels;            /* 1..4 */
    int channels;          /* must be 32 */
    int res[4][4];         /* per level: resolution along x, y, z, t */
    const float* planes[4][6];
    float* grads[4][6];    /* backward only: same layout as planes, ACCUMULATED into (+=, float atomics) */
    float aabb[6];
} MomHexPlane;
/* feat [P, levels*32] row-major.  times: optional per-point timestamps [P]; null -> `time` for all points
 * (render() uses one timestamp per camera). */
int mom_hexplane_forward(const MomHexPlane* hp, int P, const float* xyz, const float* times, float time,
                         const uint32_t* order, float* feat, mom_stream_t stream);
/* dfeat [P, levels*32]; plane gradients accumulate into hp->grads; dxyz [P,3] (may be null) is ACCUMULATED into.
 * plane_order / plane_inverse ([3][levels][P] each, from mom_hexplane_orders) and scratch (mom_hexplane_backward_scratch_bytes)
 * select the two-pass path (times == null only): pass 1 gathers in `order` and stores each space plane's per-point gradient
 * row at the point's position in that plane's order, pass 2 walks each space plane in its order and turns runs of points of
 * one texel cell into one row of float atomics.  Any of the three null: the generic path (24 atomic rows per point and level). */
size_t mom_hexplane_backward_scratch_bytes(const MomHexPlane* hp, int P);
int mom_hexplane_backward(const MomHexPlane* hp, int P, const float* xyz, const float* times, float time,
                          const uint32_t* order, const float* dfeat, float* dxyz, const uint32_t* plane_order,
                          const uint32_t* plane_inverse, void* scratch, mom_stream_t stream);
/* The same (two-pass path, one timestamp) for a caller that ran mom_deform_field_forward on this field at this `time` just
 * before: field_scratch is that call's scratch, whose head still holds the frame's time lines (the three space-time planes
 * interpolated at `time`), so they are not computed again.  The caller guarantees that neither the planes nor the scratch
 * changed in between (the fused training step: forward and backward of one camera).  Reference: as mom_hexplane_backward. */
int mom_hexplane_backward_lines(const MomHexPlane* hp, int P, const float* xyz, float time, const uint32_t* order,
                                const float* dfeat, float* dxyz, const uint32_t* plane_order, const uint32_t* plane_inverse,
                                void* scratch, const void* field_scratch, mom_stream_t stream);
/* Per-plane processing orders for the two-pass backward: for each space plane (x,y), (x,z), (y,z) and each level the
 * permutation of 0..P-1 that sorts the points by the texel cell of that level they fall into (2-D Morton order over the
 * cells), and its inverse.  Like `order`, they never change a result; they may be reused while the points drift (refresh
 * every few dozen iterations) and must be rebuilt when P changes. */
size_t mom_hexplane_orders_scratch_bytes(int P);
int mom_hexplane_orders(const MomHexPlane* hp, int P, const float* xyz, uint32_t* order /* [3][levels][P] */,
                        uint32_t* inverse /* [3][levels][P] */, void* scratch, mom_stream_t stream);
/* `order` (both calls, may be null = identity): a permutation of 0..P-1 giving the order in which points are
 * processed, e.g. mom_morton_order(xyz).  It never changes a result (float atomics aside); walking the points in a
 * spatially sorted order lets the backward sum consecutive points that hit the same texel in registers and issue one
 * atomic row for the run. */
size_t mom_morton_order_scratch_bytes(int P);
int mom_morton_order(int P, const float* points /* [P,3] */, uint32_t* order /* [P] */, void* scratch, mom_stream_t stream);

/* ---- activations of render() (gaussian_renderer/__init__.py:130-132) and their backward, one launch each ----
 * scales = exp(scales_raw), rots = rots_raw / max(|rots_raw|, 1e-12) (F.normalize), opac = sigmoid(opac_raw) */
int mom_activations_forward(int P, const float* scales_raw, const float* rots_raw, const float* opac_raw, float* scales,
                            float* rots, float* opac, mom_stream_t stream);
/* in: activated values + raw rotations + gradients wrt the activated values; out: gradients wrt the raw values */
int mom_activations_backward(int P, const float* scales, const float* rots_raw, const float* opac, const float* dscales,
                             const float* drots, const float* dopac, float* dscales_raw, float* drots_raw, float* dopac_raw,
                             mom_stream_t stream);

/* ---- fused multi-tensor Adam (torch.optim.Adam, amsgrad=False, weight_decay=0) -----------
 * One launch updates every listed tensor: exp_avg.lerp_(g, 1-b1); exp_avg_sq = b2*v + (1-b2) g*g;
 * p -= lr/bc1 * exp_avg / (sqrt(exp_avg_sq)/sqrt(bc2) + eps), bc = 1 - beta^step computed by the
 * caller in double (as torch does).  Tensors are flat views of n floats in storage order; param,
 * grad and both moments of one tensor must share that order. */
#define MOM_ADAM_MAX_TENSORS 64
typedef struct MomAdamTensor {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    size_t n;
    float lr;
    float bias_correction1;
    float bias_correction2_sqrt;
} MomAdamTensor;
/* skip_if_nonzero (device word, may be null): when it is nonzero at execution time the launch changes nothing.  The fused
 * training step points it at the rasterizer's sticky overflow word (mom_raster_forward_render), so that a step whose binning
 * buffer overflowed -- its image and gradients are truncated -- never reaches the parameters or the moments; the host, which
 * runs ahead of the GPU, finds the flag later and replays that iteration with a larger buffer. */
int mom_adam_step(const MomAdamTensor* tensors, int count, double beta1, double beta2, double eps,
                  const uint32_t* skip_if_nonzero, mom_stream_t stream);

/* ---- L1 loss + PSNR sums + gradient (utils/loss_utils.py:23-24, utils/image_utils.py:17-38) --
 * sums2[0] = sum |img-gt|, sums2[1] = sum (img-gt)^2 over n elements (zeroed by the call);
 * dimg (may be null) = sign(img-gt)/n = d mean|img-gt| / d img. */
int mom_l1_loss(size_t n, const float* img, const float* gt, float* dimg, float* sums2, mom_stream_t stream);
/* the same without zeroing sums2 first: for a caller that keeps them in memory it clears anyway */
int mom_l1_loss_acc(size_t n, const float* img, const float* gt, float* dimg, float* sums2, mom_stream_t stream);

/* ---- Row selection for densify / prune (scene/gaussian_model.py:409-482 _prune_optimizer, cat_tensors_to_optimizer,
 * prune_points; :511-581 densify_and_split, densify_and_clone, prune: `tensor[mask]` once per parameter, per Adam moment and
 * per auxiliary tensor, each with its own nonzero() and host synchronisation) ----
 * plan:  dst_index[i] (device, [n]) = position of row i among the rows with keep[i] != 0, or -1; *count_dev (device) and, if
 *        count_host != null (pinned host memory, copied on the stream), *count_host = the number of kept rows.
 * apply: for each of the `count` tensors, dst[dst_index[i]] = src[i] for the kept rows i; a row is row_bytes bytes (any size;
 *        4-byte words when size and alignment allow).  dst must hold *count rows. */
#define MOM_SELECT_MAX_TENSORS 32
typedef struct {
    const void* src;
    void* dst;
    unsigned row_bytes;
} MomRowSelect;
size_t mom_select_scratch_bytes(int n);
int mom_select_plan(int n, const uint8_t* keep, int* dst_index, int* count_dev, int* count_host, void* scratch,
                    mom_stream_t stream);
int mom_select_apply(int n, const int* dst_index, const MomRowSelect* tensors, int count, mom_stream_t stream);

/* ---- Densification statistics of one iteration (train_4DGS.py:266; scene/gaussian_model.py:713-715
 * add_densification_stats), in place, for the Gaussians with radii[i] > 0:
 *   max_radii2D[i] = max(max_radii2D[i], radii[i]);  xyz_gradient_accum[i] += |viewspace_grad[i, :2]|;  denom[i] += 1.
 * viewspace_grad is [P,3] (dL/d mean2D; the third column is unused), the three accumulators are [P] floats. */
int mom_densify_stats(int P, const int* radii, const float* viewspace_grad, float* max_radii2D, float* xyz_gradient_accum,
                      float* denom, const uint32_t* skip_if_nonzero /* as in mom_adam_step */, mom_stream_t stream);

/* ---- SSIM term of the loss (utils/loss_utils.py:29-92: ssim / _ssim / create_window / gaussian) ----
 * 11x11 Gaussian window = outer product of the 11 taps in window11 (host pointer; the reference's
 * gaussian(11, 1.5)), zero padding 5, C1 = 0.01^2, C2 = 0.03^2.  Images are [C][H][W] (any leading batch
 * dimension folded into C: the reference's conv2d is depthwise and its mean runs over every element).
 * forward:  sum points at MOM_SSIM_SUM_SLOTS doubles on the device (zeroed by the call): sum[0] receives the sum over
 *           all C*H*W elements of the SSIM map, so that ssim = sum[0] / (C*H*W); the other slots hold the partial
 *           sums the blocks spread their atomics over (one shared accumulator serialises in the L2).  dm (device, [3][C][H][W], or null when no gradient is wanted) receives the
 *           map's partial derivatives with respect to the blurred mu1, E[img1^2], E[img1*img2].
 * backward: dimg1 += scale * (scale_dev ? *scale_dev : 1) * d(*sum)/d img1, computed from dm.  For the loss
 *           term lambda * (1 - ssim) pass scale = -lambda / (C*H*W); scale_dev (device scalar, may be null)
 *           lets an autograd caller apply its upstream gradient without reading it back. */
#define MOM_SSIM_SUM_SLOTS 64
int mom_ssim_forward(int C, int H, int W, const float* window11, const float* img1, const float* img2, float* dm,
                     double* sum, mom_stream_t stream);
int mom_ssim_backward(int C, int H, int W, const float* window11, const float* img1, const float* img2, const float* dm,
                      float scale, const float* scale_dev, float* dimg1, mom_stream_t stream);
/* The same on a ROW SLAB of taller images (tile-row shard: a rank holds its own rows plus a halo): the pointers address the
 * slab's first row, H is the slab's height, channels (and the three derivative maps' channels) are chan_stride elements
 * apart (>= H*W; the full image's H*W).  The window is zero-padded at the slab's edges like at an image's, so only map rows
 * at least 5 pixels inside the slab are those of the full image: map rows in [sum_row0, sum_row1) (slab coordinates) are
 * summed, derivative maps are written for rows in [dm_row0, dm_row1) and zeroed elsewhere. */
int mom_ssim_forward_slab(int C, int H, int W, size_t chan_stride, int sum_row0, int sum_row1, int dm_row0, int dm_row1,
                          const float* window11, const float* img1, const float* img2, float* dm, double* sum,
                          mom_stream_t stream);
int mom_ssim_backward_slab(int C, int H, int W, size_t chan_stride, const float* window11, const float* img1, const float* img2,
                           const float* dm, float scale, const float* scale_dev, float* dimg1, mom_stream_t stream);

/* ---- HexPlane regularisers (scene/gaussian_model.py:730-769, scene/regulation.py:22-28) ----
 * value = sum over planes of w_smooth * mean((p[h+2]-2p[h+1]+p[h])^2) + w_l1 * mean|1-p|
 * (second difference along H, the reference's dim -2); if grad != null, grad_scale * d value / d plane is
 * ADDED into it.  Channel-last [H][W][32] planes as above. */
#define MOM_REG_MAX_PLANES 24
typedef struct MomRegPlane {
    const float* plane;
    float* grad;
    int H, W;
    float w_smooth, w_l1, grad_scale;
} MomRegPlane;
int mom_plane_regulation(const MomRegPlane* planes, int count, float* value, mom_stream_t stream);
int mom_plane_regulation_acc(const MomRegPlane* planes, int count, float* value, mom_stream_t stream);   /* value is not zeroed first */
/* _acc with every gradient also multiplied by *upstream, a DEVICE scalar (null: 1): the weight autograd hands down to the
 * regulariser's backward, without reading it back to the host first. */
int mom_plane_regulation_grad(const MomRegPlane* planes, int count, float* value, const float* upstream, mom_stream_t stream);

/* ---- fused deformation MLP (scene/deformation.py:53-65,97-135; W = 64, defor_depth = 0, heads pos/scales/rot) ----
 *   h0 = W0 feat + b0 ;  o_k = W2_k relu(W1_k relu(h0) + b1_k) + b2_k  for k in {pos, scales, rot}
 *   pts = xyz + o_pos + flow_coef * scene_flow ; scales = scaling + o_scales ; rots = rotation + o_rot
 * (flow_coef = delta_scale * frame_num, deformation.py:114).  Weights are nn.Linear tensors ([out,in] row-major);
 * both kernels are persistent and keep all four 64x64 matrices in LDS. */
typedef struct MomDeformMLP {
    const float *W0, *b0;            /* feature_out.0: [64,64], [64] */
    const float *W1[3], *b1[3];      /* {pos,scales,rotations}_deform.1: [64,64], [64] */
    const float *W2[3], *b2[3];      /* {pos,scales,rotations}_deform.3: [3|3|4,64], [3|3|4] */
    float *dW0, *db0, *dW1[3], *db1[3], *dW2[3], *db2[3];   /* backward only: ACCUMULATED into (+=) */
} MomDeformMLP;
/* a0_save (may be null for inference): [P,64], receives relu(h0) for the backward pass */
int mom_deform_forward(const MomDeformMLP* w, int P, const float* feat /* [P,64] */, const float* xyz, const float* scaling,
                       const float* rotation, const float* scene_flow, float flow_coef, float* pts, float* scales, float* rots,
                       float* a0_save, mom_stream_t stream);
/* The same, also leaving the activated values the rasterizer takes (gaussian_renderer/__init__.py:96-99: scaling_activation = exp,
 * rotation_activation = normalize, opacity_activation = sigmoid of the UNdeformed opacity) -- what mom_activations_forward would
 * compute from scales / rots / opacity_raw, without its launch.  Any of scales_act, rots_act, opacity_act may be null;
 * opacity_raw is required exactly when opacity_act is given. */
int mom_deform_forward_activated(const MomDeformMLP* w, int P, const float* feat, const float* xyz, const float* scaling,
                                 const float* rotation, const float* scene_flow, float flow_coef, float* pts, float* scales,
                                 float* rots, float* a0_save, const float* opacity_raw, float* scales_act, float* rots_act,
                                 float* opacity_act, mom_stream_t stream);
/* d{pts,scales,rots}: gradients of the three outputs; writes dfeat [P,64]; weight/bias gradients accumulate into w->d*.
 * (The identity paths d xyz += dpts etc. are the caller's.) */
size_t mom_deform_backward_scratch_bytes(int P);   /* the larger of 4 x [P,64] floats (the two-kernel form's pre-activation gradients) and the
                                                       one-kernel form's per-workgroup partial sums (17.8 MB) */
int mom_deform_backward(const MomDeformMLP* w, int P, const float* feat, const float* a0, const float* dpts,
                        const float* dscales, const float* drots, float* dfeat, void* scratch, mom_stream_t stream);
/* The same with a second stream at the callee's disposal.  Default form (one kernel on the bf16 matrix pipe, csrc/deform_bwd_b3.hip:
 * the pre-activation gradients never leave the CU): dfeat is complete on `stream`; the weight / bias gradients are complete on
 * `dw_stream` (the sum over the workgroups' partial sums runs there, ordered behind the kernel on `stream`).
 * Two-kernel form (MOM_MLP_BWD=split in the environment): dfeat and the thin output layers' gradients are complete on `stream`,
 * the 64x64 layers' weight / bias gradients on `dw_stream`, which the call orders behind the part on `stream` that produces their
 * input.  Either way the caller joins `dw_stream` before reading the gradients or reusing `scratch`.  dw_stream == stream:
 * identical to mom_deform_backward. */
int mom_deform_backward_split(const MomDeformMLP* w, int P, const float* feat, const float* a0, const float* dpts,
                              const float* dscales, const float* drots, float* dfeat, void* scratch, mom_stream_t stream,
                              mom_stream_t dw_stream);

/* ---- deformation field in one pass (the render() case: ONE timestamp for every point) ----
 * deform_network.forward = HexPlaneField lookup + trunk + heads (scene/deformation.py:97-153, scene/hexplane.py:160-183) as a
 * single persistent kernel: the 64 features of a tile of 32 Gaussians go from the texel gathers through LDS straight into the
 * matrix-core operand, feat[P,64] crosses HBM only if the caller asks for a copy (feat_save, for the backward's weight
 * gradients).  The three space-time planes are first collapsed to per-frame lines (one tiny launch into `scratch`,
 * mom_deform_field_scratch_bytes).  Same outputs and activated copies as mom_hexplane_forward + mom_deform_forward_activated;
 * `order` (optional) is the processing order as in mom_hexplane_forward.  Needs levels == 2, channels == 32 and resolutions
 * <= 1024 (mom_deform_field_supported); other shapes take the two separate calls. */
int mom_deform_field_supported(const MomHexPlane* hp);
size_t mom_deform_field_scratch_bytes(const MomHexPlane* hp, int P /* 0 if every call keeps a feature copy (feat_save) */);
int mom_deform_field_forward(const MomHexPlane* hp, const MomDeformMLP* w, int P, const float* xyz, float time,
                             const uint32_t* order, const float* scaling, const float* rotation, const float* scene_flow,
                             float flow_coef, float* pts, float* scales, float* rots, float* feat_save, float* a0_save,
                             const float* opacity_raw, float* scales_act, float* rots_act, float* opacity_act, void* scratch,
                             mom_stream_t stream);

/* ---- rendered image -> 8-bit interleaved RGB (render_4DGS.py:64 torchvision.utils.save_image: x * 255 + 0.5, clamp, truncate;
 * CHW -> HWC) in one pass, so that a frame can leave the device as the bytes a PNG encoder takes.  img [C,H,W] floats, out [H,W,C]
 * bytes; C <= 4. */
int mom_image_to_rgb8(int C, int H, int W, const float* img, uint8_t* out, mom_stream_t stream);

const char* mom_version(void);

/* Per-kernel HIP-event timing (bench.py's live roofline figure).  Slots: see mom_profile_name(0..15).
 * While a slot is enabled launches of that kernel are bracketed by two hipEventRecord on the launch stream: every launch with
 * on = 1, every on-th launch with on > 1 (an event pair costs the stream ~13 us of bubbles; a sampled mean costs 1/on of that).
 * mom_profile_read synchronises those events and returns the accumulated time and the number of launches TIMED. */
#define MOM_PROF_SLOTS 16
int mom_profile_enable(int slot, int on);
int mom_profile_read(int slot, double* total_ms, long long* count, int reset);
const char* mom_profile_name(int slot);

/* Stream ordering for a host mirror that runs parts of a step on a second stream (no counterpart in the reference, whose loop is
 * single-stream; torch's own Stream / Event objects do the same at 8-10 us of host time per call).
 * mom_stream_wait_stream: everything enqueued on `waiter` after this call runs after everything enqueued on `signaler` before it.
 * mom_stream_mark / mom_stream_wait_mark: a ring of MOM_STREAM_MARKS reusable marks per device -- record the current tail of a
 * stream under a slot number, make another stream wait for it later (a wait refers to the slot's most recent record at the time of
 * the call; waiting for a slot that was never recorded is MOM_EINVAL).
 * These calls order STREAMS OF ONE DEVICE and nothing else: their events carry no system-scope release (hipEventDisableSystemFence --
 * 3.4 instead of 6.2 us of the recording stream per record; MOM_EVENT_SYSTEM_FENCE=1 in the environment restores the default), so they
 * publish nothing to the host or to another device. */
#define MOM_STREAM_MARKS 64
int mom_stream_wait_stream(mom_stream_t waiter, mom_stream_t signaler);
int mom_stream_mark(int slot, mom_stream_t stream);
int mom_stream_wait_mark(mom_stream_t stream, int slot);
/* bytes of zeros at ptr, stream-ordered (the gradient buckets a step clears on its second stream) */
int mom_zero_async(void* ptr, size_t bytes, mom_stream_t stream);

/* ---- collectives of a multi-GPU step, over librccl (RCCL on xGMI), one process per GPU.
 * No counterpart in the reference, which is single-GPU: its batch axis (train_4DGS.py:172-229, `batch_size` cameras rendered one after
 * the other, loss averaged) is what the camera-batch shard spreads over ranks, and these are the exchanges of that shard and of the
 * tile-row shard.  They exist as C entry points because a torch.distributed call costs the rank's host 40-50 us
 * and the host paces a rank; here a collective is one ncclXxx call on a raw stream handle (~2 us), ordered against the compute
 * streams with mom_stream_mark / mom_stream_wait_mark.
 *   mom_comm_available     1 if librccl could be resolved (dlopen at first use; a process that has imported torch gets torch's copy)
 *   mom_comm_unique_id     rank 0 makes the 128-byte id and hands it to the other ranks through the launcher's own rendezvous
 *   mom_comm_create        ncclCommInitRank on the calling thread's CURRENT device; collective over all `world` ranks
 *   mom_comm_all_reduce    in place over `count` elements
 *   mom_comm_all_gather    in place: buf holds world x count_per_rank elements, this rank's slab at rank x count_per_rank going in
 *   mom_comm_reduce_scatter in place: buf holds world x count_per_rank elements; this rank's reduced block lands at rank x count_per_rank,
 *                          the rest of buf is left in an unspecified state
 *   mom_comm_group_start / _end   ncclGroupStart / ncclGroupEnd: the calls in between are submitted as one launch
 * dtype: MOM_COMM_F32 | MOM_COMM_I32 (4-byte elements); op: MOM_COMM_SUM | MOM_COMM_MAX.  Every rank must issue the same sequence. */
typedef struct MomComm MomComm;
#define MOM_COMM_F32 0
#define MOM_COMM_I32 1
#define MOM_COMM_SUM 0
#define MOM_COMM_MAX 1
int mom_comm_available(void);
const char* mom_comm_last_error(void);
int mom_comm_unique_id(void* id128);
int mom_comm_create(MomComm** out, const void* id128, int world, int rank);
int mom_comm_destroy(MomComm* comm);
int mom_comm_abort(MomComm* comm);
int mom_comm_world(const MomComm* comm);
int mom_comm_rank(const MomComm* comm);
int mom_comm_group_start(void);
int mom_comm_group_end(void);
int mom_comm_all_reduce(MomComm* comm, void* buf, size_t count, int dtype, int op, mom_stream_t stream);
int mom_comm_all_gather(MomComm* comm, void* buf, size_t count_per_rank, int dtype, mom_stream_t stream);
int mom_comm_reduce_scatter(MomComm* comm, void* buf, size_t count_per_rank, int dtype, int op, mom_stream_t stream);

/* Self test of the wave64 DPP reduction used by the render backward:
 * out[w] = sum(in[64w .. 64w+63]). */
int mom_selftest_wave_sum(const float* in, float* out, int waves, mom_stream_t stream);
/* Self test of the render backward's packed reduction (row-level DPP reduce-scatter of nvals = 9 or 10 values per lane, then the
 * 4 x 4 row transposition over four splats): in [waves][4][nvals][64] -> out [waves][4][nvals] = the sums over the 64 lanes. */
int mom_selftest_row_reduce(const float* in, float* out, int waves, int nvals, mom_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MOM4D_H_INCLUDED */
