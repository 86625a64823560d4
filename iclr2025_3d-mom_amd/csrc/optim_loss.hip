// Fused multi-tensor Adam, L1 loss (+PSNR sums, +gradient) and the HexPlane regularisers, gfx950.
//
// Replaces torch.optim.Adam.step over 8 parameter groups (reference scene/gaussian_model.py:209,
// train_4DGS.py:295-297: ~44 tensors -> one launch), l1_loss + psnr (utils/loss_utils.py:23-24,
// utils/image_utils.py:17-38: one pass that also emits dL/dimage) and compute_regulation
// (scene/gaussian_model.py:730-769, scene/regulation.py:22-28: one launch for 12 planes, value and
// gradient).  All are streaming passes: 16-byte accesses where the layout allows, grid-stride,
// one atomic per workgroup for reductions.
#include "mom_common.h"
#include <stdlib.h>

namespace {

// ---------------------------------------------------------------- Adam
struct AdamArgs {
    MomAdamTensor t[MOM_ADAM_MAX_TENSORS];
    unsigned block_start[MOM_ADAM_MAX_TENSORS + 1];
    int count;
    float beta1, beta2, eps, w1, w2;  // w = 1 - beta evaluated in double on the host, as torch does
    const uint32_t* skip;             // device word, may be null: nonzero -> the whole launch is a no-op
};
constexpr int kAdamPerBlock = 256 * 8;

__global__ void __launch_bounds__(256) adam_kernel(AdamArgs a)
{
    if (a.skip && *a.skip) return;    // the step's binning overflowed: its gradients are truncated, leave the state alone
    // find the tensor this workgroup works on (count <= 64: linear scan on the scalar unit)
    int ti = 0;
    while (ti + 1 < a.count && blockIdx.x >= a.block_start[ti + 1]) ti++;
    const MomAdamTensor T = a.t[ti];
    const size_t base = (size_t)(blockIdx.x - a.block_start[ti]) * kAdamPerBlock;
    const float w1 = a.w1, w2 = a.w2;
    const float step_size = T.lr / T.bias_correction1;
    const float inv_bc2_sqrt = 1.f / T.bias_correction2_sqrt;
#ifndef MOM_ADAM_SCALAR
    // 16 bytes per lane where the four arrays allow it (the parameters and the moments are allocations of their own; a gradient may be
    // a view into a bucket at any 4-byte offset) and the workgroup's eight rows of 256 are all inside the tensor: the same
    // arithmetic per element, a quarter of the memory instructions -- what matters when the launch has an eighth of the chip (the
    // early launch beside the MLP backward) and the stream is bound by how much each CU keeps in flight
    const bool vec = (((uintptr_t)T.param | (uintptr_t)T.grad | (uintptr_t)T.exp_avg | (uintptr_t)T.exp_avg_sq) & 15) == 0 &&
                     base + kAdamPerBlock <= T.n;
    if (vec) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const size_t i4 = base / 4 + (size_t)k * 256 + threadIdx.x;
            const float4 g4 = reinterpret_cast<const float4*>(T.grad)[i4];
            float4 m4 = reinterpret_cast<float4*>(T.exp_avg)[i4], v4 = reinterpret_cast<float4*>(T.exp_avg_sq)[i4];
            float4 p4 = reinterpret_cast<float4*>(T.param)[i4];
            float* gp = const_cast<float*>(reinterpret_cast<const float*>(&g4));
            float *mp = reinterpret_cast<float*>(&m4), *vp = reinterpret_cast<float*>(&v4), *pp = reinterpret_cast<float*>(&p4);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float g = gp[e];
                float m = mp[e], v = vp[e];
                m = m + (g - m) * w1;
                v = v * a.beta2 + w2 * g * g;
                const float denom = sqrtf(v) * inv_bc2_sqrt + a.eps;
                pp[e] = pp[e] - step_size * (m / denom);
                mp[e] = m;
                vp[e] = v;
            }
            reinterpret_cast<float4*>(T.param)[i4] = p4;
            reinterpret_cast<float4*>(T.exp_avg)[i4] = m4;
            reinterpret_cast<float4*>(T.exp_avg_sq)[i4] = v4;
        }
        return;
    }
#endif
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const size_t i = base + (size_t)k * 256 + threadIdx.x;
        if (i < T.n) {
            const float g = T.grad[i];
            float m = T.exp_avg[i], v = T.exp_avg_sq[i];
            m = m + (g - m) * w1;                       // exp_avg.lerp_(grad, 1-beta1)
            v = v * a.beta2 + w2 * g * g;               // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1-beta2)
            const float denom = sqrtf(v) * inv_bc2_sqrt + a.eps;
            T.param[i] = T.param[i] - step_size * (m / denom);
            T.exp_avg[i] = m;
            T.exp_avg_sq[i] = v;
        }
    }
}

// ---------------------------------------------------------------- L1 (+ squared error sums + gradient)
__device__ __forceinline__ float block_sum(float v, float* s)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0)
        for (int w = 0; w < (int)(blockDim.x >> 6); w++) t += s[w];
    __syncthreads();
    return t;  // valid in thread 0
}

__global__ void __launch_bounds__(256)
l1_kernel(size_t n, const float* __restrict__ img, const float* __restrict__ gt, float* __restrict__ dimg, float inv_n,
          float* __restrict__ sums /* [0]=sum|d| [1]=sum d^2 */)
{
    __shared__ float s[4];
    float a1 = 0.f, a2 = 0.f;
    auto one = [&](float x, float y) {
        const float d = x - y;
        a1 += fabsf(d);
        a2 += d * d;
        return d > 0.f ? inv_n : (d < 0.f ? -inv_n : 0.f);              // sign(d)/n, torch.abs' subgradient
    };
    // 16 bytes per lane when the three pointers allow it; the last n % 4 elements (and everything, otherwise) go one by one
    const bool vec = ((reinterpret_cast<uintptr_t>(img) | reinterpret_cast<uintptr_t>(gt) | reinterpret_cast<uintptr_t>(dimg)) & 15) == 0;
    const size_t n4 = vec ? n / 4 : 0;
    const float4* __restrict__ img4 = reinterpret_cast<const float4*>(img);
    const float4* __restrict__ gt4 = reinterpret_cast<const float4*>(gt);
    float4* __restrict__ dimg4 = reinterpret_cast<float4*>(dimg);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 x = img4[i], y = gt4[i];
        const float4 g = make_float4(one(x.x, y.x), one(x.y, y.y), one(x.z, y.z), one(x.w, y.w));
        if (dimg) dimg4[i] = g;
    }
    for (size_t i = 4 * n4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float g = one(img[i], gt[i]);
        if (dimg) dimg[i] = g;
    }
    const float t1 = block_sum(a1, s);
    const float t2 = block_sum(a2, s);
    if (threadIdx.x == 0) {
        atomicAdd(&sums[0], t1);
        atomicAdd(&sums[1], t2);
    }
}

// ---------------------------------------------------------------- plane regularisers
// One workgroup strip per (plane, column block): second difference along H for every (w, c), plus the
// |1 - p| term for space-time planes.  Planes are channel-last [H][W][32]: element (h, w, c) at (h*W + w)*32 + c,
// so a fixed h is one contiguous row of W*32 floats and all accesses are coalesced.
struct RegArgs {
    MomRegPlane p[MOM_REG_MAX_PLANES];
    unsigned block_start[MOM_REG_MAX_PLANES + 1];
    int count;
    const float* upstream;     // device scalar multiplied into every gradient (null: 1): the loss weight that autograd hands down
};

constexpr int kRegRows = 16;   // rows of H per workgroup (plus a 2-row halo on each side)

__global__ void __launch_bounds__(256) plane_reg_kernel(RegArgs a, float* __restrict__ out /* [0] = value */)
{
    const float up = a.upstream ? *a.upstream : 1.f;
    __shared__ float s[4];
    int pi = 0;
    while (pi + 1 < a.count && blockIdx.x >= a.block_start[pi + 1]) pi++;
    const MomRegPlane P = a.p[pi];
    const int row = P.W * 32;  // floats per h
    const int cblocks = (row + 255) / 256;
    const int local = blockIdx.x - a.block_start[pi];
    const int col = (local % cblocks) * 256 + threadIdx.x;
    const int h_begin = (local / cblocks) * kRegRows, h_end = min(P.H, h_begin + kRegRows);
    float val = 0.f;
    if (col < row) {
        const float* __restrict__ t = P.plane;
        float* __restrict__ g = P.grad;
        const int H = P.H;
        // smoothness: mean over (c, H-2, W) of s_h^2 with s_h = t[h+2] - 2 t[h+1] + t[h], weight w_smooth;
        // d/dt[h] sum s^2 = 2 (s_{h-2} - 2 s_{h-1} + s_h).  This workgroup owns rows [h_begin, h_end): it adds the
        // value terms s_h for h in its range and writes the gradient of its rows (needs s_{h-2} .. s_h => rows h-2 .. h+2).
        const float cs = (H > 2 && P.w_smooth != 0.f) ? P.w_smooth / ((float)(H - 2) * (float)row) : 0.f;
        const float cl = P.w_l1 != 0.f ? P.w_l1 / ((float)H * (float)row) : 0.f;
        // all kRegRows + 4 rows of the column and the kRegRows gradient values are loaded up front: 36 independent loads in
        // flight per thread instead of a chain of dependent ones (the strip is latency bound: 35 MB per launch)
        float tv[kRegRows + 4], gv[kRegRows];
#pragma unroll
        for (int k = 0; k < kRegRows + 4; k++) {
            const int h = h_begin - 2 + k;
            tv[k] = (h >= 0 && h < H) ? t[(size_t)h * row + col] : 0.f;
        }
        if (g) {
#pragma unroll
            for (int k = 0; k < kRegRows; k++) gv[k] = (h_begin + k < h_end) ? g[(size_t)(h_begin + k) * row + col] : 0.f;
        }
        auto S = [&](int h, float a0, float a1, float a2) { return (h >= 0 && h + 2 < H) ? (a2 - 2.f * a1 + a0) : 0.f; };
#pragma unroll
        for (int k = 0; k < kRegRows; k++) {
            const int h = h_begin + k;
            if (h < h_end) {
                // tv[k] = t[h-2], tv[k+1] = t[h-1], tv[k+2] = t[h], tv[k+3] = t[h+1], tv[k+4] = t[h+2]
                const float sm2 = S(h - 2, tv[k], tv[k + 1], tv[k + 2]), sm1 = S(h - 1, tv[k + 1], tv[k + 2], tv[k + 3]);
                const float sh = S(h, tv[k + 2], tv[k + 3], tv[k + 4]);
                val += cs * sh * sh;
                float gr = 2.f * cs * (sm2 - 2.f * sm1 + sh);
                if (cl != 0.f) {
                    const float d = 1.f - tv[k + 2];
                    val += cl * fabsf(d);
                    gr += d > 0.f ? -cl : (d < 0.f ? cl : 0.f);
                }
                if (g) g[(size_t)h * row + col] = gv[k] + gr * (P.grad_scale * up);
            }
        }
    }
    const float tot = block_sum(val, s);
    if (threadIdx.x == 0 && tot != 0.f) atomicAdd(out, tot);
}

}  // namespace
namespace {
// ---------------------------------------------------------------- activations (exp / normalize / sigmoid)
__global__ void __launch_bounds__(256) act_fwd_kernel(int P, const float* __restrict__ sr, const float* __restrict__ rr,
                                                     const float* __restrict__ orr, float* __restrict__ s, float* __restrict__ r,
                                                     float* __restrict__ o)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
#pragma unroll
    for (int k = 0; k < 3; k++) s[3 * i + k] = expf(sr[3 * i + k]);
    const float4 q = *reinterpret_cast<const float4*>(rr + 4 * i);
    const float n = mom_quat_norm(q.x, q.y, q.z, q.w);
    *reinterpret_cast<float4*>(r + 4 * i) = make_float4(q.x / n, q.y / n, q.z / n, q.w / n);
    o[i] = mom_sigmoid(orr[i]);
}
__global__ void __launch_bounds__(256)
act_bwd_kernel(int P, const float* __restrict__ s, const float* __restrict__ rr, const float* __restrict__ o,
               const float* __restrict__ ds, const float* __restrict__ dr, const float* __restrict__ dop, float* __restrict__ dsr,
               float* __restrict__ drr, float* __restrict__ dor)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
#pragma unroll
    for (int k = 0; k < 3; k++) dsr[3 * i + k] = ds[3 * i + k] * s[3 * i + k];
    const float4 q = *reinterpret_cast<const float4*>(rr + 4 * i);
    const float4 g = *reinterpret_cast<const float4*>(dr + 4 * i);
    const float nrm = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    const float n = fmaxf(nrm, 1e-12f);
    const float4 u = make_float4(q.x / n, q.y / n, q.z / n, q.w / n);
    // d/dq (q/|q|) : (g - u (u.g)) / |q|   (zero through the clamp when |q| < eps, as ATen does)
    const float dot = (nrm >= 1e-12f) ? (u.x * g.x + u.y * g.y + u.z * g.z + u.w * g.w) : 0.f;
    *reinterpret_cast<float4*>(drr + 4 * i) = make_float4((g.x - u.x * dot) / n, (g.y - u.y * dot) / n, (g.z - u.z * dot) / n, (g.w - u.w * dot) / n);
    const float y = o[i];
    dor[i] = dop[i] * ((1.0f - y) * y);
}

}  // namespace

extern "C" int mom_adam_step(const MomAdamTensor* tensors, int count, double beta1, double beta2, double eps,
                             const uint32_t* skip_if_nonzero, mom_stream_t stream)
{
    if (count < 0 || (count > 0 && !tensors)) return MOM_EINVAL;
    int done = 0;
    while (done < count) {
        AdamArgs a;
        a.beta1 = (float)beta1; a.beta2 = (float)beta2; a.eps = (float)eps;
        a.w1 = (float)(1.0 - beta1); a.w2 = (float)(1.0 - beta2);
        a.skip = skip_if_nonzero;
        int n = count - done;
        if (n > MOM_ADAM_MAX_TENSORS) n = MOM_ADAM_MAX_TENSORS;
        unsigned blocks = 0;
        for (int i = 0; i < n; i++) {
            a.t[i] = tensors[done + i];
            if (a.t[i].n && (!a.t[i].param || !a.t[i].grad || !a.t[i].exp_avg || !a.t[i].exp_avg_sq)) return MOM_EINVAL;
            a.block_start[i] = blocks;
            blocks += (unsigned)((a.t[i].n + kAdamPerBlock - 1) / kAdamPerBlock);
        }
        a.block_start[n] = blocks;
        a.count = n;
        if (blocks) {
            MomProfScope ps(MOM_P_ADAM, (hipStream_t)stream);
            hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
            if (hipGetLastError() != hipSuccess) return MOM_ELAUNCH;
        }
        done += n;
    }
    return MOM_OK;
}

extern "C" int mom_l1_loss(size_t n, const float* img, const float* gt, float* dimg, float* sums2, mom_stream_t stream)
{
    if (!img || !gt || !sums2) return MOM_EINVAL;
    if (hipMemsetAsync(sums2, 0, 8, (hipStream_t)stream) != hipSuccess) return MOM_ELAUNCH;
    return mom_l1_loss_acc(n, img, gt, dimg, sums2, stream);
}

extern "C" int mom_l1_loss_acc(size_t n, const float* img, const float* gt, float* dimg, float* sums2, mom_stream_t stream)
{
    if (!img || !gt || !sums2) return MOM_EINVAL;
    if (n == 0) return MOM_OK;
    // every block ends with two atomics on the same two floats, and same-address atomics serialise in the L2: with 760
    // blocks that tail cost more than streaming the images.  A few hundred fat blocks keep every CU busy and the tail short.
    static size_t cap = 0;
    if (!cap) {
        const char* e = getenv("MOM_L1_BLOCKS");
        cap = e ? (size_t)atoi(e) : 256;
        if (cap < 1) cap = 1;
    }
    size_t blocks = (n + 256 * 4 - 1) / (256 * 4);
    if (blocks > cap) blocks = cap;
    MomProfScope ps(MOM_P_L1, (hipStream_t)stream);
    hipLaunchKernelGGL(l1_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n, img, gt, dimg, 1.0f / (float)n, sums2);
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}

// Per-iteration densification statistics (train_4DGS.py:266, scene/gaussian_model.py:713-715) in one pass, in place:
// for the Gaussians the frame saw (radius > 0) the running maximum radius, the accumulated norm of the screen-space
// gradient and its count.  The reference does this with boolean-mask indexing (a nonzero() and its host sync per line).
namespace {
__global__ void __launch_bounds__(256)
densify_stats_kernel(int P, const int* __restrict__ radii, const float* __restrict__ vsp_grad, float* __restrict__ max_radii2D,
                     float* __restrict__ grad_accum, float* __restrict__ denom, const uint32_t* __restrict__ skip)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P || (skip && *skip)) return;
    const int r = radii[i];
    if (r <= 0) return;
    max_radii2D[i] = fmaxf(max_radii2D[i], (float)r);
    const float gx = vsp_grad[3 * i], gy = vsp_grad[3 * i + 1];
    grad_accum[i] += sqrtf(gx * gx + gy * gy);
    denom[i] += 1.0f;
}
}  // namespace

extern "C" int mom_densify_stats(int P, const int* radii, const float* viewspace_grad, float* max_radii2D, float* xyz_gradient_accum,
                                 float* denom, const uint32_t* skip_if_nonzero, mom_stream_t stream)
{
    if (P < 0) return MOM_EINVAL;
    if (P == 0) return MOM_OK;
    if (!radii || !viewspace_grad || !max_radii2D || !xyz_gradient_accum || !denom) return MOM_EINVAL;
    hipLaunchKernelGGL(densify_stats_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, radii, viewspace_grad,
                       max_radii2D, xyz_gradient_accum, denom, skip_if_nonzero);
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}

extern "C" int mom_plane_regulation(const MomRegPlane* planes, int count, float* value, mom_stream_t stream)
{
    if (!value) return MOM_EINVAL;
    if (hipMemsetAsync(value, 0, 4, (hipStream_t)stream) != hipSuccess) return MOM_ELAUNCH;
    return mom_plane_regulation_acc(planes, count, value, stream);
}

extern "C" int mom_plane_regulation_acc(const MomRegPlane* planes, int count, float* value, mom_stream_t stream)
{
    return mom_plane_regulation_grad(planes, count, value, nullptr, stream);
}

extern "C" int mom_plane_regulation_grad(const MomRegPlane* planes, int count, float* value, const float* upstream,
                                         mom_stream_t stream)
{
    if (count < 0 || count > MOM_REG_MAX_PLANES || !value || (count && !planes)) return MOM_EINVAL;
    RegArgs a;
    unsigned blocks = 0;
    for (int i = 0; i < count; i++) {
        a.p[i] = planes[i];
        if (!a.p[i].plane || a.p[i].H < 1 || a.p[i].W < 1) return MOM_EINVAL;
        a.block_start[i] = blocks;
        blocks += (unsigned)(((a.p[i].W * 32 + 255) / 256) * ((a.p[i].H + kRegRows - 1) / kRegRows));
    }
    a.block_start[count] = blocks;
    a.count = count;
    a.upstream = upstream;
    MomProfScope ps(MOM_P_REG, (hipStream_t)stream);
    if (blocks) hipLaunchKernelGGL(plane_reg_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, value);
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}

extern "C" int mom_activations_forward(int P, const float* scales_raw, const float* rots_raw, const float* opac_raw, float* scales,
                                       float* rots, float* opac, mom_stream_t stream)
{
    if (P < 0) return MOM_EINVAL;
    if (P == 0) return MOM_OK;
    if (!scales_raw || !rots_raw || !opac_raw || !scales || !rots || !opac) return MOM_EINVAL;
    hipLaunchKernelGGL(act_fwd_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, scales_raw, rots_raw, opac_raw, scales,
                       rots, opac);
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}
extern "C" int mom_activations_backward(int P, const float* scales, const float* rots_raw, const float* opac, const float* dscales,
                                        const float* drots, const float* dopac, float* dscales_raw, float* drots_raw,
                                        float* dopac_raw, mom_stream_t stream)
{
    if (P < 0) return MOM_EINVAL;
    if (P == 0) return MOM_OK;
    if (!scales || !rots_raw || !opac || !dscales || !drots || !dopac || !dscales_raw || !drots_raw || !dopac_raw) return MOM_EINVAL;
    hipLaunchKernelGGL(act_bwd_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, scales, rots_raw, opac, dscales, drots,
                       dopac, dscales_raw, drots_raw, dopac_raw);
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}

// ---------------------------------------------------------------- image -> 8-bit RGB (render_4DGS.py:64 save_image, :65 to8b)
// out[y][x][c] = (uint8) clamp(img[c][y][x] * 255 + 0.5, 0, 255): torchvision.utils.save_image's quantisation of one image, in one
// pass that also does its CHW -> HWC permute, so that the frame leaves the device as the bytes the PNG encoder wants.
namespace {
__global__ void __launch_bounds__(256) image_to_rgb8_kernel(int C, int HW, const float* __restrict__ img, uint8_t* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    for (int c = 0; c < C; c++) {
        const float v = fminf(fmaxf(img[(size_t)c * HW + i] * 255.0f + 0.5f, 0.f), 255.f);
        out[(size_t)i * C + c] = (uint8_t)v;
    }
}
}  // namespace

extern "C" int mom_image_to_rgb8(int C, int H, int W, const float* img, uint8_t* out, mom_stream_t stream)
{
    if (C < 1 || C > 4 || H < 0 || W < 0) return MOM_EINVAL;
    if (H == 0 || W == 0) return MOM_OK;
    if (!img || !out) return MOM_EINVAL;
    const int HW = H * W;
    hipLaunchKernelGGL(image_to_rgb8_kernel, dim3((HW + 255) / 256), dim3(256), 0, (hipStream_t)stream, C, HW, img, out);
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}
