// SSIM loss term, forward and backward, gfx950.
//
// Replaces utils/loss_utils.py:29-92 of the reference (ssim / _ssim / create_window / gaussian): an 11x11 Gaussian
// window (sigma 1.5, the outer product of an 11-tap 1-D window), zero padding 5, C1 = 0.01^2, C2 = 0.03^2, mean over
// every element.  The reference runs five depthwise conv2d forward and their backward; here one kernel per direction.
//
// Forward: a 16x16 pixel tile (one channel) stages its 26x26 neighbourhood of both images in LDS, blurs the five
// moments x, y, x^2, y^2, xy separably (rows, then columns), forms the SSIM map and its three partial derivatives with
// respect to the blurred moments that depend on the rendered image x:
//     A = mu1^2 + mu2^2 + C1,  B = s1 + s2 + C2,  Cn = 2 mu1 mu2 + C1,  D = 2 s12 + C2,  map = Cn D / (A B)
//     s1 = E[x^2] - mu1^2, s12 = E[xy] - mu1 mu2
//     d map / d mu1    = 2 mu2 (D - Cn) / (A B) - 2 mu1 Cn D (B - A) / (A B)^2
//     d map / d E[x^2] = - Cn D / (A B^2)
//     d map / d E[xy]  = 2 Cn / (A B)
// Backward: the blur with a zero-padded symmetric window is its own transpose, so
//     d sum(map) / d x(q) = blur(d map/d mu1)(q) + 2 x(q) blur(d map/d E[x^2])(q) + y(q) blur(d map/d E[xy])(q).
#include "mom_common.h"

namespace {

constexpr int kT = 16;             // tile edge
constexpr int kR = 5;              // window radius
constexpr int kE = kT + 2 * kR;    // staged edge (26)
constexpr int kES = kE + 1;        // LDS row stride
constexpr int kSsimSlots = MOM_SSIM_SUM_SLOTS;   // doubles behind `sum`: [0] the total, the rest partial sums

struct Window { float w[2 * kR + 1]; };

__device__ __forceinline__ float block_sum_256(float v, float* s_part)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) s_part[wave] = v;
    __syncthreads();
    return s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

__global__ void __launch_bounds__(256)
ssim_fwd_kernel(Window win, int H, int W, size_t chan_stride, int sum_row0, int sum_row1, int dm_row0, int dm_row1,
                const float* __restrict__ img1, const float* __restrict__ img2, float* __restrict__ dm, double* __restrict__ sum)
{
    // The images are H rows of W pixels per channel, channels chan_stride elements apart (a row slab of a taller image has
    // chan_stride > H W).  Map rows in [sum_row0, sum_row1) count toward the sum; derivative maps are written for rows in
    // [dm_row0, dm_row1) and are zero elsewhere.
    __shared__ float s_x[kE][kES], s_y[kE][kES];
    __shared__ float s_h[5][kE][kT + 1];
    __shared__ float s_part[4];
    const size_t base = (size_t)blockIdx.z * chan_stride;
    const int x0 = blockIdx.x * kT - kR, y0 = blockIdx.y * kT - kR;
    for (int i = threadIdx.x; i < kE * kE; i += 256) {
        const int ly = i / kE, lx = i % kE, gx = x0 + lx, gy = y0 + ly;
        const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;
        s_x[ly][lx] = in ? img1[base + (size_t)gy * W + gx] : 0.f;       // zero padding, as conv2d(padding=5)
        s_y[ly][lx] = in ? img2[base + (size_t)gy * W + gx] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kE * kT; i += 256) {                      // rows
        const int r = i / kT, c = i % kT;
        float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int k = 0; k <= 2 * kR; k++) {
            const float a = s_x[r][c + k], b = s_y[r][c + k], w = win.w[k];
            m1 += w * a;
            m2 += w * b;
            e11 += w * (a * a);
            e22 += w * (b * b);
            e12 += w * (a * b);
        }
        s_h[0][r][c] = m1; s_h[1][r][c] = m2; s_h[2][r][c] = e11; s_h[3][r][c] = e22; s_h[4][r][c] = e12;
    }
    __syncthreads();
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
    for (int k = 0; k <= 2 * kR; k++) {                                     // columns
        const float w = win.w[k];
        mu1 += w * s_h[0][ty + k][tx];
        mu2 += w * s_h[1][ty + k][tx];
        e11 += w * s_h[2][ty + k][tx];
        e22 += w * s_h[3][ty + k][tx];
        e12 += w * s_h[4][ty + k][tx];
    }
    const int gx = blockIdx.x * kT + tx, gy = blockIdx.y * kT + ty;
    const bool in = gx < W && gy < H;
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
    const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu12;
    const float A = mu1_sq + mu2_sq + C1, B = s1 + s2 + C2, Cn = 2.f * mu12 + C1, D = 2.f * s12 + C2;
    const float inv_AB = 1.f / (A * B);
    const float map = Cn * D * inv_AB;
    if (in && dm) {
        const size_t p = base + (size_t)gy * W + gx, n = (size_t)gridDim.z * chan_stride;
        const bool w = gy >= dm_row0 && gy < dm_row1;
        dm[p] = w ? 2.f * mu2 * (D - Cn) * inv_AB - 2.f * mu1 * map * (B - A) * inv_AB : 0.f;
        dm[n + p] = w ? -map / B : 0.f;
        dm[2 * n + p] = w ? 2.f * Cn * inv_AB : 0.f;
    }
    // 6120 blocks at 960x540 adding into ONE double serialise in the L2 (it cost more than the rest of the kernel): spread
    // the partial sums over the slots sum[1..kSsimSlots-1]; ssim_sum_kernel folds them into sum[0]
    const float tot = block_sum_256((in && gy >= sum_row0 && gy < sum_row1) ? map : 0.f, s_part);
    if (threadIdx.x == 0) {
        const unsigned b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        atomicAdd(&sum[1 + b % (kSsimSlots - 1)], (double)tot);
    }
}

__global__ void __launch_bounds__(64) ssim_sum_kernel(double* __restrict__ sum)
{
    double v = threadIdx.x + 1 < kSsimSlots ? sum[threadIdx.x + 1] : 0.0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    if (threadIdx.x == 0) sum[0] = v;
}

__global__ void __launch_bounds__(256)
ssim_bwd_kernel(Window win, int H, int W, size_t chan_stride, const float* __restrict__ img1, const float* __restrict__ img2,
                const float* __restrict__ dm, float scale, const float* __restrict__ scale_dev, float* __restrict__ dimg1)
{
    __shared__ float s_d[3][kE][kES];
    __shared__ float s_h[3][kE][kT + 1];
    const size_t base = (size_t)blockIdx.z * chan_stride, n = (size_t)gridDim.z * chan_stride;
    const int x0 = blockIdx.x * kT - kR, y0 = blockIdx.y * kT - kR;
    for (int i = threadIdx.x; i < kE * kE; i += 256) {
        const int ly = i / kE, lx = i % kE, gx = x0 + lx, gy = y0 + ly;
        const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;
        const size_t p = base + (size_t)gy * W + gx;
#pragma unroll
        for (int m = 0; m < 3; m++) s_d[m][ly][lx] = in ? dm[m * n + p] : 0.f;   // no map outside the image
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kE * kT; i += 256) {
        const int r = i / kT, c = i % kT;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int k = 0; k <= 2 * kR; k++) {
            const float w = win.w[k];
            a0 += w * s_d[0][r][c + k];
            a1 += w * s_d[1][r][c + k];
            a2 += w * s_d[2][r][c + k];
        }
        s_h[0][r][c] = a0; s_h[1][r][c] = a1; s_h[2][r][c] = a2;
    }
    __syncthreads();
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int gx = blockIdx.x * kT + tx, gy = blockIdx.y * kT + ty;
    if (gx >= W || gy >= H) return;
    float b0 = 0.f, b1 = 0.f, b2 = 0.f;
#pragma unroll
    for (int k = 0; k <= 2 * kR; k++) {
        const float w = win.w[k];
        b0 += w * s_h[0][ty + k][tx];
        b1 += w * s_h[1][ty + k][tx];
        b2 += w * s_h[2][ty + k][tx];
    }
    const size_t p = base + (size_t)gy * W + gx;
    const float g = scale_dev ? scale * scale_dev[0] : scale;
    dimg1[p] += g * (b0 + 2.f * img1[p] * b1 + img2[p] * b2);
}

}  // namespace

extern "C" int mom_ssim_forward_slab(int C, int H, int W, size_t chan_stride, int sum_row0, int sum_row1, int dm_row0, int dm_row1,
                                     const float* window11, const float* img1, const float* img2, float* dm, double* sum,
                                     mom_stream_t stream)
{
    if (C < 0 || H < 0 || W < 0 || !window11 || !sum || chan_stride < (size_t)H * (size_t)W) return MOM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(sum, 0, sizeof(double) * kSsimSlots, s) != hipSuccess) return MOM_ELAUNCH;
    if (C == 0 || H == 0 || W == 0) return MOM_OK;
    if (!img1 || !img2) return MOM_EINVAL;
    Window win;
    for (int k = 0; k <= 2 * kR; k++) win.w[k] = window11[k];
    hipLaunchKernelGGL(ssim_fwd_kernel, dim3((W + kT - 1) / kT, (H + kT - 1) / kT, C), dim3(256), 0, s, win, H, W, chan_stride,
                       sum_row0, sum_row1, dm_row0, dm_row1, img1, img2, dm, sum);
    if (hipGetLastError() != hipSuccess) return MOM_ELAUNCH;
    hipLaunchKernelGGL(ssim_sum_kernel, dim3(1), dim3(64), 0, s, sum);
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}

extern "C" int mom_ssim_forward(int C, int H, int W, const float* window11, const float* img1, const float* img2, float* dm,
                                double* sum, mom_stream_t stream)
{
    return mom_ssim_forward_slab(C, H, W, (size_t)(H > 0 ? H : 0) * (size_t)(W > 0 ? W : 0), 0, H, 0, H, window11, img1, img2, dm,
                                 sum, stream);
}

extern "C" int mom_ssim_backward_slab(int C, int H, int W, size_t chan_stride, const float* window11, const float* img1,
                                      const float* img2, const float* dm, float scale, const float* scale_dev, float* dimg1,
                                      mom_stream_t stream)
{
    if (C < 0 || H < 0 || W < 0 || !window11 || chan_stride < (size_t)H * (size_t)W) return MOM_EINVAL;
    if (C == 0 || H == 0 || W == 0) return MOM_OK;
    if (!img1 || !img2 || !dm || !dimg1) return MOM_EINVAL;
    Window win;
    for (int k = 0; k <= 2 * kR; k++) win.w[k] = window11[k];
    hipLaunchKernelGGL(ssim_bwd_kernel, dim3((W + kT - 1) / kT, (H + kT - 1) / kT, C), dim3(256), 0, (hipStream_t)stream, win, H, W,
                       chan_stride, img1, img2, dm, scale, scale_dev, dimg1);
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}

extern "C" int mom_ssim_backward(int C, int H, int W, const float* window11, const float* img1, const float* img2, const float* dm,
                                 float scale, const float* scale_dev, float* dimg1, mom_stream_t stream)
{
    return mom_ssim_backward_slab(C, H, W, (size_t)(H > 0 ? H : 0) * (size_t)(W > 0 ? W : 0), window11, img1, img2, dm, scale,
                                  scale_dev, dimg1, stream);
}
