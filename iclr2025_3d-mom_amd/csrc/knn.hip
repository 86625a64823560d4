// distCUDA2: mean squared distance to the 3 nearest neighbours (init only), gfx950.
//
// Replaces SimpleKNN::knn (reference submodules/simple-knn/simple_knn.cu:45-221): bounding box seeded with
// {0,0,0} (the reference's cub::DeviceReduce init, :192-201), 30-bit Morton codes, stable radix sort of
// (code, index), boxes of 1024 consecutive sorted points, +-3 Morton neighbours to seed the reject radius,
// then an exact box-pruned 3-NN.  No host round trips: the bounding box stays on the device.
#include "mom_common.h"
#include <float.h>

namespace {

constexpr int kBox = 1024;
constexpr int kSortItems = 4096;  // items per workgroup and radix pass

__global__ void __launch_bounds__(256) bbox_kernel(int P, const float* __restrict__ pts, float* __restrict__ bb /* min3, max3 */)
{
    __shared__ float s[6][4];
    float mn[3] = {0.f, 0.f, 0.f}, mx[3] = {0.f, 0.f, 0.f};  // seeded with 0 like the reference
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256)
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float v = pts[3 * i + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], v);
        }
#pragma unroll
    for (int a = 0; a < 3; a++) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], d));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], d));
        }
        if ((threadIdx.x & 63) == 0) {
            s[a][threadIdx.x >> 6] = mn[a];
            s[3 + a][threadIdx.x >> 6] = mx[a];
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        const float lo = fminf(fminf(s[a][0], s[a][1]), fminf(s[a][2], s[a][3]));
        const float hi = fmaxf(fmaxf(s[3 + a][0], s[3 + a][1]), fmaxf(s[3 + a][2], s[3 + a][3]));
        // lo <= 0 <= hi always (seeded with 0): among non-positive floats the most negative has the largest unsigned
        // bit pattern, among non-negative floats the largest has, so both reduce with an unsigned atomicMax from 0.
        atomicMax((unsigned*)&bb[a], __float_as_uint(lo));
        atomicMax((unsigned*)&bb[3 + a], __float_as_uint(hi));
    }
}

__device__ __forceinline__ unsigned prep_morton(unsigned x)
{
    x = (x | (x << 16)) & 0x030000FF;
    x = (x | (x << 8)) & 0x0300F00F;
    x = (x | (x << 4)) & 0x030C30C3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}
__global__ void __launch_bounds__(256) morton_kernel(int P, const float* __restrict__ pts, const float* __restrict__ bb,
                                                    unsigned* __restrict__ codes, unsigned* __restrict__ idx)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    unsigned c[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float lo = bb[a], hi = bb[3 + a];
        c[a] = prep_morton((unsigned)(((pts[3 * i + a] - lo) / (hi - lo)) * ((1 << 10) - 1)));
    }
    codes[i] = c[0] | (c[1] << 1) | (c[2] << 2);
    idx[i] = (unsigned)i;
}

// ---- stable LSD radix sort of (u32 key, u32 value), 8 bits per pass ----
__global__ void __launch_bounds__(256) rs_hist_kernel(int n, int nb, int shift, const unsigned* __restrict__ keys,
                                                     unsigned* __restrict__ counts /* [256][nb] */)
{
    __shared__ unsigned h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int base = blockIdx.x * kSortItems;
    for (int k = threadIdx.x; k < kSortItems && base + k < n; k += 256) atomicAdd(&h[(keys[base + k] >> shift) & 255u], 1u);
    __syncthreads();
    counts[threadIdx.x * nb + blockIdx.x] = h[threadIdx.x];
}

__global__ void __launch_bounds__(1024) rs_scan_kernel(int m, unsigned* __restrict__ counts)
{
    __shared__ unsigned s_wave[16];
    __shared__ unsigned s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < m; base += 1024) {
        const int t = base + threadIdx.x;
        const unsigned v = t < m ? counts[t] : 0u;
        unsigned incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        unsigned woff = 0;
        for (int w = 0; w < wave; w++) woff += s_wave[w];
        const unsigned start = s_carry + woff + incl - v;
        if (t < m) counts[t] = start;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = start + v;
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) rs_scatter_kernel(int n, int nb, int shift, const unsigned* __restrict__ keys,
                                                        const unsigned* __restrict__ vals, const unsigned* __restrict__ offsets,
                                                        unsigned* __restrict__ keys_out, unsigned* __restrict__ vals_out)
{
    __shared__ unsigned cur[256];
    cur[threadIdx.x] = offsets[threadIdx.x * nb + blockIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int base = blockIdx.x * kSortItems;
    for (int c = 0; c < kSortItems; c += 256) {
        const int i = base + c + threadIdx.x;
        const bool ok = i < n;
        const unsigned key = ok ? keys[i] : 0u;
        const unsigned dgt = (key >> shift) & 255u;
        // lanes of this wave with the same digit
        unsigned long long peers = __ballot(ok);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const unsigned long long m = __ballot((dgt >> b) & 1u);
            peers &= ((dgt >> b) & 1u) ? m : ~m;
        }
        const unsigned rank = __popcll(peers & ((1ull << lane) - 1ull));
        const bool leader = ok && rank == 0;
        unsigned pos = 0;
        for (int w = 0; w < 4; w++) {  // waves in order => stable
            if (wave == w && ok) pos = cur[dgt] + rank;
            __syncthreads();
            if (wave == w && leader) cur[dgt] += (unsigned)__popcll(peers);
            __syncthreads();
        }
        if (ok) {
            keys_out[pos] = key;
            vals_out[pos] = vals[i];
        }
    }
}

struct Box { float mn[3], mx[3]; };

__global__ void __launch_bounds__(kBox) box_minmax_kernel(int P, const float* __restrict__ pts, const unsigned* __restrict__ idx,
                                                         Box* __restrict__ boxes)
{
    __shared__ float s[6][16];
    const int i = blockIdx.x * kBox + threadIdx.x;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    if (i < P) {
        const unsigned g = idx[i];
#pragma unroll
        for (int a = 0; a < 3; a++) mn[a] = mx[a] = pts[3 * g + a];
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], d));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], d));
        }
        if ((threadIdx.x & 63) == 0) {
            s[a][threadIdx.x >> 6] = mn[a];
            s[3 + a][threadIdx.x >> 6] = mx[a];
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        float lo = FLT_MAX, hi = -FLT_MAX;
        for (int w = 0; w < 16; w++) {
            lo = fminf(lo, s[a][w]);
            hi = fmaxf(hi, s[3 + a][w]);
        }
        boxes[blockIdx.x].mn[a] = lo;
        boxes[blockIdx.x].mx[a] = hi;
    }
}

__device__ __forceinline__ void k_best3(const float* ref, const float* p, float* best)
{
    const float dx = p[0] - ref[0], dy = p[1] - ref[1], dz = p[2] - ref[2];
    float dist = dx * dx + dy * dy + dz * dz;
#pragma unroll
    for (int j = 0; j < 3; j++)
        if (best[j] > dist) {
            const float t = best[j];
            best[j] = dist;
            dist = t;
        }
}

__global__ void __launch_bounds__(256) box_mean_dist_kernel(int P, const float* __restrict__ pts, const unsigned* __restrict__ idx,
                                                           const Box* __restrict__ boxes, float* __restrict__ out)
{
    const int i0 = blockIdx.x * 256 + threadIdx.x;
    if (i0 >= P) return;
    const unsigned self = idx[i0];
    const float p[3] = {pts[3 * self], pts[3 * self + 1], pts[3 * self + 2]};
    float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    for (int i = max(0, i0 - 3); i <= min(P - 1, i0 + 3); i++)
        if (i != i0) k_best3(p, pts + 3 * idx[i], best);
    const float reject = best[2];
    best[0] = best[1] = best[2] = FLT_MAX;
    const int nbx = (P + kBox - 1) / kBox;
    for (int b = 0; b < nbx; b++) {
        const Box bx = boxes[b];
        float d2 = 0.f;
#pragma unroll
        for (int a = 0; a < 3; a++)
            if (p[a] < bx.mn[a] || p[a] > bx.mx[a]) {
                const float d = fminf(fabsf(p[a] - bx.mn[a]), fabsf(p[a] - bx.mx[a]));
                d2 += d * d;
            }
        if (d2 > reject || d2 > best[2]) continue;
        for (int i = b * kBox; i < min(P, (b + 1) * kBox); i++)
            if (i != i0) k_best3(p, pts + 3 * idx[i], best);
    }
    out[self] = (best[0] + best[1] + best[2]) / 3.0f;
}

}  // namespace

// stable LSD radix sort of (u32 key, u32 value) pairs on the low `bits` bits, 8 per pass; keys[0] / vals[0] hold the input,
// the pair of buffers holding the result is returned (0 / 1).  counts: mom_sort_pairs_counts_bytes(n) bytes.
size_t mom_sort_pairs_counts_bytes(int n)
{
    const size_t nb = ((size_t)(n > 0 ? n : 1) + kSortItems - 1) / kSortItems;
    return 256 * nb * 4;
}
int mom_sort_pairs_u32(int n, int bits, unsigned* keys[2], unsigned* vals[2], unsigned* counts, hipStream_t s)
{
    const int nb = (int)(((size_t)n + kSortItems - 1) / kSortItems);
    int cur = 0;
    for (int shift = 0; shift < bits; shift += 8) {
        hipLaunchKernelGGL(rs_hist_kernel, dim3(nb), dim3(256), 0, s, n, nb, shift, keys[cur], counts);
        hipLaunchKernelGGL(rs_scan_kernel, dim3(1), dim3(1024), 0, s, 256 * nb, counts);
        hipLaunchKernelGGL(rs_scatter_kernel, dim3(nb), dim3(256), 0, s, n, nb, shift, keys[cur], vals[cur], counts, keys[cur ^ 1],
                           vals[cur ^ 1]);
        cur ^= 1;
    }
    return hipGetLastError() == hipSuccess ? cur : MOM_ELAUNCH;
}

extern "C" size_t mom_knn_scratch_bytes(int P)
{
    const size_t n = (size_t)(P > 0 ? P : 1);
    const size_t nb = (n + kSortItems - 1) / kSortItems;
    return mom_align_up(64) + 4 * mom_align_up(n * 4) + mom_align_up(256 * nb * 4) + mom_align_up(((n + kBox - 1) / kBox) * sizeof(Box)) +
           MOM_ALIGN;
}

extern "C" int mom_knn_mean_dist2(int P, const float* points, float* mean_dist2, void* scratch, mom_stream_t stream)
{
    if (P < 0) return MOM_EINVAL;
    if (P == 0) return MOM_OK;
    if (!points || !mean_dist2 || !scratch) return MOM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)P;
    const int nb = (int)((n + kSortItems - 1) / kSortItems);
    char* base = mom_align_ptr(scratch);
    float* bb = (float*)base; base += mom_align_up(64);
    unsigned* keys[2]; unsigned* vals[2];
    keys[0] = (unsigned*)base; base += mom_align_up(n * 4);
    keys[1] = (unsigned*)base; base += mom_align_up(n * 4);
    vals[0] = (unsigned*)base; base += mom_align_up(n * 4);
    vals[1] = (unsigned*)base; base += mom_align_up(n * 4);
    unsigned* counts = (unsigned*)base; base += mom_align_up((size_t)256 * nb * 4);
    Box* boxes = (Box*)base;

    if (hipMemsetAsync(bb, 0, 64, s) != hipSuccess) return MOM_ELAUNCH;
    int rb = (P + 255) / 256;
    if (rb > 1024) rb = 1024;
    hipLaunchKernelGGL(bbox_kernel, dim3(rb), dim3(256), 0, s, P, points, bb);
    hipLaunchKernelGGL(morton_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, points, bb, keys[0], vals[0]);
    int cur = 0;
    for (int shift = 0; shift < 32; shift += 8) {
        hipLaunchKernelGGL(rs_hist_kernel, dim3(nb), dim3(256), 0, s, P, nb, shift, keys[cur], counts);
        hipLaunchKernelGGL(rs_scan_kernel, dim3(1), dim3(1024), 0, s, 256 * nb, counts);
        hipLaunchKernelGGL(rs_scatter_kernel, dim3(nb), dim3(256), 0, s, P, nb, shift, keys[cur], vals[cur], counts, keys[cur ^ 1],
                           vals[cur ^ 1]);
        cur ^= 1;
    }
    const int nbx = (P + kBox - 1) / kBox;
    hipLaunchKernelGGL(box_minmax_kernel, dim3(nbx), dim3(kBox), 0, s, P, points, vals[cur], boxes);
    hipLaunchKernelGGL(box_mean_dist_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, points, vals[cur], boxes, mean_dist2);
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}

// Spatial (Morton) processing order of a point set: the first half of SimpleKNN::knn on its own.  Used to walk the
// Gaussians in a cache- and atomics-friendly order in the HexPlane kernels (results never depend on it).
extern "C" size_t mom_morton_order_scratch_bytes(int P) { return mom_knn_scratch_bytes(P); }
extern "C" int mom_morton_order(int P, const float* points, uint32_t* order, void* scratch, mom_stream_t stream)
{
    if (P < 0) return MOM_EINVAL;
    if (P == 0) return MOM_OK;
    if (!points || !order || !scratch) return MOM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)P;
    const int nb = (int)((n + kSortItems - 1) / kSortItems);
    char* base = mom_align_ptr(scratch);
    float* bb = (float*)base; base += mom_align_up(64);
    unsigned* keys[2]; unsigned* vals[2];
    keys[0] = (unsigned*)base; base += mom_align_up(n * 4);
    keys[1] = (unsigned*)base; base += mom_align_up(n * 4);
    vals[0] = (unsigned*)base; base += mom_align_up(n * 4);
    vals[1] = (unsigned*)base; base += mom_align_up(n * 4);
    unsigned* counts = (unsigned*)base;
    if (hipMemsetAsync(bb, 0, 64, s) != hipSuccess) return MOM_ELAUNCH;
    int rb = (P + 255) / 256;
    if (rb > 1024) rb = 1024;
    hipLaunchKernelGGL(bbox_kernel, dim3(rb), dim3(256), 0, s, P, points, bb);
    hipLaunchKernelGGL(morton_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, points, bb, keys[0], vals[0]);
    int cur = 0;
    for (int shift = 0; shift < 32; shift += 8) {
        hipLaunchKernelGGL(rs_hist_kernel, dim3(nb), dim3(256), 0, s, P, nb, shift, keys[cur], counts);
        hipLaunchKernelGGL(rs_scan_kernel, dim3(1), dim3(1024), 0, s, 256 * nb, counts);
        hipLaunchKernelGGL(rs_scatter_kernel, dim3(nb), dim3(256), 0, s, P, nb, shift, keys[cur], vals[cur], counts, keys[cur ^ 1],
                           vals[cur ^ 1]);
        cur ^= 1;
    }
    if (hipMemcpyAsync(order, vals[cur], n * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return MOM_ELAUNCH;
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}
