// Row selection for densify / prune, gfx950.
//
// The reference's optimizer surgery (scene/gaussian_model.py:409-482 _prune_optimizer / cat_tensors_to_optimizer /
// prune_points, :511-581 densify_and_split / densify_and_clone / prune) selects rows with `tensor[mask]`, once per
// parameter, per Adam moment and per auxiliary tensor -- about two dozen boolean-mask gathers per round, each with its own
// nonzero() and host synchronisation.  Here the mask is scanned ONCE into a plan (dst_index[i] = position of row i among the
// kept rows, or -1) and one kernel applies the plan to every tensor.
#include "mom_common.h"

namespace {

constexpr int kItems = 2048;       // rows per workgroup in the scan (256 threads x 8)

__device__ __forceinline__ int block_exclusive_scan_256(int v, int* s_wave, int& total)
{
    // exclusive prefix of one int per thread over 256 threads; total = sum over the workgroup
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wv; w++) base += s_wave[w];
    total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
    return base + incl - v;
}

__global__ void __launch_bounds__(256) select_count_kernel(int n, const uint8_t* __restrict__ keep, int* __restrict__ block_counts)
{
    __shared__ int s_wave[4];
    const int base = blockIdx.x * kItems + threadIdx.x * 8;
    int c = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) c += (base + k < n && keep[base + k]) ? 1 : 0;
    int total;
    block_exclusive_scan_256(c, s_wave, total);
    if (threadIdx.x == 0) block_counts[blockIdx.x] = total;
}

// exclusive scan of the workgroup counts in place (one workgroup; any number of counts), total to count_dev
__global__ void __launch_bounds__(256) select_scan_kernel(int nblocks, int* __restrict__ block_counts, int* __restrict__ count_dev)
{
    __shared__ int s_wave[4];
    int carry = 0;
    for (int b0 = 0; b0 < nblocks; b0 += 256) {
        const int i = b0 + threadIdx.x;
        const int v = i < nblocks ? block_counts[i] : 0;
        int total;
        const int ex = block_exclusive_scan_256(v, s_wave, total);
        if (i < nblocks) block_counts[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) *count_dev = carry;
}

__global__ void __launch_bounds__(256) select_index_kernel(int n, const uint8_t* __restrict__ keep, const int* __restrict__ block_offsets,
                                                          int* __restrict__ dst_index)
{
    __shared__ int s_wave[4];
    const int base = blockIdx.x * kItems + threadIdx.x * 8;
    int c = 0;
    bool k8[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        k8[k] = base + k < n && keep[base + k];
        c += k8[k] ? 1 : 0;
    }
    int total;
    int pos = block_offsets[blockIdx.x] + block_exclusive_scan_256(c, s_wave, total);
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (base + k < n) {
            dst_index[base + k] = k8[k] ? pos : -1;
            pos += k8[k] ? 1 : 0;
        }
}

struct SelectArgs {
    MomRowSelect t[MOM_SELECT_MAX_TENSORS];
};

// blockIdx.y = tensor, blockIdx.x = chunk of 256 rows.  The threads walk the chunk's words in order, so reads are coalesced and
// writes nearly so (kept rows are contiguous in the output).  WORD = 4 when the row size allows it, else 1.
template <int WORD>
__device__ __forceinline__ void apply_chunk(const MomRowSelect& t, int n, const int* __restrict__ dst_index)
{
    const int row0 = blockIdx.x * 256, rows = min(256, n - row0);
    const unsigned rw = t.row_bytes / WORD;
    const unsigned words = (unsigned)rows * rw;
    const char* __restrict__ src = (const char*)t.src;
    char* __restrict__ dst = (char*)t.dst;
    for (unsigned w = threadIdx.x; w < words; w += 256) {
        const unsigned r = w / rw, c = w - r * rw;
        const int di = dst_index[row0 + r];
        if (di < 0) continue;
        const size_t so = ((size_t)(row0 + r) * rw + c) * WORD, dof = ((size_t)di * rw + c) * WORD;
        if (WORD == 4) *(uint32_t*)(dst + dof) = *(const uint32_t*)(src + so);
        else dst[dof] = src[so];
    }
}
__global__ void __launch_bounds__(256) select_apply_kernel(SelectArgs a, int n, const int* __restrict__ dst_index)
{
    const MomRowSelect t = a.t[blockIdx.y];
    if (t.row_bytes == 0) return;
    if ((t.row_bytes & 3) == 0 && (((uintptr_t)t.src | (uintptr_t)t.dst) & 3) == 0) apply_chunk<4>(t, n, dst_index);
    else apply_chunk<1>(t, n, dst_index);
}

}  // namespace

extern "C" size_t mom_select_scratch_bytes(int n)
{
    const size_t blocks = ((size_t)(n > 0 ? n : 1) + kItems - 1) / kItems;
    return mom_align_up(blocks * sizeof(int)) + MOM_ALIGN;
}

extern "C" int mom_select_plan(int n, const uint8_t* keep, int* dst_index, int* count_dev, int* count_host, void* scratch,
                               mom_stream_t stream)
{
    if (n < 0 || !count_dev) return MOM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        if (hipMemsetAsync(count_dev, 0, sizeof(int), s) != hipSuccess) return MOM_ELAUNCH;
    } else {
        if (!keep || !dst_index || !scratch) return MOM_EINVAL;
        int* block_counts = (int*)mom_align_ptr(scratch);
        const int blocks = (n + kItems - 1) / kItems;
        hipLaunchKernelGGL(select_count_kernel, dim3(blocks), dim3(256), 0, s, n, keep, block_counts);
        hipLaunchKernelGGL(select_scan_kernel, dim3(1), dim3(256), 0, s, blocks, block_counts, count_dev);
        hipLaunchKernelGGL(select_index_kernel, dim3(blocks), dim3(256), 0, s, n, keep, block_counts, dst_index);
        if (hipGetLastError() != hipSuccess) return MOM_ELAUNCH;
    }
    if (count_host && hipMemcpyAsync(count_host, count_dev, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess) return MOM_ELAUNCH;
    return MOM_OK;
}

extern "C" int mom_select_apply(int n, const int* dst_index, const MomRowSelect* tensors, int count, mom_stream_t stream)
{
    if (n < 0 || count < 0 || count > MOM_SELECT_MAX_TENSORS || (count && !tensors)) return MOM_EINVAL;
    if (n == 0 || count == 0) return MOM_OK;
    if (!dst_index) return MOM_EINVAL;
    SelectArgs a;
    for (int i = 0; i < count; i++) {
        a.t[i] = tensors[i];
        if (a.t[i].row_bytes && (!a.t[i].src || !a.t[i].dst)) return MOM_EINVAL;
    }
    hipLaunchKernelGGL(select_apply_kernel, dim3((n + 255) / 256, count), dim3(256), 0, (hipStream_t)stream, a, n, dst_index);
    return hipGetLastError() == hipSuccess ? MOM_OK : MOM_ELAUNCH;
}
