// What does one ds_add_f32 wave instruction cost in the shape a row-list compositing backward would use -- four 16-lane rows, each
// adding nine consecutive floats into the accumulator record of ITS splat (36 active lanes, four records) -- against the same shape
// as a global atomic?  Prints cycles per wave instruction per CU at the measured wall time (2.1 GHz nominal).
//   hipcc --offload-arch=gfx950 -O3 tools/probe/lds_row_atomic.hip -o /tmp/lds_row && /tmp/lds_row
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#pragma clang diagnostic ignored "-Wunused-result"

constexpr int kSlots = 256;
template <int STRIDE, int MODE, int LANES = 9, int ROWS = 4>      // MODE 0: LDS atomic, 1: global atomic (records of STRIDE floats), 2: LDS atomic + 48 filler VALU
__global__ void __launch_bounds__(256) probe(int iters, float* g, int P, float* out)
{
    __shared__ float acc[kSlots * STRIDE];
    for (int i = threadIdx.x; i < kSlots * STRIDE; i += 256) acc[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, row = lane >> 4, k = lane & 15;
    unsigned h = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2654435761u + row * 40503u;
    float v = 1.0f + lane, f = 0.5f;
    for (int i = 0; i < iters; i++) {
        h = h * 1664525u + 1013904223u;
        const unsigned slot = (h >> 10);
        if (MODE == 2) {
#pragma unroll
            for (int q = 0; q < 48; q++) f = __builtin_fmaf(f, 0.999f, v);
        }
        if (k < LANES && row < ROWS) {
            if (MODE == 1) atomicAdd(&g[(size_t)(slot % (unsigned)P) * STRIDE + k], v);
            else atomicAdd(&acc[(slot % kSlots) * STRIDE + k], v);
        }
    }
    __syncthreads();
    if (out) out[blockIdx.x * 256 + threadIdx.x] = acc[threadIdx.x] + f;
}

template <int STRIDE, int MODE, int LANES = 9, int ROWS = 4>
void run(const char* name, float* g, int P, float* out, int cus = 256)
{
    const int iters = 2000, wgs = cus * 5;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    probe<STRIDE, MODE, LANES, ROWS><<<wgs, 256>>>(50, g, P, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe<STRIDE, MODE, LANES, ROWS><<<wgs, 256>>>(iters, g, P, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double per_cu = (double)iters * 4 * 5;          // wave instructions per CU
    printf("%-60s %8.3f ms  %7.1f cycles per wave instruction per CU\n", name, ms, ms * 1e-3 * 2.1e9 / per_cu);
}

int main()
{
    const int P = 200000;
    float *g, *out;
    hipMalloc(&g, (size_t)P * 16 * 4); hipMemset(g, 0, (size_t)P * 16 * 4);
    hipMalloc(&out, 256 * 5 * 256 * 4);
    run<12, 0>("ds_add_f32, 4 rows x 9 lanes, records of 12 floats", g, P, out);
    run<16, 0>("ds_add_f32, 4 rows x 9 lanes, records of 16 floats", g, P, out);
    run<9, 0>("ds_add_f32, 4 rows x 9 lanes, records of 9 floats", g, P, out);
    run<12, 2>("ds_add_f32 + 48 fma per iteration, records of 12 floats", g, P, out);
    run<12, 1>("global_atomic_add_f32, records of 12 floats (200k records)", g, P, out);
    run<16, 1>("global_atomic_add_f32, records of 16 floats (200k records)", g, P, out);
    run<16, 1, 9, 1>("global, 1 row x 9 lanes, 16-float records", g, P, out);
    run<16, 1, 9, 2>("global, 2 rows x 9 lanes, 16-float records", g, P, out);
    run<16, 1, 4, 4>("global, 4 rows x 4 lanes, 16-float records", g, P, out);
    run<16, 1, 16, 4>("global, 4 rows x 16 lanes, 16-float records", g, P, out);
    run<16, 1, 1, 4>("global, 4 rows x 1 lane, 16-float records", g, P, out);
    run<16, 1, 9, 4>("global, 4 rows x 9 lanes, 16-float records, 64 workgroup slots (1/4 of the chip)", g, P, out, 64);
    run<16, 1, 9, 4>("global, 4 rows x 9 lanes, 16-float records, 16 (1/16 of the chip)", g, P, out, 16);
    run<16, 0, 9, 1>("ds_add_f32, 1 row x 9 lanes", g, P, out);
    run<16, 0, 1, 4>("ds_add_f32, 4 rows x 1 lane", g, P, out);
    return 0;
}
