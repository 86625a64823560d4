// Throughput of LDS float atomics (ds_add_f32, no return) in the access shapes the HexPlane backward could use, against plain
// ds_write_b32 of the same shape.  Prints cycles per wave instruction per CU (at the measured wall time and 2.4 GHz nominal).
//   hipcc --offload-arch=gfx950 -O3 tools/probe/lds_atomic_probe.hip -o /tmp/lds_probe && /tmp/lds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int kRows = 256;     // 32 KB of accumulators
template <int MODE>
__global__ void __launch_bounds__(1024) probe(int iters, const int* __restrict__ rows, float* out)
{
    __shared__ float acc[kRows * 32];
    for (int i = threadIdx.x; i < kRows * 32; i += blockDim.x) acc[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ch = lane & 31, h = lane >> 5, g8 = lane >> 3, c = lane & 7;
    float v = 1.0f + lane;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0 || MODE == 1) {           // half-wave per row, lane = channel
            const int r = rows[(i * 16 + wv) * 2 + h & 4095];
            if (MODE == 0) atomicAdd(&acc[r * 32 + ch], v); else acc[r * 32 + ch] = v;
        } else if (MODE == 2 || MODE == 3) {    // eight lanes per row, four channels per lane: 4 instructions per row batch
            const int r = MODE == 2 ? rows[(i * 16 + wv) * 8 + g8 & 4095] : rows[(i * 16 + wv) & 4095];
#pragma unroll
            for (int j = 0; j < 4; j++) atomicAdd(&acc[r * 32 + 4 * c + j], v);
        } else if (MODE == 4) {                 // eight lanes per row, channel-major image: acc[j][row][c] -> bank = 8 row + c
            const int r = rows[(i * 16 + wv) * 8 + g8 & 4095];
#pragma unroll
            for (int j = 0; j < 4; j++) atomicAdd(&acc[j * (kRows * 8) + r * 8 + c], v);
        } else if (MODE == 5) {                 // returning atomic, half-wave rows
            const int r = rows[(i * 16 + wv) * 2 + h & 4095];
            v += atomicAdd(&acc[r * 32 + ch], v) * 1e-30f;
        }
    }
    __syncthreads();
    if (out) out[blockIdx.x * blockDim.x + threadIdx.x] = acc[threadIdx.x] + v;
}

template <int MODE>
void run(const char* name, int threads, int instr_per_iter, const int* rows, float* out)
{
    const int iters = 4096;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    probe<MODE><<<256, threads>>>(64, rows, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe<MODE><<<256, threads>>>(iters, rows, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double wave_instr_per_cu = (double)iters * (threads / 64) * instr_per_iter;
    printf("%-44s %4d thr  %8.3f ms  %6.1f ns/wave-instr/CU = %5.1f cycles @2.4GHz\n", name, threads, ms, ms * 1e6 / wave_instr_per_cu,
           ms * 1e6 / wave_instr_per_cu * 2.4);
}

int main()
{
    int* rows; float* out;
    int h[4096];
    srand(1);
    for (int i = 0; i < 4096; i++) h[i] = rand() % kRows;
    hipMalloc(&rows, sizeof(h)); hipMemcpy(rows, h, sizeof(h), hipMemcpyHostToDevice);
    hipMalloc(&out, 256 * 1024 * 4);
    for (int threads : {256, 512, 1024}) {
        run<0>("ds_add_f32 half-wave rows (lane=channel)", threads, 1, rows, out);
        run<1>("ds_write_b32 half-wave rows", threads, 1, rows, out);
        run<2>("ds_add_f32 8 lanes/row, 8 random rows", threads, 4, rows, out);
        run<3>("ds_add_f32 8 lanes/row, all the same row", threads, 4, rows, out);
        run<4>("ds_add_f32 8 lanes/row, channel-major image", threads, 4, rows, out);
        run<5>("ds_add_rtn_f32 half-wave rows", threads, 1, rows, out);
    }
    return 0;
}
