// Micro-benchmark of v_mfma_f32_32x32x2_f32 issue patterns on gfx950 (build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip).
// Prints TFLOP/s for: one accumulator chain, two and four interleaved chains, and each with the A operand read from LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDSA, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) probe(float* out, int iters, const float* wsrc)
{
    __shared__ float w[64 * 65];
    for (int i = threadIdx.x; i < 64 * 65; i += blockDim.x) w[i] = wsrc[i % 4096];
    __syncthreads();
    const int lane = threadIdx.x & 63, col = lane & 31, h = lane >> 5;
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; a++)
        for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
    float b = out[threadIdx.x & 7], areg = wsrc[lane];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 32; k++) {
#pragma unroll
            for (int a = 0; a < NACC; a++) {
                const float av = LDSA ? w[(k + 2 * h) * 65 + col + 32 * (a & 1)] : areg;
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[a], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int a = 0; a < NACC; a++)
        for (int r = 0; r < 16; r++) s += acc[a][r];
    if (s == 123.456f) out[0] = s;
}

template <int NACC, bool LDSA, int WAVES>
void run(const char* name, float* d_out, float* d_w)
{
    const int iters = 400 / NACC;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256;
    probe<NACC, LDSA, WAVES><<<blocks, 64 * WAVES>>>(d_out, 10, d_w);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<NACC, LDSA, WAVES><<<blocks, 64 * WAVES>>>(d_out, iters, d_w);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)blocks * WAVES * iters * 32 * NACC;
    const double tf = mfmas * 4096.0 / (ms * 1e-3) / 1e12;
    const double cyc = (ms * 1e-3 * 2.4e9) / ((double)iters * 32 * NACC * WAVES / 4.0);   // cycles per MFMA per SIMD at 2.4 GHz
    printf("%-44s %7.1f TFLOP/s   %6.1f cyc/MFMA/SIMD @2.4GHz   (%.3f ms)\n", name, tf, cyc, ms);
}

int main()
{
    float *d_out, *d_w;
    hipMalloc(&d_out, 4096); hipMalloc(&d_w, 4096 * 4);
    hipMemset(d_out, 0, 4096); hipMemset(d_w, 0, 4096 * 4);
    run<1, false, 4>("1 chain, A in register, 1 wave/SIMD", d_out, d_w);
    run<2, false, 4>("2 chains, A in register, 1 wave/SIMD", d_out, d_w);
    run<4, false, 4>("4 chains, A in register, 1 wave/SIMD", d_out, d_w);
    run<1, true, 4>("1 chain, A from LDS, 1 wave/SIMD", d_out, d_w);
    run<2, true, 4>("2 chains, A from LDS, 1 wave/SIMD", d_out, d_w);
    run<4, true, 4>("4 chains, A from LDS, 1 wave/SIMD", d_out, d_w);
    run<2, true, 8>("2 chains, A from LDS, 2 waves/SIMD", d_out, d_w);
    run<2, true, 16>("2 chains, A from LDS, 4 waves/SIMD", d_out, d_w);
    run<1, true, 16>("1 chain, A from LDS, 4 waves/SIMD", d_out, d_w);
    return 0;
}
