// Do vector instructions of one wave issue while another wave's f32 MFMAs occupy the SIMD's matrix pipe?
// Workgroups of 512 threads (two waves per SIMD): waves 0-3 run `nm` dependent-chain v_mfma_f32_32x32x2_f32, waves 4-7 run `nv`
// v_fma_f32 (8 independent chains).  Times: MFMA alone, VALU alone, both.  If they overlap, both ~ max; if not, both ~ sum.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_valu_coissue.hip -o /tmp/coissue && /tmp/coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(512) k(int nm, int nv, float* out)
{
    const int wv = threadIdx.x >> 6;
    float r = 0.f;
    if (wv < 4) {
        f32x16 acc = {0};
        float a = threadIdx.x * 1e-3f, b = 1.0f;
        for (int i = 0; i < nm; i++) {
#pragma unroll
            for (int j = 0; j < 16; j++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        r = acc[0] + acc[5];
    } else {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = threadIdx.x + j;
        for (int i = 0; i < nv; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++)
#pragma unroll
                for (int j = 0; j < 8; j++) x[j] = __builtin_fmaf(x[j], 1.0001f, 0.5f);
        }
#pragma unroll
        for (int j = 0; j < 8; j++) r += x[j];
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

static float run(int nm, int nv, float* out)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<<<256, 512>>>(nm / 8, nv / 8, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    k<<<256, 512>>>(nm, nv, out);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main()
{
    float* out;
    (void)hipMalloc(&out, 256 * 512 * 4);
    const int nm = 4096;            // x16 MFMAs = 65536 MFMAs per wave = 4.19 M cycles at 64 each
    for (int nv : {0, 4096, 8192, 16384, 32768}) {       // x64 v_fma per wave
        const float tm = run(nm, 0, out), tv = nv ? run(0, nv, out) : 0.f, tb = run(nm, nv, out);
        printf("MFMA %d x16: %.3f ms alone | VALU %d x64 fma: %.3f ms alone | together %.3f ms  (sum %.3f, max %.3f)\n", nm, tm, nv, tv, tb,
               tm + tv, tm > tv ? tm : tv);
    }
    return 0;
}
