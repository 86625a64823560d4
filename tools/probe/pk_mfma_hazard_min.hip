// Minimal stand-alone attempt at the packed-fp32 / bf16-MFMA corruption, built from what the ISA-level bisection found
// (tools/probe/pk_bisect/, profiles/r06_probes/pk_bisect.json): in deform_field_fwd_b3_kernel compiled with the SLP vectoriser, TWO
// instructions are each sufficient and together necessary for the wrong features --
//     v_pk_fma_f32 v[4:5], v[68:69], v[40:41], v[4:5] op_sel:[0,1,0]      (and the same with v[70:71])
// -- the only two packed instructions of the kernel whose LOW result reads the HIGH half of a source pair (op_sel bit set on src1;
// every op_sel_hi form, op_sel on src0 of a v_pk_mul and the plain forms are innocent).  Here: vector waves evaluate that form and
// its scalar equivalent on the same operands and count disagreements by lane quarter, while matrix waves on the same SIMDs loop
// v_mfma_f32_32x32x16_bf16.      hipcc --offload-arch=gfx950 -O3 tools/probe/pk_mfma_hazard_min.hip -o /tmp/pkmin && /tmp/pkmin [launches]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int kIters = 256;

// FORM 0: op_sel:[0,1,0] (the failing one)  1: op_sel:[1,0,0]  2: op_sel:[0,0,1]  3: op_sel_hi:[1,0,1] (innocent in the real kernel)
template <int FORM> __device__ __forceinline__ f32x2 pk(f32x2 a, f32x2 b, f32x2 c)
{
    f32x2 d;
    if (FORM == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    if (FORM == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    if (FORM == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    if (FORM == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    // FORM 4: the failing instruction with the real kernel's registers -- v[4:5] += v[68:69] * v41: the low computation reads banks
    // 0, 1, 0 (v68, v41, v4), the high one 1, 1, 1 (v69, v41, v5)
    if (FORM == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "={v[4:5]}"(d) : "{v[68:69]}"(a), "{v[40:41]}"(b), "0"(c));
    return d;
}
template <int FORM> __device__ __forceinline__ f32x2 scalar(f32x2 a, f32x2 b, f32x2 c)
{
    const float alo = FORM == 1 ? a.y : a.x, blo = (FORM == 0 || FORM == 4) ? b.y : b.x, clo = FORM == 2 ? c.y : c.x, bhi = FORM == 3 ? b.x : b.y;
    f32x2 d;
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d.x) : "v"(alo), "v"(blo), "v"(clo));
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d.y) : "v"(a.y), "v"(bhi), "v"(c.y));
    return d;
}

template <int FORM, int WAVES, bool MFMA>
__global__ void __launch_bounds__(64 * WAVES) probe(const float4* __restrict__ data, unsigned* __restrict__ wrong, float* sink)
{
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __shared__ bf16x8 s_w[4][64 * 8];
    if (wv < 4) {                                   // matrix waves: one per SIMD (waves w, w + 4, w + 8 ... share SIMD w & 3)
        if (!MFMA) return;
        f32x16 acc0 = {0}, acc1 = {0};
        bf16x8 a, b;
        for (int j = 0; j < 8; j++) { a[j] = (__bf16)(0.001f * (lane + j)); b[j] = (__bf16)1.0f; }
        for (int j = 0; j < 8; j++) s_w[wv][j * 64 + lane] = a;
        __builtin_amdgcn_wave_barrier();
        for (int i = 0; i < kIters * 6; i++) {
            const bf16x8 w = s_w[wv][(i & 7) * 64 + lane];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, w, acc1, 0, 0, 0);
        }
        if (acc0[3] + acc1[7] == 12345.678f) sink[0] = acc0[0];
        return;
    }
    unsigned bad = 0;
    const int vw = blockIdx.x * (WAVES - 4) + (wv - 4);
    f32x2 b = {0.25f + 0.001f * lane, 0.75f - 0.002f * lane};          // long-lived, as v[40:41] in the real kernel
    for (int it = 0; it < kIters; it++) {
        const unsigned idx = ((unsigned)(vw * kIters + it) * 64u + lane) & 0xFFFFu;
        const float4 q = data[idx], r = data[(idx * 2654435761u) >> 16];        // two loads in flight, as the gather has
        f32x2 c = {r.x, r.y};
        const f32x2 p0 = pk<FORM>(f32x2{q.x, q.y}, b, c), s0 = scalar<FORM>(f32x2{q.x, q.y}, b, c);
        const f32x2 p1 = pk<FORM>(f32x2{q.z, q.w}, b, p0), s1 = scalar<FORM>(f32x2{q.z, q.w}, b, s0);
        bad += (__float_as_uint(p1.x) != __float_as_uint(s1.x)) + 2 * (__float_as_uint(p1.y) != __float_as_uint(s1.y) ? 1 : 0) * 0x10000;
        b.x += 1e-6f * p1.y;                         // keep b live and changing
    }
    if (bad) atomicAdd(&wrong[(lane >> 4) * 2 + 0], bad & 0xFFFF), atomicAdd(&wrong[(lane >> 4) * 2 + 1], bad >> 17);
}

template <int FORM, int WAVES, bool MFMA> static void run(const char* name, int launches, const float4* data, unsigned* wrong, float* sink)
{
    (void)hipMemset(wrong, 0, 32);
    long wrong_launches = 0;
    unsigned h[8], prev[8] = {0};
    for (int l = 0; l < launches; l++) {
        hipLaunchKernelGGL((probe<FORM, WAVES, MFMA>), dim3(256), dim3(64 * WAVES), 0, 0, data, wrong, sink);
        (void)hipMemcpy(h, wrong, 32, hipMemcpyDeviceToHost);
        bool any = false;
        for (int k = 0; k < 8; k++) { any |= h[k] != prev[k]; prev[k] = h[k]; }
        wrong_launches += any;
    }
    printf("  {\"case\": \"%s\", \"waves_per_workgroup\": %d, \"mfma\": %s, \"wrong_launches\": %ld, \"of\": %d, \"wrong_low_results_by_lane_quarter\": [%u, %u, %u, %u], "
           "\"wrong_high_results_by_lane_quarter\": [%u, %u, %u, %u]},\n", name, WAVES, MFMA ? "true" : "false", wrong_launches, launches, h[0], h[2], h[4], h[6], h[1], h[3], h[5], h[7]);
}

int main(int argc, char** argv)
{
    const int launches = argc > 1 ? atoi(argv[1]) : 200;
    float4* data; unsigned* wrong; float* sink;
    (void)hipMalloc(&data, 65536 * 16); (void)hipMalloc(&wrong, 32); (void)hipMalloc(&sink, 4);
    float4* h = (float4*)malloc(65536 * 16);
    unsigned s = 777u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) * (1.0f / 16777216.0f) * 2.f - 1.f; };
    for (int i = 0; i < 65536; i++) h[i] = make_float4(rnd(), rnd(), rnd(), rnd());
    (void)hipMemcpy(data, h, 65536 * 16, hipMemcpyHostToDevice);
    printf("{\"cases\": [\n");
    run<0, 8, true>("op_sel:[0,1,0] beside bf16 MFMA", launches, data, wrong, sink);
    run<0, 16, true>("op_sel:[0,1,0] beside bf16 MFMA", launches, data, wrong, sink);
    run<0, 16, false>("op_sel:[0,1,0], matrix waves idle", launches, data, wrong, sink);
    run<1, 16, true>("op_sel:[1,0,0] beside bf16 MFMA", launches, data, wrong, sink);
    run<2, 16, true>("op_sel:[0,0,1] beside bf16 MFMA", launches, data, wrong, sink);
    run<3, 16, true>("op_sel_hi:[1,0,1] beside bf16 MFMA", launches, data, wrong, sink);
    run<4, 16, true>("the kernel's own registers: v[4:5] += v[68:69] * v41, op_sel:[0,1,0], beside bf16 MFMA", launches, data, wrong, sink);
    run<4, 8, true>("the same", launches, data, wrong, sink);
    printf("  {}]}\n");
    return 0;
}
