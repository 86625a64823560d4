// Standalone reproducer for the packed-fp32 / bf16-MFMA observation (csrc/Makefile: -fno-slp-vectorize).
//
// Round 3 saw the fused deformation-field forward leave wrong values in lanes 48-63 of a few feature rows about one launch in ten
// when its gather arithmetic had been vectorised into v_pk_fma_f32 / v_pk_mul_f32 WHILE OTHER WAVES of the same SIMD issued
// v_mfma_f32_32x32x16_bf16.  This program isolates the ingredients: workgroups of 512 threads, two waves per SIMD (waves w and
// w + 4 share one).  "Matrix" waves 0-3 issue bf16 (or, mode 'f', f32) MFMAs in a loop; "vector" waves 4-7 evaluate a bilinear
// interpolation on float4 texels loaded from global memory -- the gather's arithmetic -- either with packed fp32 instructions
// (inline asm: v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32, mode P) or with scalar v_fma_f32 / v_mul_f32 (mode S), and store the
// result.  Every launch is compared bit for bit with a reference launch of the SAME vector code without matrix waves (mode
// without MFMAs: the matrix waves idle).  Output: wrong launches per 1000, wrong elements by lane and component.
//
//   hipcc --offload-arch=gfx950 -O3 tools/probe/pk_mfma_hazard.hip -o tools/probe/pk_mfma_hazard && tools/probe/pk_mfma_hazard [launches]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kBlocks = 256, kIters = 64, kTexels = 1 << 16;

__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b)
{
    f32x2 d;
    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c)
{
    f32x2 d;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// the forms the SLP vectoriser emits in the real kernel: one 32-bit register broadcast to both halves through op_sel_hi
// (src0 = {b, b} read from the LOW register of the pair `b2`: op_sel_hi:[0,1,1])
__device__ __forceinline__ f32x2 pk_fma_bcast0(f32x2 b2, f32x2 x, f32x2 c)
{
    f32x2 d;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "v"(b2), "v"(x), "v"(c));
    return d;
}
__device__ __forceinline__ f32x2 pk_mul_bcast1(f32x2 x, f32x2 b2)
{
    f32x2 d;
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(x), "v"(b2));
    return d;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b)
{
    f32x2 d;
    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b)
{
    f32x2 d;
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// mfma: 0 none (matrix waves idle), 1 bf16 32x32x16, 2 f32 32x32x2.  packed: the vector waves' arithmetic.
template <int MFMA, bool PACKED, bool OPSEL, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) probe(const float4* __restrict__ texels, const float2* __restrict__ frac, float4* __restrict__ out, float* sink)
{
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __shared__ bf16x8 s_w[4][64 * 8];
    if (wv < 4) {
        if (MFMA == 0) return;
        f32x16 acc0 = {0}, acc1 = {0};
        if (MFMA == 1) {
            bf16x8 a, b;
#pragma unroll
            for (int j = 0; j < 8; j++) { a[j] = (__bf16)(0.001f * (lane + j)); b[j] = (__bf16)1.0f; }
#pragma unroll
            for (int j = 0; j < 8; j++) s_w[wv][j * 64 + lane] = a;
            __builtin_amdgcn_wave_barrier();
            for (int i = 0; i < kIters * 12; i++) {
                const bf16x8 w = s_w[wv][(i & 7) * 64 + lane];          // the A operand comes out of LDS, as the weights do in the real kernel
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, w, acc1, 0, 0, 0);
            }
        } else {
            const float a = 0.001f * lane, b = 1.0f;
            for (int i = 0; i < kIters * 3; i++) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
            }
        }
        if (acc0[3] + acc1[7] == 12345.678f) sink[0] = acc0[0];          // keep the chain alive
        return;
    }
    // vector waves: kIters bilinear samples of four float4 texels each, as the HexPlane gather forms them
    const int vw = blockIdx.x * (WAVES - 4) + (wv - 4);
    for (int it = 0; it < kIters; it++) {
        const unsigned idx = (unsigned)(vw * kIters + it) * 64u + lane;
        const unsigned t0 = (idx * 2654435761u) >> 16;                   // 16-bit texel index
        const float4 q00 = texels[t0 & (kTexels - 1)], q01 = texels[(t0 + 1) & (kTexels - 1)];
        const float4 q10 = texels[(t0 + 257) & (kTexels - 1)], q11 = texels[(t0 + 258) & (kTexels - 1)];
        const float2 b = frac[idx & (kTexels - 1)];
        float4 r;
        if (PACKED && OPSEL) {
            // the same arithmetic with the broadcast forms (the fractions live in ONE register each, the low one of a pair)
            const f32x2 bxy = {b.x, b.y}, axy = {1.f - b.x, 1.f - b.y}, byx = {b.y, b.x}, ayx = {1.f - b.y, 1.f - b.x};
            const f32x2 a0 = {q00.x, q00.y}, a1 = {q00.z, q00.w}, b0 = {q01.x, q01.y}, b1 = {q01.z, q01.w};
            const f32x2 c0 = {q10.x, q10.y}, c1 = {q10.z, q10.w}, d0 = {q11.x, q11.y}, d1 = {q11.z, q11.w};
            const f32x2 top0 = pk_fma_bcast0(bxy, b0, pk_mul_bcast1(a0, axy)), top1 = pk_fma_bcast0(bxy, b1, pk_mul_bcast1(a1, axy));
            const f32x2 bot0 = pk_fma_bcast0(bxy, d0, pk_mul_bcast1(c0, axy)), bot1 = pk_fma_bcast0(bxy, d1, pk_mul_bcast1(c1, axy));
            const f32x2 v0 = pk_fma_bcast0(byx, bot0, pk_mul_bcast1(top0, ayx)), v1 = pk_fma_bcast0(byx, bot1, pk_mul_bcast1(top1, ayx));
            const f32x2 w0 = pk_add(pk_mul(v0, v0), top0), w1 = pk_add(pk_mul(v1, v1), top1);
            r = make_float4(w0.x, w0.y, w1.x, w1.y);
        } else if (PACKED) {
            const f32x2 bx = {b.x, b.x}, by = {b.y, b.y}, ax = {1.f - b.x, 1.f - b.x}, ay = {1.f - b.y, 1.f - b.y};
            const f32x2 a0 = {q00.x, q00.y}, a1 = {q00.z, q00.w}, b0 = {q01.x, q01.y}, b1 = {q01.z, q01.w};
            const f32x2 c0 = {q10.x, q10.y}, c1 = {q10.z, q10.w}, d0 = {q11.x, q11.y}, d1 = {q11.z, q11.w};
            const f32x2 top0 = pk_fma(b0, bx, pk_mul(a0, ax)), top1 = pk_fma(b1, bx, pk_mul(a1, ax));
            const f32x2 bot0 = pk_fma(d0, bx, pk_mul(c0, ax)), bot1 = pk_fma(d1, bx, pk_mul(c1, ax));
            const f32x2 v0 = pk_fma(bot0, by, pk_mul(top0, ay)), v1 = pk_fma(bot1, by, pk_mul(top1, ay));
            const f32x2 w0 = pk_add(pk_mul(v0, v0), top0), w1 = pk_add(pk_mul(v1, v1), top1);
            r = make_float4(w0.x, w0.y, w1.x, w1.y);
        } else {
            const float ax = 1.f - b.x, ay = 1.f - b.y;
            auto bil = [&](float a, float bb, float c, float d, float& top) {
                top = __builtin_fmaf(bb, b.x, a * ax);
                const float bot = __builtin_fmaf(d, b.x, c * ax);
                const float v = __builtin_fmaf(bot, b.y, top * ay);
                return v * v + top;
            };
            float t;
            r.x = bil(q00.x, q01.x, q10.x, q11.x, t);
            r.y = bil(q00.y, q01.y, q10.y, q11.y, t);
            r.z = bil(q00.z, q01.z, q10.z, q11.z, t);
            r.w = bil(q00.w, q01.w, q10.w, q11.w, t);
        }
        out[idx] = r;
    }
}

template <int MFMA, bool PACKED, bool OPSEL = false, int WAVES = 8>
static void launch(const float4* tex, const float2* frac, float4* out, float* sink)
{
    hipLaunchKernelGGL((probe<MFMA, PACKED, OPSEL, WAVES>), dim3(kBlocks), dim3(64 * WAVES), 0, 0, tex, frac, out, sink);
}

int main(int argc, char** argv)
{
    const int launches = argc > 1 ? atoi(argv[1]) : 1000;
    const size_t n = (size_t)kBlocks * 12 * kIters * 64;          // the largest case: twelve vector waves per workgroup
    std::vector<float4> h_tex(kTexels);
    std::vector<float2> h_frac(kTexels);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) * (1.0f / 16777216.0f); };
    for (auto& t : h_tex) t = make_float4(rnd() * 2.f - 1.f, rnd() * 2.f - 1.f, rnd() * 2.f - 1.f, rnd() * 2.f - 1.f);
    for (auto& f : h_frac) f = make_float2(rnd(), rnd());
    float4 *tex, *out, *ref;
    float2* frac;
    float* sink;
    (void)hipMalloc(&tex, kTexels * 16); (void)hipMalloc(&frac, kTexels * 8);
    (void)hipMalloc(&out, n * 16); (void)hipMalloc(&ref, n * 16); (void)hipMalloc(&sink, 4);
    (void)hipMemcpy(tex, h_tex.data(), kTexels * 16, hipMemcpyHostToDevice);
    (void)hipMemcpy(frac, h_frac.data(), kTexels * 8, hipMemcpyHostToDevice);
    std::vector<float4> h_out(n), h_ref(n);
    struct Case { const char* name; void (*ref_fn)(const float4*, const float2*, float4*, float*); void (*fn)(const float4*, const float2*, float4*, float*); };
    const Case cases[] = {
        {"packed fp32 beside bf16 MFMA (v_mfma_f32_32x32x16_bf16)", launch<0, true>, launch<1, true>},
        {"packed fp32 beside f32 MFMA  (v_mfma_f32_32x32x2_f32)  ", launch<0, true>, launch<2, true>},
        {"scalar fp32 beside bf16 MFMA                            ", launch<0, false>, launch<1, false>},
        {"packed fp32, matrix waves idle                          ", launch<0, true>, launch<0, true>},
        {"packed fp32 with op_sel broadcasts beside bf16 MFMA     ", launch<0, true, true>, launch<1, true, true>},
        {"the same, 16 waves per workgroup (1 matrix + 3 vector waves per SIMD, the real kernel's shape)", launch<0, true, true, 16>, launch<1, true, true, 16>},
    };
    printf("{\"launches_per_case\": %d, \"elements_per_launch\": %zu, \"cases\": [\n", launches, n * 4);
    for (size_t c = 0; c < sizeof(cases) / sizeof(cases[0]); c++) {
        (void)hipMemset(ref, 0xff, n * 16);
        cases[c].ref_fn(tex, frac, ref, sink);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h_ref.data(), ref, n * 16, hipMemcpyDeviceToHost);
        long wrong_launches = 0, wrong_elems = 0, by_lane_quarter[4] = {0, 0, 0, 0}, by_comp[4] = {0, 0, 0, 0};
        for (int l = 0; l < launches; l++) {
            (void)hipMemsetAsync(out, 0xff, n * 16, 0);
            cases[c].fn(tex, frac, out, sink);
            (void)hipMemcpy(h_out.data(), out, n * 16, hipMemcpyDeviceToHost);
            if (memcmp(h_out.data(), h_ref.data(), n * 16) == 0) continue;
            wrong_launches++;
            for (size_t i = 0; i < n; i++) {
                const unsigned* a = reinterpret_cast<const unsigned*>(&h_out[i]);
                const unsigned* b = reinterpret_cast<const unsigned*>(&h_ref[i]);
                for (int k = 0; k < 4; k++)
                    if (a[k] != b[k]) { wrong_elems++; by_lane_quarter[(i & 63) >> 4]++; by_comp[k]++; }
            }
        }
        printf("  {\"case\": \"%s\", \"wrong_launches\": %ld, \"wrong_elements\": %ld, \"by_lanes_0_15_16_31_32_47_48_63\": [%ld, %ld, %ld, %ld], "
               "\"by_component_xyzw\": [%ld, %ld, %ld, %ld]}%s\n", cases[c].name, wrong_launches, wrong_elems, by_lane_quarter[0], by_lane_quarter[1],
               by_lane_quarter[2], by_lane_quarter[3], by_comp[0], by_comp[1], by_comp[2], by_comp[3], c + 1 < sizeof(cases) / sizeof(cases[0]) ? "," : "");
    }
    printf("]}\n");
    return 0;
}
