// What does one wave64 vector instruction cost the SIMD, by instruction class and by the number of waves sharing the SIMD?
// The compositing backward (csrc/raster_render.hip) is bound by vector-instruction issue; its budget is only as good as the
// per-class prices.  Every workgroup is 256 threads (one wave per SIMD); `k` workgroups per CU give k waves per SIMD.
// Each wave runs `iters` x 64 instructions of one class on 8 independent register chains and stamps s_memtime around the loop;
// the figure printed is  SIMD cycles per wave-instruction = (median wave's cycles) / (64 iters) / k ... per SIMD, i.e. the
// rate at which the SIMD retires that class when k waves offer it.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define R8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define BODY8(INS) asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                                : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) \
                                : "v"(c), "s"(sidx), "v"(c2) : "vcc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47")
#define BODY64(INS) BODY8(INS); BODY8(INS); BODY8(INS); BODY8(INS); BODY8(INS); BODY8(INS); BODY8(INS); BODY8(INS)

#define I_FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %8\n"
#define I_MUL(n) "v_mul_f32 %" #n ", %" #n ", %8\n"
#define I_ADD(n) "v_add_f32 %" #n ", %" #n ", %8\n"
#define I_MOV(n) "v_mov_b32 %" #n ", %8\n"
#define I_EXP(n) "v_exp_f32 %" #n ", %" #n "\n"
#define I_RCP(n) "v_rcp_f32 %" #n ", %" #n "\n"
#define I_DPPQ(n) "v_add_f32_dpp %" #n ", %" #n ", %" #n " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define I_DPPR(n) "v_add_f32_dpp %" #n ", %" #n ", %" #n " row_ror:4 row_mask:0xf bank_mask:0xf\n"
#define I_DPPB(n) "v_add_f32_dpp %" #n ", %" #n ", %" #n " row_bcast:15 row_mask:0xa bank_mask:0xf\n"
#define I_CND(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define I_CMP(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n"
#define I_RDL(n) "v_readlane_b32 s4" #n ", %" #n ", %9\n"
#define I_SWAP32(n) "v_permlane32_swap_b32 %" #n ", %" #n "\n"
#define I_SWAP16(n) "v_permlane16_swap_b32 %" #n ", %" #n "\n"
#define I_CND64(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, s[40:41]\n"
#define I_CNDV(n) "v_cndmask_b32 %" #n ", %8, %" #n ", vcc\n"
#define I_FMAC(n) "v_fmac_f32 %" #n ", %8, %8\n"
#define I_FMA3(n) "v_fma_f32 %" #n ", %8, %10, %" #n "\n"
#define I_MULS(n) "v_mul_f32 %" #n ", s42, %" #n "\n"
#define I_MAX(n) "v_max_f32 %" #n ", %" #n ", %8\n"
#define I_SUB(n) "v_sub_f32 %" #n ", %" #n ", %8\n"
#define I_MULLIT(n) "v_mul_f32 %" #n ", 0x3fb8aa3b, %" #n "\n"
#define I_AND(n) "v_and_b32 %" #n ", %" #n ", %8\n"
#define I_PKFMA(n) "v_pk_fma_f32 %" #n ", %" #n ", %8, %8\n"      // 64-bit operands: uses a pair variant below

template <int KIND>
__global__ void __launch_bounds__(256) k_rate(int iters, unsigned long long* cyc, float* out)
{
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; j++) x[j] = 1.0f + 1e-3f * (threadIdx.x + j);
    const float c = 1.0001f, c2 = 0.999f + 1e-6f * threadIdx.x;
    const int sidx = 3;
    asm volatile("s_mov_b32 s40, 0x55555555\n s_mov_b32 s41, 0x55555555\n s_mov_b32 s42, 0x3f800347\n s_mov_b32 vcc_lo, 0x33333333\n s_mov_b32 vcc_hi, 0x33333333" ::: "s40", "s41", "s42", "vcc");
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) { BODY64(I_FMA); }
        if (KIND == 1) { BODY64(I_MUL); }
        if (KIND == 2) { BODY64(I_ADD); }
        if (KIND == 3) { BODY64(I_MOV); }
        if (KIND == 4) { BODY64(I_EXP); }
        if (KIND == 5) { BODY64(I_RCP); }
        if (KIND == 6) { BODY64(I_DPPQ); }
        if (KIND == 7) { BODY64(I_DPPR); }
        if (KIND == 8) { BODY64(I_DPPB); }
        if (KIND == 9) { BODY64(I_CND); }
        if (KIND == 10) { BODY64(I_CMP); }
        if (KIND == 11) { BODY64(I_RDL); }
        if (KIND == 12) { BODY64(I_SWAP32); }
        if (KIND == 13) { BODY64(I_SWAP16); }
        if (KIND == 14) { BODY64(I_CND64); }
        if (KIND == 15) { BODY64(I_CNDV); }
        if (KIND == 16) { BODY64(I_FMAC); }
        if (KIND == 17) { BODY64(I_FMA3); }
        if (KIND == 18) { BODY64(I_MULS); }
        if (KIND == 19) { BODY64(I_MAX); }
        if (KIND == 20) { BODY64(I_SUB); }
        if (KIND == 21) { BODY64(I_MULLIT); }
        if (KIND == 22) { BODY64(I_AND); }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j++) r += x[j];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// Packed fp32 (two floats per lane in a 64-bit register pair): what does one v_pk_* wave64 instruction cost beside the plain forms?
// (No MFMAs in the compositing kernels, so the packed-fp32 / MFMA observation of deform_field.hip does not apply there.)
typedef float v2f __attribute__((ext_vector_type(2)));
#define PBODY8(INS) asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                                : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) \
                                : "v"(c), "v"(c2), "s"(sc))
#define PBODY64(INS) PBODY8(INS); PBODY8(INS); PBODY8(INS); PBODY8(INS); PBODY8(INS); PBODY8(INS); PBODY8(INS); PBODY8(INS)
#define P_FMA(n) "v_pk_fma_f32 %" #n ", %" #n ", %8, %8\n"
#define P_FMA3(n) "v_pk_fma_f32 %" #n ", %8, %9, %" #n "\n"
#define P_MUL(n) "v_pk_mul_f32 %" #n ", %" #n ", %8\n"
#define P_ADD(n) "v_pk_add_f32 %" #n ", %" #n ", %8\n"
#define P_MULB(n) "v_pk_mul_f32 %" #n ", %" #n ", %8 op_sel_hi:[1,0]\n"      // second operand's low half broadcast to both
#define P_FMAS(n) "v_pk_fma_f32 %" #n ", %" #n ", %10, %10\n"               // SGPR-pair operand
#define P_MOV(n) "v_pk_mov_b32 %" #n ", %8, %8\n"
template <int KIND>
__global__ void __launch_bounds__(256) k_rate_pk(int iters, unsigned long long* cyc, float* out)
{
    v2f x[8];
#pragma unroll
    for (int j = 0; j < 8; j++) x[j] = v2f{1.0f + 1e-3f * (threadIdx.x + j), 1.0f - 1e-3f * j};
    const v2f c = {1.0001f, 0.9999f}, c2 = {0.999f + 1e-6f * threadIdx.x, 1.001f};
    const unsigned long long sc = 0x3f8003473f800347ull;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) { PBODY64(P_FMA); }
        if (KIND == 1) { PBODY64(P_FMA3); }
        if (KIND == 2) { PBODY64(P_MUL); }
        if (KIND == 3) { PBODY64(P_ADD); }
        if (KIND == 4) { PBODY64(P_MULB); }
        if (KIND == 5) { PBODY64(P_FMAS); }
        if (KIND == 6) { PBODY64(P_MOV); }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j++) r += x[j].x + x[j].y;
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// A mix as the compositing loops have it: of every 8 instructions, `NPK` packed and the rest plain v_fma_f32 -- is a packed
// instruction's price the same inside a stream of plain ones?
template <int KIND>
__global__ void __launch_bounds__(256) k_rate_mix(int iters, unsigned long long* cyc, float* out)
{
    v2f x[4];
    float y[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { x[j] = v2f{1.0f + 1e-3f * (threadIdx.x + j), 1.0f - 1e-3f * j}; y[j] = 1.0f + 1e-4f * j; }
    const v2f c = {1.0001f, 0.9999f};
    const float d = 1.0001f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (KIND == 0)      // 4 packed fma + 4 plain fma
                asm volatile("v_pk_fma_f32 %0, %0, %8, %8\nv_fma_f32 %4, %4, %9, %9\nv_pk_fma_f32 %1, %1, %8, %8\nv_fma_f32 %5, %5, %9, %9\n"
                             "v_pk_fma_f32 %2, %2, %8, %8\nv_fma_f32 %6, %6, %9, %9\nv_pk_fma_f32 %3, %3, %8, %8\nv_fma_f32 %7, %7, %9, %9\n"
                             : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]) : "v"(c), "v"(d));
            else                // 8 plain fma on the same 12 registers (the unpacked equivalent of the 4 packed: 8 + 4 = 12 plain)
                asm volatile("v_fma_f32 %0, %0, %8, %8\nv_fma_f32 %4, %4, %8, %8\nv_fma_f32 %1, %1, %8, %8\nv_fma_f32 %5, %5, %8, %8\n"
                             "v_fma_f32 %2, %2, %8, %8\nv_fma_f32 %6, %6, %8, %8\nv_fma_f32 %3, %3, %8, %8\nv_fma_f32 %7, %7, %8, %8\n"
                             : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(x[0].x), "+v"(x[1].x), "+v"(x[2].x), "+v"(x[3].x) : "v"(d));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) r += x[j].x + x[j].y + y[j];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// LDS broadcast reads (every lane the same address): the compositing loops read their splat records this way
template <int WIDTH>
__global__ void __launch_bounds__(256) k_lds(int iters, unsigned long long* cyc, float* out)
{
    __shared__ float4 s[512];
    for (int i = threadIdx.x; i < 512; i += 256) s[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    float acc = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 64; u++) {
            const int j = (i * 64 + u) & 511;
            if (WIDTH == 16) { const float4 v = s[j]; acc += v.x + v.w; }
            else { acc += reinterpret_cast<const float*>(s)[j * 4]; }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <typename F>
static void run(const char* name, F launch, unsigned long long* cyc_d, int iters)
{
    printf("%-28s", name);
    for (int k : {1, 2, 3, 4, 5, 6, 8}) {
        const int blocks = 256 * k;
        launch(blocks, iters / 8);
        (void)hipDeviceSynchronize();
        hipEvent_t a, b;
        (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        (void)hipEventRecord(a);
        launch(blocks, iters);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        std::vector<unsigned long long> h(blocks * 4);
        (void)hipMemcpy(h.data(), cyc_d, h.size() * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double med = (double)h[h.size() / 2];
        // s_memtime ticks at 100 MHz on this part?  print both: ticks per instruction and wall ns per instruction per SIMD
        const double n_inst = 64.0 * iters;
        printf("  k=%d: %6.2f tick/inst/wave %6.3f ns/inst/SIMD", k, med / n_inst, ms * 1e6 / (n_inst * k));
    }
    printf("\n");
}

int main()
{
    unsigned long long* cyc;
    float* out;
    (void)hipMalloc(&cyc, 256 * 8 * 4 * 8);
    (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    const int iters = 4096;
#define RUNK(name, K) run(name, [&](int blocks, int it) { hipLaunchKernelGGL(k_rate<K>, dim3(blocks), dim3(256), 0, 0, it, cyc, out); }, cyc, iters)
    RUNK("v_fma_f32", 0);
    RUNK("v_mul_f32", 1);
    RUNK("v_add_f32", 2);
    RUNK("v_mov_b32", 3);
    RUNK("v_exp_f32", 4);
    RUNK("v_rcp_f32", 5);
    RUNK("v_add_f32 dpp quad_perm", 6);
    RUNK("v_add_f32 dpp row_ror", 7);
    RUNK("v_add_f32 dpp row_bcast", 8);
    RUNK("v_cndmask_b32", 9);
    RUNK("v_cmp_lt_f32", 10);
    RUNK("v_readlane_b32", 11);
    RUNK("v_permlane32_swap", 12);
    RUNK("v_permlane16_swap", 13);
    RUNK("v_cndmask_e64 s[40:41]", 14);
    RUNK("v_cndmask vcc (src swapped)", 15);
    RUNK("v_fmac_f32 (VOP2)", 16);
    RUNK("v_fma_f32 3 distinct regs", 17);
    RUNK("v_mul_f32 sgpr operand", 18);
    RUNK("v_max_f32", 19);
    RUNK("v_sub_f32", 20);
    RUNK("v_mul_f32 literal", 21);
    RUNK("v_and_b32", 22);
#define RUNP(name, K) run(name, [&](int blocks, int it) { hipLaunchKernelGGL(k_rate_pk<K>, dim3(blocks), dim3(256), 0, 0, it, cyc, out); }, cyc, iters)
    RUNP("v_pk_fma_f32 (x,c,c)", 0);
    RUNP("v_pk_fma_f32 3 distinct", 1);
    RUNP("v_pk_mul_f32", 2);
    RUNP("v_pk_add_f32", 3);
    RUNP("v_pk_mul_f32 op_sel_hi bcast", 4);
    RUNP("v_pk_fma_f32 sgpr pair", 5);
    RUNP("v_pk_mov_b32", 6);
    run("mix 4 pk_fma + 4 fma (per 8)", [&](int blocks, int it) { hipLaunchKernelGGL(k_rate_mix<0>, dim3(blocks), dim3(256), 0, 0, it, cyc, out); }, cyc, iters);
    run("8 plain fma (per 8)", [&](int blocks, int it) { hipLaunchKernelGGL(k_rate_mix<1>, dim3(blocks), dim3(256), 0, 0, it, cyc, out); }, cyc, iters);
    run("ds_read_b128 broadcast", [&](int blocks, int it) { hipLaunchKernelGGL(k_lds<16>, dim3(blocks), dim3(256), 0, 0, it, cyc, out); }, cyc, iters);
    run("ds_read_b32 broadcast", [&](int blocks, int it) { hipLaunchKernelGGL(k_lds<4>, dim3(blocks), dim3(256), 0, 0, it, cyc, out); }, cyc, iters);
    return 0;
}
