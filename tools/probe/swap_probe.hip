// Semantics of v_permlane16_swap / v_permlane32_swap and of DPP adds with a partial bank mask (gfx950), printed per lane.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/swap_probe.hip -o /tmp/swap_probe && /tmp/swap_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(float* out)
{
    const int lane = threadIdx.x;
    const float a = 100.f + lane, b = 200.f + lane;
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    out[lane] = __uint_as_float(r[0]);
    out[64 + lane] = __uint_as_float(r[1]);
    const auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    out[128 + lane] = __uint_as_float(q[0]);
    out[192 + lane] = __uint_as_float(q[1]);
    float d = -1.f, x = 1000.f + lane, y = 2000.f + lane, e = -1.f, f = -1.f;
    asm volatile("s_nop 4\n"
                 "v_add_f32_dpp %0, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0x5\n"
                 "v_add_f32_dpp %0, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xa\n"
                 "v_add_f32_dpp %1, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0x3\n"
                 "v_add_f32_dpp %1, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xc\n"
                 "v_add_f32_dpp %2, %3, %4 row_ror:4 row_mask:0xf bank_mask:0xf\n"
                 : "+v"(d), "+v"(e), "+v"(f) : "v"(x), "v"(y));
    out[256 + lane] = d;
    out[320 + lane] = e;
    out[384 + lane] = f;
}
int main()
{
    float* out; float h[448];
    (void)hipMalloc(&out, sizeof(h));
    k<<<1, 64>>>(out);
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"swap16 r[0] (a=100+l,b=200+l)", "swap16 r[1]", "swap32 r[0]", "swap32 r[1]", "qp1 bank5:x bankA:y (x=1000+l,y=2000+l)", "qp2 bank3:x bankC:y", "ror4: dpp(x)+y"};
    for (int t = 0; t < 7; t++) {
        printf("%s\n", names[t]);
        for (int l = 0; l < 64; l++) printf("%6.0f%s", h[t * 64 + l], (l & 15) == 15 ? "\n" : " ");
    }
    return 0;
}
