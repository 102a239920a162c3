// probe: does the wave's TRAPSTS.EXCP accumulate IEEE exception bits (underflow = bit 4, inexact = bit 5, input denormal = bit 1) of
// VALU f32 operations on gfx950 with exceptions NOT enabled (MODE.EXCP_EN = 0), and does s_setreg clear them?  One wave, a few cases.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/trapsts_probe.hip -o scripts/ubench/trapsts_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__global__ void k(const float *in, unsigned *out, float *res) {
    const float a = in[0], b = in[1], c = in[2], d = in[3], e = in[4], f = in[5];
    unsigned t0, t1, t2, t3, t4, mode;
    float r1, r2, r3;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_MODE)" : "=s"(mode));
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_TRAPSTS, 0, 9), 0");
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS)" : "=s"(t0));
    asm volatile("v_fma_f32 %0, %1, %2, %3\n\ts_nop 7\n\ts_nop 7" : "=v"(r1) : "v"(a), "v"(b), "v"(c));       // normal operands, normal inexact result
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS)" : "=s"(t1));
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_TRAPSTS, 0, 9), 0");
    asm volatile("v_mul_f32 %0, %1, %2\n\ts_nop 7\n\ts_nop 7" : "=v"(r2) : "v"(d), "v"(e));                     // tiny x small: denormal inexact result -> underflow?
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS)" : "=s"(t2));
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_TRAPSTS, 0, 9), 0");
    asm volatile("v_mul_f32 %0, %1, %2\n\ts_nop 7\n\ts_nop 7" : "=v"(r3) : "v"(f), "v"(b));                     // denormal x 2.0: exact denormal result -> no underflow
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS)" : "=s"(t3));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS)" : "=s"(t4));
    if (threadIdx.x == 0) { out[0] = mode; out[1] = t0; out[2] = t1; out[3] = t2; out[4] = t3; out[5] = t4; res[0] = r1; res[1] = r2; res[2] = r3; }
}
int main() {
    float h[6] = {1.1f, 2.0f, 0.3f, 1e-30f, 1.3e-10f, 1e-40f};
    float *in, *res; unsigned *out;
    (void)hipMalloc(&in, 64); (void)hipMalloc(&out, 64); (void)hipMalloc(&res, 64);
    (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 64>>>(in, out, res);
    unsigned o[6]; float r[3];
    (void)hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost); (void)hipMemcpy(r, res, sizeof(r), hipMemcpyDeviceToHost);
    printf("MODE %08x (EXCP_EN = bits 12-20: %03x)\nTRAPSTS after clear %08x\nafter a normal inexact fma %08x (EXCP %03x)\nafter tiny x small (result %g) %08x (EXCP %03x)\nafter denormal x 2 exact (result %g) %08x (EXCP %03x)\n",
           o[0], (o[0] >> 12) & 0x1ff, o[1], o[2], o[2] & 0x1ff, r[1], o[3], o[3] & 0x1ff, r[2], o[4], o[4] & 0x1ff);
    return 0;
}
