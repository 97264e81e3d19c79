// Does MODE.FP16_OVFL (bit 23 of HW_REG_MODE, GFX9 ISA) clamp v_cvt_pk_f16_f32 on gfx950?
// With the bit set "an overflowed FP16 result is clamped to +/-MAX_FP16 regardless of round mode, while still preserving
// true INF values" — if it applies to the packed conversion, the fused kernel's fp16 snapshot can drop its saturating
// v_pk_min_i16 (one VALU per converted pair).  Prints the fp16 bit patterns of a few conversions with the bit clear and set.
// Build: hipcc -O3 --offload-arch=gfx950 -o ovfl_ubench_test ovfl_ubench.hip ; run once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

template <int OVFL>
__global__ void k(const float* in, unsigned* out, int n) {
    int i = threadIdx.x;
    if (i >= n) return;
    float a = in[i], b = -in[i];
    unsigned r;
    if (OVFL)
        asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1\n\ts_nop 3\n\t"
                     "v_cvt_pk_f16_f32 %0, %1, %2\n\ts_nop 3\n\t"
                     "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 0" : "=v"(r) : "v"(a), "v"(b));
    else
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    out[i] = r;
}

int main() {
    const float h[8] = {1.0f, 65504.0f, 65519.0f, 65520.0f, 1.0e6f, 3.0e38f, INFINITY, NAN};
    float* d_in; unsigned* d_out; unsigned o[2][8];
    hipMalloc(&d_in, sizeof(h)); hipMalloc(&d_out, 8 * sizeof(unsigned));
    hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d_in, d_out, 8);
    hipMemcpy(o[0], d_out, sizeof(o[0]), hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d_in, d_out, 8);
    hipMemcpy(o[1], d_out, sizeof(o[1]), hipMemcpyDeviceToHost);
    printf("%14s  %-22s %-22s   (lo = cvt(x), hi = cvt(-x); 0x7bff = 65504, 0x7c00 = inf)\n", "x", "FP16_OVFL=0", "FP16_OVFL=1");
    for (int i = 0; i < 8; ++i)
        printf("%14g  lo %04x hi %04x        lo %04x hi %04x\n", h[i], o[0][i] & 0xffff, o[0][i] >> 16, o[1][i] & 0xffff, o[1][i] >> 16);
    return 0;
}
