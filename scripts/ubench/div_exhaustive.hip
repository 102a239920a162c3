// Exhaustive check (gfx950): is the 3-instruction quotient
//     q0 = n*y;  r = fma(-d, q0, n);  q = fma(r, y, q0)        with  y = RN(1/d)  (IEEE 1.0f/d)
// bit-identical to the IEEE-754 correctly rounded n/d (hipcc's `/` with
// -fhip-fp32-correctly-rounded-divide-sqrt) for EVERY pair of f32 significands?
// d ranges over all 2^23 significands in [1,2); n over all 2^24 values in [1,4) (two binades, so the
// quotient covers (0.5, 4)).  Scaling n or d by a power of two scales every intermediate exactly (absent
// under/overflow), so agreement here is agreement for all normal-range operands.  Also checks the 5-op
// tail of hipcc's own expansion with the refined (not correctly rounded) reciprocal.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k(unsigned d_begin, unsigned d_count, unsigned long long *bad3, unsigned long long *bad5, unsigned *example) {
    const unsigned di = d_begin + blockIdx.x;
    if (blockIdx.x >= d_count) return;
    const float d = __uint_as_float(0x3f800000u | di);
    const float y = 1.0f / d;                               // correctly rounded reciprocal
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float y5 = __builtin_fmaf(__builtin_fmaf(-d, r0, 1.0f), r0, r0);   // hipcc's refined reciprocal
    unsigned long long b3 = 0, b5 = 0;
    for (unsigned ni = threadIdx.x; ni < (1u << 24); ni += blockDim.x) {
        const float n = __uint_as_float(0x3f800000u + ni);  // [1,4): exponent field increments at 2^23
        const float q = n / d;
        const float q0 = n * y;
        const float r = __builtin_fmaf(-d, q0, n);
        const float q3 = __builtin_fmaf(r, y, q0);
        const float m = n * y5;
        const float f2 = __builtin_fmaf(-d, m, n);
        const float f3 = __builtin_fmaf(f2, y5, m);
        const float f4 = __builtin_fmaf(-d, f3, n);
        const float q5 = __builtin_fmaf(f4, y5, f3);
        if (__float_as_uint(q3) != __float_as_uint(q)) { b3++; if (atomicAdd(&example[0], 1u) == 0) { example[1] = __float_as_uint(n); example[2] = __float_as_uint(d); } }
        if (__float_as_uint(q5) != __float_as_uint(q)) b5++;
    }
    if (b3) atomicAdd(bad3, b3);
    if (b5) atomicAdd(bad5, b5);
}
int main(int argc, char **argv) {
    const unsigned total = argc > 1 ? (unsigned)atol(argv[1]) : (1u << 23);     // number of divisor significands to cover
    unsigned long long *bad3, *bad5; unsigned *ex;
    hipMalloc(&bad3, 8); hipMalloc(&bad5, 8); hipMalloc(&ex, 16); hipMemset(bad3, 0, 8); hipMemset(bad5, 0, 8); hipMemset(ex, 0, 16);
    const unsigned chunk = 1u << 16;
    for (unsigned b = 0; b < total; b += chunk) {
        const unsigned cnt = total - b < chunk ? total - b : chunk;
        // stride the chunks over the whole significand range when only a subset is requested
        const unsigned begin = total == (1u << 23) ? b : (unsigned)(((unsigned long long)b << 23) / total);
        k<<<cnt, 256>>>(begin, cnt, bad3, bad5, ex);
        hipDeviceSynchronize();
        if ((b / chunk) % 8 == 0) { printf("progress %u / %u divisors\n", b + cnt, total); fflush(stdout); }
    }
    unsigned long long h3, h5; unsigned hex[4];
    hipMemcpy(&h3, bad3, 8, hipMemcpyDeviceToHost); hipMemcpy(&h5, bad5, 8, hipMemcpyDeviceToHost); hipMemcpy(hex, ex, 16, hipMemcpyDeviceToHost);
    printf("divisors %u x numerators 2^24: mismatches 3-op (correctly rounded reciprocal) = %llu, 5-op (refined reciprocal) = %llu\n", total, h3, h5);
    if (h3) printf("first 3-op mismatch: n = 0x%08x d = 0x%08x\n", hex[1], hex[2]);
    return 0;
}
