// Issue-rate probe for gfx950: what the 32-bit integer multiplies of the Philox rounds cost beside v_fma_f32.
// Independent chains, 4 waves per SIMD; prints nominal 2.4 GHz cycles per wave64 instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ __launch_bounds__(256) void k_op(unsigned* out, int iters, unsigned a, unsigned b) {
    unsigned x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 2654435761u + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) x[i] = __float_as_uint(__builtin_fmaf(__uint_as_float(x[i]), __uint_as_float(a), __uint_as_float(b)));
            if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            if (OP == 3) { unsigned long long r; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(r) : "v"(x[i]), "v"(a) : "vcc"); x[i] = (unsigned)r ^ (unsigned)(r >> 32); }
            if (OP == 4) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            if (OP == 5) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[i]) : "v"(a));
        }
    }
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s ^= x[i];
    if (s == 0x12345678u) out[0] = s;
}

template <int OP> void run(const char* name, unsigned* d, int per_iter) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000, blocks = 256 * 4;
    hipLaunchKernelGGL(k_op<OP>, dim3(blocks), dim3(256), 0, 0, d, 100, 0x3f800100u, 0x3f000000u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_op<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 0x3f800100u, 0x3f000000u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0.0f; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %.3f ms  %.2f cycles per wave64 instruction per SIMD (%d instruction(s) per chain step)\n", name, ms,
           ms * 1e-3 * 2.4e9 / ((double)iters * 8 * 4 * per_iter), per_iter);
}

int main() {
    unsigned* d = nullptr; if (hipMalloc(&d, 4) != hipSuccess) return 1;
    run<0>("v_fma_f32", d, 1);
    run<4>("v_xor_b32", d, 1);
    run<1>("v_mul_lo_u32", d, 1);
    run<2>("v_mul_hi_u32", d, 1);
    run<5>("v_mul_u32_u24", d, 1);
    run<3>("v_mad_u64_u32 (+ v_xor_b32)", d, 2);
    return 0;
}
