// Does a packed v_pk_fma_f32 (two FP32 fused multiply-adds per lane) issue as fast as a plain v_fma_f32 on gfx950?
// Independent chains of each, 8 waves per SIMD; prints shader cycles per wave64 instruction per SIMD for both.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float float2v __attribute__((ext_vector_type(2)));

template <bool PACKED>
__global__ __launch_bounds__(256) void k_rate(float* out, int iters, float a, float b, unsigned long long* clocks) {
    const unsigned long long c0 = clock64();
    float2v x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = float2v{(float)(threadIdx.x + i), (float)(threadIdx.x - i)};
    const float2v A = {a, a}, B = {b, b};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (PACKED) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(A), "v"(B));
            else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i].x) : "v"(a), "v"(b));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += x[i].x + x[i].y;
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0) clocks[blockIdx.x] = clock64() - c0;
}

int main() {
    float* d; hipMalloc(&d, 4);
    unsigned long long* dc; hipMalloc(&dc, 4096 * sizeof(unsigned long long));
    static unsigned long long hc[4096];
    const int iters = 20000, waves = 8, blocks = 256 * waves;
    for (int packed = 0; packed < 2; packed++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (packed) hipLaunchKernelGGL(k_rate<true>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f, dc);
            else hipLaunchKernelGGL(k_rate<false>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f, dc);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        hipMemcpy(hc, dc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double sc = 0;
        for (int b = 0; b < blocks; b++) sc += (double)hc[b];
        printf("%s: %.3f shader cycles per wave64 instruction per SIMD (8 waves per SIMD), launch %.3f ms = %.1f T fused multiply-adds/s\n",
               packed ? "v_pk_fma_f32" : "v_fma_f32   ", (sc / blocks) / ((double)iters * 8 * waves), ms,
               (double)blocks * 256 * iters * 8 * (packed ? 2 : 1) / (ms * 1e-3) / 1e12);
    }
    return 0;
}
