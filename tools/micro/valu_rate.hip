// VALU issue-rate probe for gfx950: independent v_fma_f32 chains, waves per SIMD swept by the
// launch size.  Prints wave64 VALU instructions per cycle per SIMD at a nominal 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>

// clocks[2 * block], [2 * block + 1]: shader-clock cycles (s_memtime) and constant 100 MHz ticks
// (s_memrealtime) the block's first wave spent in its loop.  Their ratio is the clock the SIMDs really
// ran at under this load, which the nominal 2.4 GHz of the host-side figure is not.
template <int CHAINS>
__global__ __launch_bounds__(256) void k_fma(float* out, int iters, float a, float b, unsigned long long* clocks) {
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    float x[CHAINS];
#pragma unroll
    for (int i = 0; i < CHAINS; i++) x[i] = (float)(threadIdx.x + i);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i++) x[i] = __builtin_fmaf(x[i], a, b);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; i++) s += x[i];
    if (s == 12345.678f) out[0] = s;
    if (clocks && threadIdx.x == 0) { clocks[2 * blockIdx.x] = clock64() - c0; clocks[2 * blockIdx.x + 1] = wall_clock64() - w0; }
}

int main() {
    float* d; hipMalloc(&d, 4);
    unsigned long long* dc; hipMalloc(&dc, 2 * 2048 * sizeof(unsigned long long));
    static unsigned long long hc[2 * 2048];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int waves_per_simd = 1; waves_per_simd <= 8; waves_per_simd *= 2) {
        // 256 CUs x 4 SIMDs; a block of 256 threads = 4 waves = one wave per SIMD of a CU
        const int blocks = 256 * waves_per_simd;
        hipLaunchKernelGGL(k_fma<8>, dim3(blocks), dim3(256), 0, 0, d, 100, 1.0001f, 0.5f, (unsigned long long*)nullptr);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_fma<8>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f, dc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)iters * 8 * waves_per_simd;   // wave-instructions per SIMD (if blocks spread evenly)
        const double cycles = ms * 1e-3 * 2.4e9;
        printf("waves/SIMD %d: %.3f ms, %.3f cycles per wave64 v_fma per SIMD, %.1f TFLOP/s\n", waves_per_simd, ms,
               cycles / instr_per_simd, (double)blocks * 256 * iters * 8 * 2 / (ms * 1e-3) / 1e12);
        hipMemcpy(hc, dc, 2 * blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double sc = 0, rt = 0;
        for (int b = 0; b < blocks; b++) { sc += (double)hc[2 * b]; rt += (double)hc[2 * b + 1]; }
        // a wave issues iters * 8 instructions while waves_per_simd waves share its SIMD
        printf("    in-kernel: shader clock %.0f MHz under this load (s_memtime / s_memrealtime x 100 MHz), %.3f shader cycles per wave64 v_fma per SIMD\n",
               sc / rt * 100.0, (sc / blocks) / ((double)iters * 8 * waves_per_simd));
    }
    return 0;
}
