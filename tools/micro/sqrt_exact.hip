// Exhaustive check: which short square-root sequences return the bits of the correctly rounded
// sqrtf(x) for every float x >= 0 (and what they do elsewhere)?
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 tools/micro/sqrt_exact.hip -o sqrt_exact
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

constexpr int kCand = 5;

__device__ __forceinline__ bool same(float a, float b) {
    if (a != a && b != b) return true;
    return __float_as_uint(a) == __float_as_uint(b);
}

__global__ void k_check(uint32_t lo, uint32_t hi, unsigned long long* bad, uint32_t* example, uint32_t* min_exp, uint32_t* max_exp) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long local[kCand] = {};
    uint32_t lmn[kCand], lmx[kCand], lex[kCand];
    for (int k = 0; k < kCand; k++) { lmn[k] = 0xFFFFFFFFu; lmx[k] = 0; lex[k] = 0; }
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 31); i += stride) {   // sign bit clear
        const float x = __uint_as_float((uint32_t)i);
        const float ref = sqrtf(x);
        float c[kCand];
        const float s0 = __builtin_amdgcn_sqrtf(x);
        const float y = __builtin_amdgcn_rsqf(x);
        const float g = x * y, h = 0.5f * y;
        const float g1 = fmaf(fmaf(-g, g, x), h, g);
        const float g2 = fmaf(fmaf(-g1, g1, x), h, g1);
        c[0] = s0;
        c[1] = g1;
        c[2] = g2;
        // hardware square root + one correction with half its reciprocal estimate
        const float hh = 0.5f * __builtin_amdgcn_rcpf(s0);
        c[3] = fmaf(fmaf(-s0, s0, x), hh, s0);
        // hardware square root + one correction with h from the reciprocal square root
        c[4] = fmaf(fmaf(-s0, s0, x), h, s0);
        const uint32_t ex = ((uint32_t)i >> 23) & 0xFFu;
        for (int k = 0; k < kCand; k++) {
            if (!same(c[k], ref) && ex >= lo && ex <= hi) {
                local[k]++;
                lmn[k] = ex < lmn[k] ? ex : lmn[k];
                lmx[k] = ex > lmx[k] ? ex : lmx[k];
                lex[k] = (uint32_t)i;
            }
        }
    }
    for (int k = 0; k < kCand; k++) {
        if (!local[k]) continue;
        atomicAdd(&bad[k], local[k]);
        atomicMin(&min_exp[k], lmn[k]);
        atomicMax(&max_exp[k], lmx[k]);
        if (lex[k]) example[k] = lex[k];
    }
}

int main(int argc, char** argv) {
    unsigned long long* bad; uint32_t *example, *mn, *mx;
    (void)hipMalloc(&bad, kCand * 8); (void)hipMalloc(&example, kCand * 4); (void)hipMalloc(&mn, kCand * 4); (void)hipMalloc(&mx, kCand * 4);
    (void)hipMemset(bad, 0, kCand * 8); (void)hipMemset(example, 0, kCand * 4); (void)hipMemset(mn, 0xFF, kCand * 4); (void)hipMemset(mx, 0, kCand * 4);
    // mismatches are counted for biased exponents in [lo, hi] (arguments; default: all)
    const uint32_t lo = argc > 2 ? (uint32_t)atoi(argv[1]) : 0u, hi = argc > 2 ? (uint32_t)atoi(argv[2]) : 255u;
    hipLaunchKernelGGL(k_check, dim3(256 * 32), dim3(256), 0, 0, lo, hi, bad, example, mn, mx);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    unsigned long long hb[kCand]; uint32_t he[kCand], hmn[kCand], hmx[kCand];
    (void)hipMemcpy(hb, bad, sizeof hb, hipMemcpyDeviceToHost); (void)hipMemcpy(he, example, sizeof he, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hmn, mn, sizeof hmn, hipMemcpyDeviceToHost); (void)hipMemcpy(hmx, mx, sizeof hmx, hipMemcpyDeviceToHost);
    const char* names[kCand] = {"sqrt", "rsq: g + (x - g*g) * y/2", "the same, twice", "sqrt + (x - s*s) * rcp(s)/2", "sqrt + (x - s*s) * rsq(x)/2"};
    for (int k = 0; k < kCand; k++) {
        float x; std::memcpy(&x, &he[k], 4);
        printf("%-32s mismatches %12llu  biased exponent range [%u, %u]  e.g. x = %a (0x%08x)\n", names[k], hb[k], hmn[k], hmx[k], x, he[k]);
    }
    return 0;
}
