// Exhaustive check: which short reciprocal sequences return the bits of the IEEE division 1.0f / x
// for every float x that the triangle test can divide by (everything except |x| < 1e-6)?
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 tools/micro/rcp_exact.hip -o rcp_exact
//   ./rcp_exact        divisors of the triangle test (|x| >= 1e-6, NaN and infinities included)
//   ./rcp_exact all    every float, mismatches counted for biased exponents below 200
// Result on MI355X: "rcp+newton" (v_rcp_f32, then r + r * fma(-x, r, 1)) differs from the division
// exactly for biased exponent 0 (zero, denormals) and for 253, 254, 255 (|x| >= 2^126, infinities):
// for all 2 x 252 x 2^23 floats with 2^-126 <= |x| < 2^126 it returns the same bits.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

constexpr int kCand = 6;

__device__ __forceinline__ bool same(float a, float b) {
    if (a != a && b != b) return true;
    return __float_as_uint(a) == __float_as_uint(b);
}

__global__ void k_check(int all, unsigned long long* bad, uint32_t* example, uint32_t* min_exp, uint32_t* max_exp) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long local[kCand] = {};
    uint32_t lmn[kCand], lmx[kCand], lex[kCand];
    for (int k = 0; k < kCand; k++) { lmn[k] = 0xFFFFFFFFu; lmx[k] = 0; lex[k] = 0; }
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const float x = __uint_as_float((uint32_t)i);
        if (!all && fabsf(x) < 1e-6f) continue;
        const float ref = 1.0f / x;
        float c[kCand];
        const float r0 = __builtin_amdgcn_rcpf(x);
        const float e0 = fmaf(-x, r0, 1.0f);
        const float r1 = fmaf(e0, r0, r0);
        const float e1 = fmaf(-x, r1, 1.0f);
        const float r2 = fmaf(e1, r1, r1);
        c[0] = r0;
        c[1] = r1;
        c[2] = __builtin_amdgcn_div_fixupf(r1, x, 1.0f);
        c[3] = r2;
        c[4] = __builtin_amdgcn_div_fixupf(r2, x, 1.0f);
        // residual against the first refinement, correction with the raw estimate
        c[5] = __builtin_amdgcn_div_fixupf(fmaf(e1, r0, r1), x, 1.0f);
        const uint32_t ex = ((uint32_t)i >> 23) & 0xFFu;
        for (int k = 0; k < kCand; k++) {
            if (!same(c[k], ref) && (!all || ex < 200u)) {
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
        example[k] = lex[k];
    }
}

int main(int argc, char**) {
    unsigned long long* bad; uint32_t *example, *mn, *mx;
    hipMalloc(&bad, kCand * 8); hipMalloc(&example, kCand * 4); hipMalloc(&mn, kCand * 4); hipMalloc(&mx, kCand * 4);
    hipMemset(bad, 0, kCand * 8); hipMemset(example, 0, kCand * 4); hipMemset(mn, 0xFF, kCand * 4); hipMemset(mx, 0, kCand * 4);
    const int all = argc > 1;   // any argument: every float, not only |x| >= 1e-6
    hipLaunchKernelGGL(k_check, dim3(256 * 32), dim3(256), 0, 0, all, bad, example, mn, mx);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    unsigned long long hb[kCand]; uint32_t he[kCand], hmn[kCand], hmx[kCand];
    hipMemcpy(hb, bad, sizeof hb, hipMemcpyDeviceToHost); hipMemcpy(he, example, sizeof he, hipMemcpyDeviceToHost);
    hipMemcpy(hmn, mn, sizeof hmn, hipMemcpyDeviceToHost); hipMemcpy(hmx, mx, sizeof hmx, hipMemcpyDeviceToHost);
    const char* names[kCand] = {"rcp", "rcp+newton", "rcp+newton+fixup", "rcp+2newton", "rcp+2newton+fixup", "rcp+newton+residual(r0)+fixup"};
    for (int k = 0; k < kCand; k++) {
        float x; std::memcpy(&x, &he[k], 4);
        printf("%-32s mismatches %12llu  biased exponent range [%u, %u]  e.g. x = %a (0x%08x)\n", names[k], hb[k], hmn[k], hmx[k], x, he[k]);
    }
    return 0;
}
