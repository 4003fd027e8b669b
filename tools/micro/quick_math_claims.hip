// Exhaustive check of the claims the kernels rely on, against the shipped functions themselves:
//   recip_quick(x) == 1.0f / x   bit for bit, for every float with 2^-126 <= |x| < 2^126
//   sqrt_quick(x)  == sqrtf(x)   bit for bit, for every float with 2^-102 <= x <= FLT_MAX
// and that sqrt_quick_ok() / dir_class() admit nothing outside those ranges.
// Prints the mismatch counts; exit status 0 iff both are zero.  Built by __graft_entry__.build()
// with the kernels' own flags; run by tests/test_gpu_parity.py.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "../../orion_amd/csrc/shading.h"

using namespace orion_dev;

__global__ void k_claims(unsigned long long* bad) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long r_bad = 0, s_bad = 0, g_bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const float x = __uint_as_float((uint32_t)i);
        const uint32_t ex = ((uint32_t)i >> 23) & 0xFFu;
        if (ex >= 1u && ex <= 252u) {
            if (__float_as_uint(recip_quick(x)) != __float_as_uint(1.0f / x)) r_bad++;
        }
        // the guard of the ray reciprocals: `normal` may only hold inside recip_quick's range
        bool small, normal;
        dir_class(mk(x, 1.0f, 1.0f), 8.50705917e+37f /* 2^126 */, small, normal);
        if (normal && !(ex >= 1u && ex <= 252u)) g_bad++;
        if (!(i >> 31)) {
            const bool in_range = ex >= 25u && ex <= 254u;
            if (sqrt_quick_ok(x) != in_range) g_bad++;
            if (in_range && __float_as_uint(sqrt_quick(x)) != __float_as_uint(sqrtf(x))) s_bad++;
        } else if (sqrt_quick_ok(x)) g_bad++;
    }
    if (r_bad) atomicAdd(&bad[0], r_bad);
    if (s_bad) atomicAdd(&bad[1], s_bad);
    if (g_bad) atomicAdd(&bad[2], g_bad);
}

int main() {
    unsigned long long* d_bad = nullptr;
    unsigned long long h_bad[3] = {~0ull, ~0ull, ~0ull};
    if (hipMalloc(&d_bad, sizeof h_bad) != hipSuccess || hipMemset(d_bad, 0, sizeof h_bad) != hipSuccess) { std::printf("no device\n"); return 2; }
    hipLaunchKernelGGL(k_claims, dim3(256 * 32), dim3(256), 0, 0, d_bad);
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h_bad, d_bad, sizeof h_bad, hipMemcpyDeviceToHost) != hipSuccess) { std::printf("kernel failed\n"); return 2; }
    std::printf("recip_quick mismatches: %llu  sqrt_quick mismatches: %llu  guard errors: %llu\n", h_bad[0], h_bad[1], h_bad[2]);
    return (h_bad[0] | h_bad[1] | h_bad[2]) ? 1 : 0;
}
