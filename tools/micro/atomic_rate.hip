// How fast can the blocks of a launch reserve queue slots through ONE 64-bit counter?  Each block of 256 threads
// takes 256 slots with one returning atomic (lane 0), hands the base to its threads through LDS and writes a 16-byte
// record per thread -- the queue append of the bounce loop without anything around it -- against the same launch
// with the base computed from the block index, and against counters spread over 2..64 addresses (128 bytes apart).
// hipcc --offload-arch=gfx950 -O3 tools/micro/atomic_rate.hip -o /tmp/atomic_rate && /tmp/atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int KIND>
__global__ __launch_bounds__(256) void k_append(unsigned long long* ctr, float4* out, unsigned parts, unsigned stride_blocks) {
    __shared__ unsigned s_base;
    unsigned base;
    if (KIND == 0) base = blockIdx.x * 256u;
    else {
        if (threadIdx.x == 0) {
            const unsigned part = blockIdx.x % parts;
            s_base = part * stride_blocks * 256u + (unsigned)atomicAdd(ctr + 16ull * part, 256ull);
        }
        __syncthreads();
        base = s_base;
    }
    out[base + threadIdx.x] = make_float4((float)base, 1.0f, 2.0f, 3.0f);
}

int main() {
    const unsigned blocks = 1u << 20;
    unsigned long long* ctr; float4* out;
    CK(hipMalloc(&ctr, 64 * 128)); CK(hipMalloc(&out, (size_t)blocks * 256 * 16));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run = [&](int kind, unsigned parts) -> float {
        float best = 1e9f;
        for (int rep = 0; rep < 4; rep++) {
            (void)hipMemset(ctr, 0, 64 * 128);
            (void)hipEventRecord(a);
            const unsigned stride = (blocks + parts - 1) / parts;
            if (kind == 0) hipLaunchKernelGGL(k_append<0>, dim3(blocks), dim3(256), 0, 0, ctr, out, parts, stride);
            else hipLaunchKernelGGL(k_append<1>, dim3(blocks), dim3(256), 0, 0, ctr, out, parts, stride);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        return best;
    };
    const float t0 = run(0, 1);
    printf("no counter:        %.3f ms  (%.2f TB/s written)\n", t0, blocks * 256.0 * 16 / t0 / 1e9);
    for (unsigned parts : {1u, 2u, 4u, 8u, 16u, 64u}) {
        const float t = run(1, parts);
        printf("%2u counter(s):     %.3f ms  -> %.1f ns per atomic beyond the plain launch\n", parts, t, (t - t0) * 1e6 / blocks);
    }
    return 0;
}
