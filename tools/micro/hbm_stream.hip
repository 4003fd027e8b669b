// What HBM bandwidth does this chip sustain?  Copy (read + write), read-only sum and write-only fill
// over 4 GiB buffers (far beyond the 256 MB Infinity Cache), float4 per lane, grid-stride.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/hbm_stream.hip -o hbm_stream
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void k_read(const float4* __restrict__ a, float* out, size_t n) {
    float s = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.678f) *out = s;   // never true for the zero-filled input; keeps the loads
}
__global__ void k_fill(float4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = make_float4(1.0f, 2.0f, 3.0f, 4.0f);
}

int main() {
    const size_t bytes = 4ull << 30, n = bytes / 16;
    float4 *a, *b; float* out;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) { std::printf("alloc failed\n"); return 1; }
    (void)hipMemset(a, 0, bytes); (void)hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const dim3 grid(256 * 16), block(256);
    for (int which = 0; which < 3; which++) {
        float best = 1e30f;
        for (int rep = 0; rep < 6; rep++) {
            (void)hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(k_copy, grid, block, 0, 0, a, b, n);
            if (which == 1) hipLaunchKernelGGL(k_read, grid, block, 0, 0, a, out, n);
            if (which == 2) hipLaunchKernelGGL(k_fill, grid, block, 0, 0, b, n);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        const double moved = (which == 0 ? 2.0 : 1.0) * bytes;
        std::printf("%s: %.3f ms, %.2f TB/s\n", which == 0 ? "copy (read + write)" : which == 1 ? "read" : "write", best, moved / best / 1e9);
    }
    return 0;
}
