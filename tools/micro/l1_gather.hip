// How many scattered 16-byte loads per cycle does a CU's vector memory path take?  The walk of a large tree
// has every lane fetch its own 64-byte node (four 16-byte loads, 64 different cache lines per wave
// instruction); this measures that pattern against the same bytes fetched by four neighbouring lanes per
// node (16 lines per instruction) and against a contiguous read, with the records resident in L2
// (a table of 320 KB, the size of a 5 000-node tree) or in L1 (16 KB).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/l1_gather.hip -o l1_gather
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

constexpr int kBlock = 256;

__device__ __forceinline__ uint32_t lcg(uint32_t x) { return x * 1664525u + 1013904223u; }

// MODE 0: lane -> own record, 4 loads of 16 B.  MODE 1: four lanes share a record, one 16-byte load each,
// four records per group of 4 instructions (same bytes per lane as MODE 0).  MODE 2: lane -> own record, 1 load.
// MODE 3: consecutive lanes read consecutive 16 B (fully coalesced), 4 loads.
// PATTERN (which lanes take part): 0 all, 1 lanes 0-15 (four whole quads... sixteen lanes in four-lane groups),
// 2 one lane in four (every quad, a quarter full), 3 a random quarter of the lanes, 4 lanes 0-31, 5 a random half.
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_gather(const float4* __restrict__ tab, uint32_t n_rec, uint32_t iters, float* out, int pattern) {
    uint32_t x = (blockIdx.x * kBlock + threadIdx.x) * 2654435761u + 12345u;
    const uint32_t lane = threadIdx.x & 63u;
    float s = 0.0f;
    const uint32_t hsh = lcg(lcg(x)) >> 16;
    const bool on = pattern == 0 || (pattern == 1 && lane < 16u) || (pattern == 2 && (lane & 3u) == 0u) || (pattern == 3 && (hsh & 3u) == 0u) ||
                    (pattern == 4 && lane < 32u) || (pattern == 5 && (hsh & 1u) == 0u);
    if (!on) return;
    for (uint32_t it = 0; it < iters; it++) {
        x = lcg(x);
        float4 a, b, c, d;
        if (MODE == 0) {
            const uint32_t r = (x >> 8) % n_rec;
            const float4* p = tab + 4ull * r;
            a = p[0]; b = p[1]; c = p[2]; d = p[3];
        } else if (MODE == 1) {
            // records chosen per quad: lanes 4q..4q+3 take the quad leader's random numbers
            const uint32_t xq = __shfl(x, (int)(lane & ~3u));
            const uint32_t r0 = (xq >> 8) % n_rec, r1 = (xq >> 9) % n_rec, r2 = (xq >> 10) % n_rec, r3 = (xq >> 11) % n_rec;
            const uint32_t part = lane & 3u;
            a = tab[4ull * r0 + part]; b = tab[4ull * r1 + part]; c = tab[4ull * r2 + part]; d = tab[4ull * r3 + part];
        } else if (MODE == 2) {
            const uint32_t r = (x >> 8) % n_rec;
            a = tab[4ull * r]; b = c = d = make_float4(0, 0, 0, 0);
        } else {
            const uint32_t r = ((x >> 8) % n_rec) & ~63u;   // wave-uniform? no: per lane; make it wave-uniform
            const uint32_t ru = __builtin_amdgcn_readfirstlane(r) % (n_rec - 64u);
            const float4* p = tab + 4ull * ru + lane;
            a = p[0]; b = p[64]; c = p[128]; d = p[192];
        }
        s += (a.x + b.y) + (c.z + d.w);
    }
    if (s == 12345.678f) *out = s;
}

int main() {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float* out; (void)hipMalloc(&out, 4);
    const uint32_t iters = 2000;
    for (uint32_t n_rec : {256u, 5000u, 20000u}) {
        std::vector<float> h(16ull * n_rec, 1.0f);
        float4* tab; (void)hipMalloc(&tab, 64ull * n_rec);
        (void)hipMemcpy(tab, h.data(), 64ull * n_rec, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 4; mode++) {
          for (int pattern = 0; pattern < (mode == 0 ? 6 : 1); pattern++) {
            for (int per_cu : {6}) {              // blocks per CU (4 waves each): 6 = what the walk kernel holds
                const dim3 grid(256 * per_cu), block(kBlock);
                float best = 1e30f;
                for (int rep = 0; rep < 4; rep++) {
                    (void)hipEventRecord(e0);
                    if (mode == 0) hipLaunchKernelGGL(k_gather<0>, grid, block, 0, 0, tab, n_rec, iters, out, pattern);
                    if (mode == 1) hipLaunchKernelGGL(k_gather<1>, grid, block, 0, 0, tab, n_rec, iters, out, pattern);
                    if (mode == 2) hipLaunchKernelGGL(k_gather<2>, grid, block, 0, 0, tab, n_rec, iters, out, pattern);
                    if (mode == 3) hipLaunchKernelGGL(k_gather<3>, grid, block, 0, 0, tab, n_rec, iters, out, pattern);
                    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                    if (rep && ms < best) best = ms;
                }
                const double loads = (mode == 2 ? 1.0 : 4.0) * iters * (double)grid.x * (kBlock / 64);   // wave-level load instructions
                const double per_cu_ns = loads / 256.0 / (best * 1e6);
                std::printf("table %7u records (%6.0f KB)  mode %d  lanes %d  %d waves/CU: %.3f ms, %.4f wave loads / ns / CU, %.2f lane loads / ns / CU, %.1f GB/s per CU\n",
                            n_rec, 64.0 * n_rec / 1024.0, mode, pattern, per_cu * 4, best, per_cu_ns, per_cu_ns * 64.0, per_cu_ns * 64.0 * 16.0);
            }
          }
        }
        (void)hipFree(tab);
    }
    return 0;
}
