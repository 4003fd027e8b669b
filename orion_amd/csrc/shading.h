// Surface evaluation for the bounce loop: attribute interpolation, texel fetch,
// the two material models, light-surface sampling and the counter-based RNG.
// Operation order follows the reference expression by expression so that, with
// -ffp-contract=off, only libm-level functions (sinf, cosf, powf) can differ
// from a CPU evaluation of the same formulas.
#pragma once

#include "intersect.h"

namespace orion_dev {

// --- counter-based RNG ------------------------------------------------------
// Philox4x32-10.  counter = (pixel, sample, depth, block), key = seed.  One
// 32-bit word per draw.  Blocks 0..E*L-1 hold the light samples of a vertex
// (words: triangle pick, a, b), kBlockBounce the roulette / direction draws,
// kBlockPattern (with pixel = kPatternPixel) the per-sample pixel jitter.
constexpr uint32_t kBlockBounce = 0x80000000u;
constexpr uint32_t kBlockPattern = 0xC0000000u;
constexpr uint32_t kPatternPixel = 0xFFFFFFFFu;

struct Words4 { uint32_t w[4]; };

__device__ __forceinline__ Words4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        // full 32x32->64 products: one v_mad_u64_u32 each instead of a mul_hi/mul_lo pair
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    Words4 out; out.w[0] = c0; out.w[1] = c1; out.w[2] = c2; out.w[3] = c3;
    return out;
}

// Uniform float strictly inside (0,1): ((w >> 9) + 0.5) * 2^-23, exact in FP32.
__device__ __forceinline__ float u01(uint32_t w) { return ((float)(w >> 9) + 0.5f) * 1.1920928955078125e-07f; }

// --- helpers ------------------------------------------------------------------
// sqrtf(x) as reciprocal-square-root estimate and one correction step.  For every float with
// 2^-102 <= x <= FLT_MAX this returns the bits of the correctly rounded square root (checked for
// each of them, tools/micro/sqrt_exact.hip) in 5 instructions instead of about 20; for zero,
// smaller values and infinity it does not.
__device__ __forceinline__ float sqrt_quick(float x) {
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y, h = 0.5f * y;
    return fmaf(fmaf(-g, g, x), h, g);
}
__device__ __forceinline__ bool sqrt_quick_ok(float x) { return x >= 1.97215226e-31f && x <= 3.40282347e+38f; }   // 2^-102 .. FLT_MAX

// sqrtf(x), with the short form when every active lane's argument allows it (the same bits either way).
__device__ __forceinline__ float sqrt_exact(float x) {
    if (__all(sqrt_quick_ok(x))) return sqrt_quick(x);
    return sqrtf(x);
}

// normalize (orion/math.hpp:128-132): the reference multiplies by rsqrtps, a
// hardware-specific approximation; this build uses the exact 1/sqrt.  (A square root in
// sqrt_quick's range lies in [2^-51, 2^64], inside recip_quick's.)
__device__ __forceinline__ V3 normalize(V3 a) {
    const float x = dot(a, a);
    float r;
    if (__all(sqrt_quick_ok(x))) r = recip_quick(sqrt_quick(x));
    else r = 1.0f / sqrtf(x);
    return a * r;
}

// reflect (orion/math.hpp:321-323)
__device__ __forceinline__ V3 reflect(V3 I, V3 N) {
    float k = 2.0f * dot(N, I);
    return I - N * k;
}

// float -> int as the reference's cast compiles on x86 (cvttss2si): truncation,
// and INT_MIN for NaN or out-of-range values.
__device__ __forceinline__ int f2i_x86(float x) {
    return (x > -2147483904.0f && x < 2147483648.0f) ? (int)x : (int)0x80000000;
}

// Texture::color (orion/texture.hpp:72-86): nearest texel with the reference's
// wrap.  For a negative row index the reference reads past its array
// (vi = height + ui); that read is undefined there, so the row is wrapped by
// height here.
__device__ __forceinline__ V3 tex_fetch(const SceneView& S, uint32_t tex, float x, float y) {
    const Texture T = S.textures[tex];
    const int w = (int)T.width, hgt = (int)T.height;
    uint32_t texel = 0;
    if ((T.width | T.height) != 1u) {   // a 1x1 texture (every solid colour) maps any (x, y) to texel 0
        int ui = f2i_x86(x * (float)w);
        int vi = f2i_x86(y * (float)hgt);
        ui %= w;
        vi %= hgt;
        if (ui < 0) ui = w + ui;
        if (vi < 0) vi = hgt + vi;
        texel = (uint32_t)(w * vi + ui);
    }
    if (T.format == 0u) {
        const float* p = S.texels_f32 + T.offset + 3u * texel;
        return mk(p[0], p[1], p[2]);
    }
    const uint8_t* q = S.texels_u8 + T.offset;
    if (T.format == 1u) { q += 3u * texel; return mk((float)q[0] / 255.0f, (float)q[1] / 255.0f, (float)q[2] / 255.0f); }
    if (T.format == 2u) { q += 4u * texel; return mk((float)q[0] / 255.0f, (float)q[1] / 255.0f, (float)q[2] / 255.0f); }
    return splat((float)q[texel] / 255.0f);
}

enum { kTexAmbient = 0, kTexDiffuse = 1, kTexSpecular = 2, kTexEmissive = 3 };

struct Surface {
    V3 normal;      // interpolated, not yet normalised (orion/mesh.hpp:124-128)
    float tu, tv;   // interpolated texture coordinate (orion/mesh.hpp:148-152)
    uint32_t mesh;
};

__device__ __forceinline__ Surface surface_at(const SceneView& S, uint32_t gtri, float u, float v) {
    const float4* ap = reinterpret_cast<const float4*>(S.attrs + gtri);
    const float4 A0 = ap[0], A1 = ap[1], A2 = ap[2], A3 = ap[3];
    const float w = 1.0f - u - v;
    Surface s;
    s.normal = (mk(A0.x, A0.y, A0.z) * w + mk(A1.x, A1.y, A1.z) * u) + mk(A2.x, A2.y, A2.z) * v;
    s.tu = (w * A0.w + u * A2.w) + v * A3.y;
    s.tv = (w * A1.w + u * A3.x) + v * A3.z;
    s.mesh = __float_as_uint(A3.w);
    return s;
}

// Material::colorBRDF (orion/material.hpp:95-105); kd = diffuse texel at the shaded point.
__device__ __forceinline__ V3 color_brdf(V3 kd, V3 normal, V3 hit, V3 lpos, V3 lcol, float lint, V3 lnormal) {
    const V3 toL = lpos - hit;
    const float distance2 = dot(toL, toL);
    const V3 lightDir = normalize(toL);
    const float diff = tpl_max(dot(normal, lightDir) * dot(lnormal, -lightDir), 0.0f);
    const V3 diffuse = kd * diff;
    const V3 r = (lcol * diffuse) * lint;
    return r / (1 + distance2);
}

// Material::color, 5-argument overload (orion/material.hpp:72-93)
__device__ __forceinline__ V3 color_phong(V3 ka, V3 kd, V3 ks, float shininess, V3 rayDir, V3 normal, V3 hit,
                                          V3 lpos, V3 lcol, float lint) {
    const V3 toL = lpos - hit;
    const float distance2 = dot(toL, toL);
    const V3 lightDir = normalize(toL);
    const float diff = tpl_max(dot(normal, lightDir), 0.0f);
    const V3 diffuse = kd * diff;
    const V3 viewDir = normalize(-rayDir);
    const V3 reflectDir = reflect(-lightDir, normal);
    const float spec = 0.5f * powf(tpl_max(dot(viewDir, reflectDir), 0.0f), shininess);
    const V3 specular = ks * spec;
    const V3 sum = (ka + diffuse) + specular;
    const V3 r = (lcol * sum) * lint;
    return r / distance2;
}

// TracedMesh::randomPointOnSurface + Triangle::randomPointOnSurface
// (orion/mesh.hpp:179-184, orion/geometry.hpp:160-171), draws taken from one Philox block.
// bias = (triangle area / mesh area) * triangle count and intensity = bias * mesh area
// (orion/raytracer.cpp:153) depend on the picked triangle only; the scene upload computes them
// once per triangle with the same FP32 operations.
__device__ __forceinline__ void light_point(const SceneView& S, const Mesh& lm, Words4 r, V3& point, float& intensity) {
    // x % n without the division when n is a power of two (same value)
    const uint32_t ri = (lm.n_tri & (lm.n_tri - 1u)) == 0u ? (r.w[0] & (lm.n_tri - 1u)) : (r.w[0] % lm.n_tri);
    const float4* tp = reinterpret_cast<const float4*>(S.tris + lm.light_tris + ri);   // (the light's triangles in their original order)
    const float4 T0 = tp[0], T1 = tp[1], T2 = tp[2];
    const V3 v0 = mk(T0.x, T0.y, T0.z), e1 = mk(T1.x, T1.y, T1.z), e2 = mk(T2.x, T2.y, T2.z);
    intensity = T2.w;
    float a = u01(r.w[1]);
    float b = u01(r.w[2]);
    if (a + b > 1) { a = 1.0f - a; b = 1.0f - b; }
    point = (v0 + e1 * a) + e2 * b;
}

} // namespace orion_dev
