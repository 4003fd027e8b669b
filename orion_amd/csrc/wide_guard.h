// The guard of the wide walk's two unproven skips (wide_walk.h, header comment), as a table: for every node of the 4-wide
// tree and every cell of a grid over all ray directions, four bits -- "may slot k hold a triangle whose test is ill
// conditioned for SOME direction of this cell".  Shared by the device code that builds and reads the table and by the CPU
// model of the walk (oracle/wide_model.cpp): plain C++.
//
// A triangle with unit normal n^ is ill conditioned for a direction d^ when |d^ . n^| < sin(beta), beta = asin(kGuardKappa /
// sin(apex)): d^ lies in a band of half-width beta around the great circle perpendicular to n^ (0.03 to 0.3 degrees wide for
// triangles that are not needles).  The union of such bands over a subtree is what a slot's certificate has to cover, and no
// compact shape does: a cone around a handful of scattered normals is crossed by most great circles (a leaf of four triangles
// of a bumpy model came out dangerous for 42 % of all directions, a subtree of 64 for 94 %).  A grid over the directions
// does: a cell of radius rho meets the band of a triangle iff |c^ . n^| < sin(beta + rho), c^ the cell's centre, so with cells
// about a degree across a leaf of four is dangerous for 5 % of the directions and a subtree of 16 for 19 %.
//
// Directions count as lines (d and -d are ill conditioned for the same triangles): a direction is folded onto z >= 0 and
// mapped to the square [-1, 1]^2 by the octahedral projection p = d / (|x| + |y| + z), (u, v) = (p.x + p.y, p.x - p.y); the grid
// is G x G equal squares.  Straight lines of the square are great circles on the sphere inside each of the map's four faces,
// and the faces meet along the square's diagonals, which pass through cell corners: a cell is a union of spherical polygons
// whose vertices are its four corners, so the largest angle between its centre and any of its directions is attained at a
// corner.  rho is that angle plus 2e-4 rad for the rounding of the map itself (a direction on a cell's border may be filed
// under the neighbour).
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define ORION_GUARD_FN __host__ __device__ inline
#else
#define ORION_GUARD_FN inline
#endif

namespace orion_dev {

constexpr uint32_t kGuardNoCell = 0xFFFFFFFFu;   // a ray that needs no guard

// The cell of a direction (components finite, not all zero).
ORION_GUARD_FN uint32_t guard_cell_of(float dx, float dy, float dz, uint32_t G) {
    const float s = dz < 0.0f ? -1.0f : 1.0f;
    const float x = dx * s, y = dy * s, z = dz * s;
    const float inv = 1.0f / (fabsf(x) + fabsf(y) + z);
    const float px = x * inv, py = y * inv;
    const float half = 0.5f * (float)G;
    int ix = (int)floorf((px + py + 1.0f) * half), iy = (int)floorf((px - py + 1.0f) * half);
    ix = ix < 0 ? 0 : (ix >= (int)G ? (int)G - 1 : ix);
    iy = iy < 0 ? 0 : (iy >= (int)G ? (int)G - 1 : iy);
    return (uint32_t)iy * G + (uint32_t)ix;
}

// The unit direction at a point of the square.
ORION_GUARD_FN void guard_square_to_dir(float u, float v, float out[3]) {
    const float px = 0.5f * (u + v), py = 0.5f * (u - v);
    float z = 1.0f - fabsf(px) - fabsf(py);
    z = z < 0.0f ? 0.0f : z;
    const float l = 1.0f / sqrtf(px * px + py * py + z * z);
    out[0] = px * l; out[1] = py * l; out[2] = z * l;
}

// Centre and radius (as sine and cosine) of a cell.
ORION_GUARD_FN void guard_cell_geometry(uint32_t G, uint32_t cell, float centre[3], float& sin_rho, float& cos_rho) {
    const uint32_t ix = cell % G, iy = cell / G;
    const float step = 2.0f / (float)G;
    const float u0 = -1.0f + step * (float)ix, v0 = -1.0f + step * (float)iy;
    guard_square_to_dir(u0 + 0.5f * step, v0 + 0.5f * step, centre);
    float cmin = 1.0f;
    for (int k = 0; k < 4; k++) {
        float c[3];
        guard_square_to_dir(u0 + ((k & 1) ? step : 0.0f), v0 + ((k & 2) ? step : 0.0f), c);
        const float dp = centre[0] * c[0] + centre[1] * c[1] + centre[2] * c[2];
        cmin = dp < cmin ? dp : cmin;
    }
    cmin = cmin < -1.0f ? -1.0f : cmin;
    const float rho = acosf(cmin) + 2.0e-4f;
    sin_rho = sinf(rho); cos_rho = cosf(rho);
}

// A triangle's band: its unit normal and sin(beta) (wide_bvh.h: guard_bands; sin(beta) > 1: dangerous for every direction).
struct GuardBand { float n[3]; float sin_beta; };

// Does the band meet the cell?
ORION_GUARD_FN bool guard_band_meets_cell(const GuardBand& b, const float centre[3], float sin_rho, float cos_rho) {
    if (!(b.sin_beta < cos_rho)) return true;      // beta + rho reaches a right angle (or the triangle has no normal)
    const float cos_beta = sqrtf(1.0f - b.sin_beta * b.sin_beta);
    const float lim = b.sin_beta * cos_rho + cos_beta * sin_rho + 1.0e-6f;
    return fabsf(centre[0] * b.n[0] + centre[1] * b.n[1] + centre[2] * b.n[2]) < lim;
}

// Where the four bits of (node, cell) live: the table is cut into blocks of 128 consecutive nodes, and a block holds one
// 64-byte line per cell with its 128 nodes' bits side by side.  Nodes are numbered depth first (wide_bvh.h), so the nodes a
// ray visits at the bottom of the tree fall into one or two blocks and the ray -- which asks about one cell only -- finds
// their bits in the line it already has, instead of one line per node step.  Returns the byte; the node's bits are its
// low half if the node's index is even, else its high half.
constexpr uint32_t kGuardBlockLog2 = 7;     // 128 nodes: 64 bytes of bits per cell
ORION_GUARD_FN size_t guard_table_offset(uint32_t node, uint32_t cell, uint32_t n_cells) {
    return (((size_t)(node >> kGuardBlockLog2) * n_cells + cell) << 6) + ((node & 127u) >> 1);
}
inline size_t guard_table_bytes(uint64_t n_nodes, uint32_t G) { return (size_t)((n_nodes + 127u) >> kGuardBlockLog2) * G * G * 64u; }

// Grid size for a tree of n_nodes wide nodes: G * G / 2 bytes per node.  A finer grid opens fewer subtrees, but the walk
// reads a byte of the table per node step, and with some 400 000 rays in flight, each asking about its own cell, a table
// beyond the L2's few megabytes answers from farther away: what the precision saves in visits the node steps lose in
// waiting.  On a 19 k-triangle model (5 018 nodes) G = 32 / 64 / 128 / 256 (2.6 / 10 / 41 / 164 MB) gave 3.51 / 3.30 / 3.22 /
// 3.19 G bounce rays/s and 1.25 / 1.31 / 1.37 / 1.40 G rays/s across the model.  So: the finest grid whose table stays within
// 8 MiB, and no coarser than 32 x 32 (cells about 3 degrees in radius).
inline uint32_t guard_grid_for(uint64_t n_nodes) {
    const uint64_t budget = 8ull << 20;
    for (uint32_t g = 256u; g > 32u; g >>= 1)
        if (n_nodes * (uint64_t)(g * g / 2u) <= budget) return g;
    return 32u;
}

} // namespace orion_dev
