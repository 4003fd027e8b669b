// Host-side construction of the per-mesh binary BVH, following the reference's
// SBVH build (orion/avx/sbvh.cpp:94-235): per-triangle boxes and box centroids,
// split axis = largest extent, leaves of at most four triangles, and the three
// split rules MEDIAN (the live default, orion/avx/sbvh.hpp:141), MIDDLE and SAH.
#pragma once

#include "scene_types.hpp"

namespace orion_amd {

enum class SplitStrategy { MEDIAN = 0, MIDDLE = 1, SAH = 2 };

// Fills mesh.nodes (nodes[0] = root) and mesh.tri_order.
void build_bvh(Mesh& mesh, SplitStrategy strategy = SplitStrategy::MEDIAN);

inline void build_bvh(Model& model, SplitStrategy strategy = SplitStrategy::MEDIAN) {
    for (Mesh& m : model.meshes) build_bvh(m, strategy);
}

} // namespace orion_amd
