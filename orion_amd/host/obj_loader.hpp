// Wavefront OBJ/MTL import producing the per-mesh arrays the tracer consumes.
//
// The reference imports models through assimp (orion/model.cpp:19-193) with
// JoinIdenticalVertices | Triangulate | FlipUVs | GenNormals.  assimp is a
// system library this build does not depend on, so this loader reproduces the
// parts of that pipeline the tracer observes:
//   * one mesh per run of faces sharing a material / group, in file order;
//   * polygons fan-triangulated from their first corner;
//   * v texture coordinate flipped (v -> 1 - v);
//   * flat per-triangle normals generated when the file has none;
//   * bit-identical (position, normal, uv) corners joined per mesh.
// Mesh and triangle numbering are therefore defined by THIS loader.
#pragma once

#include <string>

#include "scene_types.hpp"

namespace orion_amd {

// Returns false (after printing the reason) when the file cannot be read; the
// model is then empty, as the reference's is after a failed import
// (orion/model.cpp:31-35).
bool load_obj(const std::string& path, Model& out);

} // namespace orion_amd
