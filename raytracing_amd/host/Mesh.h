// Mesh.h — the reference's target-design mesh container (Mesh.h:8-29).
#pragma once
#include "BVH.h"
#include <cstdint>
#include <vector>

namespace rt {

struct Vertex { // 32 B
    vec3 position;
    vec3 normal;
    vec2 uv;
};
static_assert(sizeof(Vertex) == 32, "Vertex must stay 32 bytes");

struct MeshTriangle { // 16 B
    uint32_t indicies[3] = {0, 0, 0};
    uint32_t materialIndex = 0;
};
static_assert(sizeof(MeshTriangle) == 16, "MeshTriangle must stay 16 bytes");

class Mesh {
public:
    // Declared but never defined in the reference (Mesh.h:24).  Defined here as
    // what the legacy TriangleMesh constructor does (Object.cpp:3-29): one
    // Bounds per triangle, BVH::build(bounds, 1, BVH_SAH).
    void build();
    void build(int max_per_leaf, PartitionStrategy strat);

    std::vector<Vertex> verts;
    std::vector<MeshTriangle> triangles;
    BVH accelStructure;
};

Bounds tri_bounds(const vec3& v1, const vec3& v2, const vec3& v3); // Object.cpp:3-9

} // namespace rt
