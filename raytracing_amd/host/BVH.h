// BVH.h — host BVH container and builder, API of the reference's BVH.h:9-33.
// The builder runs on the CPU exactly as in the reference (the north star
// keeps it on the host); its output is then flattened for the GPU (Flatten.h).
#pragma once
#include "rt_math.h"
#include <vector>

namespace rt {

enum PartitionStrategy { BVH_MIDDLE, BVH_SAH };

constexpr int BVH_BRANCH = -1;

struct BVHNode { // 32 B, BVH.h:16-21
    Bounds aabb;
    int left_node = 0;      // branch: first child (second is left_node + 1); leaf: first slot in indicies
    int count = BVH_BRANCH; // leaf: number of slots
};
static_assert(sizeof(BVHNode) == 32, "BVHNode must stay 32 bytes");

class BVH {
public:
    // BVH.cpp:131-149.  Produces the same nodes/indicies, bit for bit, as the
    // reference's recursive builder for the same bounds list.
    static BVH build(const std::vector<Bounds>& bounds, int max_per_node, PartitionStrategy strat);

    std::vector<BVHNode> nodes;
    std::vector<int> indicies;
};

} // namespace rt
