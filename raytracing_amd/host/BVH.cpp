// BVH.cpp — top-down BVH builder with the reference's split rules
// (BVH.cpp:58-128): leaf when n <= max_per_node; BVH_MIDDLE = centroid
// partition at the node centre on the longest axis; BVH_SAH = best of
// 3 axes x 100 uniform planes by full-sweep surface-area cost; degenerate
// partitions fall back to the index median; the two children are allocated
// adjacently, left subtree numbered before the right one.
//
// Written as an explicit work list instead of recursion (10 M-triangle scenes
// produce deep trees); the node numbering it yields is the recursion's
// pre-order, so output is identical.
#include "BVH.h"

#include <algorithm>
#include <cstdio>

namespace rt {
namespace {

struct Range {
    int begin, end, node;
};

class TopDownBuilder {
public:
    TopDownBuilder(const std::vector<Bounds>& prims, int leaf_size, PartitionStrategy how)
        : prims_(prims), leaf_size_(leaf_size), how_(how) {
        const int n = (int)prims.size();
        order_.resize(n);
        centre_.resize(n);
        for (int i = 0; i < n; i++) {
            order_[i] = i;
            centre_[i] = prims[i].get_center();
        }
        nodes_.reserve(2 * (size_t)n + 1);
    }

    void run() {
        nodes_.emplace_back();
        std::vector<Range> todo;
        todo.push_back({0, (int)prims_.size(), 0});
        while (!todo.empty()) {
            Range r = todo.back();
            todo.pop_back();
            int split = 0;
            if (!split_node(r, &split)) continue;
            const int left = (int)nodes_.size();
            nodes_.emplace_back();
            nodes_.emplace_back();
            nodes_[r.node].count = BVH_BRANCH;
            nodes_[r.node].left_node = left;
            // right is pushed first so that the whole left subtree is numbered before it
            todo.push_back({split, r.end, left + 1});
            todo.push_back({r.begin, split, left});
        }
    }

    std::vector<BVHNode> nodes_;
    std::vector<int> order_;

private:
    Bounds enclose(int b, int e) const {
        Bounds box;
        for (int i = b; i < e; i++) box = bounds_union(box, prims_[order_[i]]);
        return box;
    }

    // cost of cutting [b,e) at `plane` on `axis` (EvalSAH, BVH.cpp:30-46)
    float sweep_cost(int b, int e, int axis, float plane) const {
        Bounds lo, hi;
        int n_lo = 0, n_hi = 0;
        for (int i = b; i < e; i++) {
            const int p = order_[i];
            if (centre_[p][axis] < plane) {
                n_lo++;
                lo = bounds_union(lo, prims_[p]);
            } else {
                n_hi++;
                hi = bounds_union(hi, prims_[p]);
            }
        }
        float cost = n_lo * lo.surface_area() + n_hi * hi.surface_area();
        return cost > 0 ? cost : 1e30f;
    }

    int partition_at(const Range& r, int axis, float plane) {
        auto first = order_.begin() + r.begin, last = order_.begin() + r.end;
        auto mid = std::partition(first, last, [&](int p) { return centre_[p][axis] < plane; });
        int split = int(mid - order_.begin());
        if (split == r.begin || split == r.end) split = (r.begin + r.end) / 2;
        return split;
    }

    // returns false when the range became a leaf
    bool split_node(const Range& r, int* split) {
        BVHNode& node = nodes_[r.node];
        node.aabb = enclose(r.begin, r.end);
        const int n = r.end - r.begin;
        if (n <= leaf_size_) {
            node.left_node = r.begin;
            node.count = n;
            return false;
        }
        const Bounds box = node.aabb;
        if (how_ == BVH_MIDDLE) {
            const int axis = box.longest_axis();
            *split = partition_at(r, axis, box.get_center()[axis]);
            return true;
        }
        int best_axis = -1;
        float best_plane = 0, best_cost = 1e30f;
        for (int axis = 0; axis < 3; axis++) {
            const float lo = box.bmin[axis], hi = box.bmax[axis];
            if (lo == hi) continue;
            const float step = (hi - lo) / 100.f;
            for (int k = 0; k < 100; k++) {
                const float plane = lo + k * step;
                const float cost = sweep_cost(r.begin, r.end, axis, plane);
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = axis;
                    best_plane = plane;
                }
            }
        }
        // The reference reads an uninitialised axis/plane when no plane scores
        // below 1e30 (BVH.cpp:91-97, e.g. coincident primitives).  Defined here as
        // the median split, which is also what its degenerate-partition rule gives.
        *split = best_axis < 0 ? (r.begin + r.end) / 2 : partition_at(r, best_axis, best_plane);
        return true;
    }

    const std::vector<Bounds>& prims_;
    const int leaf_size_;
    const PartitionStrategy how_;
    std::vector<vec3> centre_;
};

} // namespace

BVH BVH::build(const std::vector<Bounds>& bounds, int max_per_node, PartitionStrategy strat) {
    BVH out;
    if (bounds.empty() || max_per_node < 1) {
        std::printf("Error on BVH construction\n"); // BVH.cpp:138-141
        return out;
    }
    TopDownBuilder b(bounds, max_per_node, strat);
    b.run();
    out.nodes = std::move(b.nodes_);
    out.indicies = std::move(b.order_);
    return out;
}

} // namespace rt
