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
//
// Costs: the reference sweeps every candidate plane over every primitive of the
// node, O(300 n) per node — 40 s for 1 M triangles on one core.  Here all 300
// costs of a node come from ONE binning pass (binned_costs: same values bit for
// bit, 8.3 -> 1.2 s for 1 M random triangles on 8 threads with the larger tasks below).
//
// Threads (BVH_SAH with one primitive per leaf, the only way the reference
// calls it: Object.cpp:11-22):  With single-primitive leaves a subtree over n
// primitives has exactly 2n - 1 nodes, so where every node will be numbered is
// known before it is built: the nodes of a range rooted at node r whose
// descendants start at index d occupy [d, d + 2n - 2).  Large nodes are binned
// by several threads (private bins, merged; the arg-min is taken in the
// reference's axis/plane order);
// ranges below a threshold become independent tasks that write their own
// block.  Nodes and indices are bit-identical to the one-thread build.  Every other case (larger leaves, BVH_MIDDLE)
// goes through run_threaded_general: subtrees built into private arrays and renumbered afterwards.
#include "BVH.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <thread>

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

    // BVH_SAH, leaf size 1, more than `small` primitives: see the file comment.
    void run_threaded(int n_threads, int small) {
        const int n = (int)prims_.size();
        nodes_.assign(2 * (size_t)n - 1, BVHNode());
        struct Task {
            int begin, end, node, next;
        };
        std::vector<Task> tasks, top;
        top.push_back({0, n, 0, 1});
        // top of the tree, one node at a time, candidate planes spread over the threads
        while (!top.empty()) {
            Task t = top.back();
            top.pop_back();
            if (t.end - t.begin <= small) {
                tasks.push_back(t);
                continue;
            }
            int split = 0;
            Range r{t.begin, t.end, t.node};
            if (!split_node(r, &split, n_threads)) continue; // (n > small >= 1: never a leaf)
            nodes_[t.node].count = BVH_BRANCH;
            nodes_[t.node].left_node = t.next;
            const int n_left = split - t.begin;
            top.push_back({split, t.end, t.next + 1, t.next + 2 * n_left});
            top.push_back({t.begin, split, t.next, t.next + 2});
        }
        // the small subtrees, each by one thread into its own block of nodes
        std::atomic<size_t> cursor{0};
        auto worker = [&]() {
            std::vector<Task> todo;
            for (;;) {
                const size_t i = cursor.fetch_add(1);
                if (i >= tasks.size()) return;
                todo.push_back(tasks[i]);
                while (!todo.empty()) {
                    Task t = todo.back();
                    todo.pop_back();
                    int split = 0;
                    Range r{t.begin, t.end, t.node};
                    if (!split_node(r, &split, 1)) continue;
                    nodes_[t.node].count = BVH_BRANCH;
                    nodes_[t.node].left_node = t.next;
                    const int n_left = split - t.begin;
                    todo.push_back({split, t.end, t.next + 1, t.next + 2 * n_left});
                    todo.push_back({t.begin, split, t.next, t.next + 2});
                }
            }
        };
        std::vector<std::thread> pool;
        for (int i = 1; i < n_threads; i++) pool.emplace_back(worker);
        worker();
        for (auto& th : pool) th.join();
    }

    // Any leaf size, any strategy: subtree sizes are not known in advance, so the build runs in three steps.  (1) The top
    // of the tree, one node at a time with the candidate planes spread over the threads, down to ranges of at most
    // `small` primitives.  (2) Every such range is built by one thread into its OWN node array, numbered as run() numbers
    // a tree (root 0, children allocated in pairs, left subtree before the right one).  (3) One pass in the recursion's
    // order gives every node its final index: a top node allocates its pair and descends left then right; a range's local
    // nodes 1.. follow as one block, local index i -> base + i - 1.  Same nodes and indices as run(), bit for bit.
    void run_threaded_general(int n_threads, int small) {
        struct Top {
            int begin, end, left, right; // children: indices into `top`, -1 = this range is a task
            BVHNode node;
        };
        std::vector<Top> top;
        top.push_back({0, (int)prims_.size(), -1, -1, BVHNode()});
        std::vector<BVHNode> scratch(1);
        std::vector<int> task_of; // indices into `top` of the ranges built in step 2
        for (size_t i = 0; i < top.size(); i++) {
            const int b = top[i].begin, e = top[i].end;
            if (e - b <= small) {
                task_of.push_back((int)i);
                continue;
            }
            int split = 0;
            scratch[0] = BVHNode();
            if (!split_node(scratch, Range{b, e, 0}, &split, n_threads)) { // (only when leaf_size_ >= small)
                top[i].node = scratch[0];
                continue;
            }
            top[i].node = scratch[0];
            top[i].left = (int)top.size();
            top[i].right = (int)top.size() + 1;
            top.push_back({b, split, -1, -1, BVHNode()});
            top.push_back({split, e, -1, -1, BVHNode()});
        }
        std::vector<std::vector<BVHNode>> local(task_of.size());
        std::atomic<size_t> cursor{0};
        auto worker = [&]() {
            std::vector<Range> todo;
            for (;;) {
                const size_t k = cursor.fetch_add(1);
                if (k >= task_of.size()) return;
                std::vector<BVHNode>& nodes = local[k];
                nodes.reserve(2 * (size_t)(top[task_of[k]].end - top[task_of[k]].begin));
                nodes.emplace_back();
                todo.push_back({top[task_of[k]].begin, top[task_of[k]].end, 0});
                while (!todo.empty()) {
                    Range r = todo.back();
                    todo.pop_back();
                    int split = 0;
                    if (!split_node(nodes, r, &split, 1)) continue;
                    const int left = (int)nodes.size();
                    nodes.emplace_back();
                    nodes.emplace_back();
                    nodes[r.node].count = BVH_BRANCH;
                    nodes[r.node].left_node = left;
                    todo.push_back({split, r.end, left + 1});
                    todo.push_back({r.begin, split, left});
                }
            }
        };
        std::vector<std::thread> pool;
        for (int i = 1; i < n_threads; i++) pool.emplace_back(worker);
        worker();
        for (auto& th : pool) th.join();
        std::vector<int> local_of(top.size(), -1);
        for (size_t k = 0; k < task_of.size(); k++) local_of[task_of[k]] = (int)k;
        // step 3: final numbering in the recursion's order
        nodes_.clear();
        nodes_.emplace_back();
        struct Place {
            int top, node;
        };
        std::vector<Place> todo;
        todo.push_back({0, 0});
        while (!todo.empty()) {
            const Place p = todo.back();
            todo.pop_back();
            const Top& t = top[p.top];
            if (local_of[p.top] >= 0) {
                const std::vector<BVHNode>& nodes = local[local_of[p.top]];
                const int base = (int)nodes_.size(); // local node i >= 1 -> base + i - 1
                nodes_[p.node] = nodes[0];
                if (nodes[0].count == BVH_BRANCH) nodes_[p.node].left_node = base + nodes[0].left_node - 1;
                for (size_t i = 1; i < nodes.size(); i++) {
                    BVHNode n = nodes[i];
                    if (n.count == BVH_BRANCH) n.left_node = base + n.left_node - 1;
                    nodes_.push_back(n);
                }
                continue;
            }
            nodes_[p.node] = t.node;
            if (t.left < 0) continue; // a leaf made in step 1
            const int left = (int)nodes_.size();
            nodes_.emplace_back();
            nodes_.emplace_back();
            nodes_[p.node].count = BVH_BRANCH;
            nodes_[p.node].left_node = left;
            todo.push_back({t.right, left + 1});
            todo.push_back({t.left, left});
        }
    }

    std::vector<BVHNode> nodes_;
    std::vector<int> order_;

private:
    // Ranges of at least this many primitives spread their gathers (boxes, centres: one cache miss each) over the threads.
    static constexpr int kParallelRange = 262144;
    template <class F>
    static void for_chunks(int b, int e, int parts, F&& f) { // f(part, begin, end), the last part on the calling thread
        std::vector<std::thread> pool;
        const int64_t n = e - b;
        for (int t = 0; t < parts; t++) {
            const int tb = b + (int)(n * t / parts), te = b + (int)(n * (t + 1) / parts);
            if (t + 1 < parts) pool.emplace_back([&f, t, tb, te]() { f(t, tb, te); });
            else f(t, tb, te);
        }
        for (auto& th : pool) th.join();
    }
    Bounds enclose(int b, int e, int n_threads = 1) const {
        if (n_threads > 1 && e - b >= kParallelRange) { // min / max: the same box whatever the grouping
            std::vector<Bounds> part((size_t)n_threads);
            for_chunks(b, e, n_threads, [&](int t, int tb, int te) {
                Bounds box;
                for (int i = tb; i < te; i++) box = bounds_union(box, prims_[order_[i]]);
                part[(size_t)t] = box;
            });
            Bounds box;
            for (const Bounds& p : part) box = bounds_union(box, p);
            return box;
        }
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

    // All 300 costs of sweep_cost in ONE pass over the range.  On an axis the planes lo + k * step are non-decreasing in k
    // (rounding is monotone), so the planes a centre is NOT below form a prefix 0 .. m-1: the primitive is on the low side of
    // plane k exactly when k >= m.  Primitives are collected per m (101 bins: box and count), the low side of plane k is the
    // union of bins 0 .. k, the high side of bins k+1 .. 100.  A union is componentwise min / max — exact, whatever the order —
    // and the counts are integers, so every cost is the value sweep_cost computes, bit for bit (an empty side keeps the default
    // box and turns the cost into the same NaN -> 1e30).  O(n) per node instead of O(300 n); large ranges are binned by several
    // threads into private bins that are then merged (the same exactness argument).  false: planes not finite / not monotone
    // (overflowing extents) — the caller sweeps.
    struct Bins {
        Bounds box[3][101];
        int count[3][101];
    };
    void bin_range(int b, int e, const float (*planes)[100], const float* lo, const float* inv_step, const bool* active, Bins* out) const {
        for (int axis = 0; axis < 3; axis++)
            for (int m = 0; m <= 100; m++) out->count[axis][m] = 0;
        for (int i = b; i < e; i++) {
            const int p = order_[i];
            const Bounds& pb = prims_[p];
            for (int axis = 0; axis < 3; axis++) {
                if (!active[axis]) continue;
                const float c = centre_[p][axis];
                const float* pl = planes[axis];
                // m = number of planes <= c: a guess from the uniform spacing, then made exact against the plane values themselves
                const float g = (c - lo[axis]) * inv_step[axis];
                int m = g >= 0.f ? (g < 100.f ? (int)g + 1 : 100) : 0; // (NaN -> 0, corrected below like any other guess)
                while (m > 0 && pl[m - 1] > c) m--;
                while (m < 100 && pl[m] <= c) m++;
                out->box[axis][m] = bounds_union(out->box[axis][m], pb);
                out->count[axis][m]++;
            }
        }
    }
    bool binned_costs(int b, int e, const Bounds& box, int n_threads, float* cost_of) const {
        float planes[3][100], lo[3], inv_step[3];
        bool active[3];
        for (int axis = 0; axis < 3; axis++) {
            lo[axis] = box.bmin[axis];
            const float hi = box.bmax[axis];
            active[axis] = lo[axis] != hi;
            inv_step[axis] = 0.f;
            if (!active[axis]) continue;
            const float step = (hi - lo[axis]) / 100.f;
            inv_step[axis] = 1.f / step;
            for (int k = 0; k < 100; k++) {
                planes[axis][k] = lo[axis] + k * step;
                if (!(planes[axis][k] - planes[axis][k] == 0.f) || (k > 0 && planes[axis][k] < planes[axis][k - 1])) return false;
            }
            if (!(inv_step[axis] - inv_step[axis] == 0.f)) inv_step[axis] = 0.f; // (a guess only)
        }
        const int n = e - b;
        const int parts = std::max(1, std::min(n_threads, n / 65536));
        std::vector<Bins> bins((size_t)parts);
        if (parts == 1) {
            bin_range(b, e, planes, lo, inv_step, active, &bins[0]);
        } else {
            std::vector<std::thread> pool;
            for (int t = 0; t < parts; t++) {
                const int tb = b + (int)((int64_t)n * t / parts), te = b + (int)((int64_t)n * (t + 1) / parts);
                if (t + 1 < parts) pool.emplace_back([&, tb, te, t]() { bin_range(tb, te, planes, lo, inv_step, active, &bins[(size_t)t]); });
                else bin_range(tb, te, planes, lo, inv_step, active, &bins[(size_t)t]);
            }
            for (auto& th : pool) th.join();
            for (int t = 1; t < parts; t++)
                for (int axis = 0; axis < 3; axis++)
                    for (int m = 0; m <= 100; m++) {
                        bins[0].box[axis][m] = bounds_union(bins[0].box[axis][m], bins[(size_t)t].box[axis][m]);
                        bins[0].count[axis][m] += bins[(size_t)t].count[axis][m];
                    }
        }
        const Bins& all = bins[0];
        for (int axis = 0; axis < 3; axis++) {
            if (!active[axis]) continue;
            Bounds hi_of[101]; // hi_of[k] = union of bins k+1 .. 100
            int n_hi_of[101];
            Bounds acc;
            int cnt = 0;
            for (int m = 100; m >= 1; m--) {
                acc = bounds_union(acc, all.box[axis][m]);
                cnt += all.count[axis][m];
                hi_of[m - 1] = acc;
                n_hi_of[m - 1] = cnt;
            }
            Bounds low;
            int n_lo = 0;
            for (int k = 0; k < 100; k++) {
                low = bounds_union(low, all.box[axis][k]);
                n_lo += all.count[axis][k];
                const int n_hi = n_hi_of[k];
                const float cost = n_lo * low.surface_area() + n_hi * hi_of[k].surface_area();
                cost_of[axis * 100 + k] = cost > 0 ? cost : 1e30f;
            }
        }
        return true;
    }

    int partition_at(const Range& r, int axis, float plane, int n_threads = 1) {
        int split;
        if (n_threads > 1 && r.end - r.begin >= kParallelRange) {
            // std::partition's own permutation (libstdc++, bidirectional iterators: the k-th misplaced element from the front is
            // swapped with the k-th misplaced one from the back), with the predicate — the gather — evaluated beforehand by all threads
            std::vector<uint8_t> low((size_t)(r.end - r.begin));
            for_chunks(r.begin, r.end, n_threads, [&](int, int tb, int te) {
                for (int i = tb; i < te; i++) low[(size_t)(i - r.begin)] = centre_[order_[i]][axis] < plane ? 1 : 0;
            });
            int first = r.begin, last = r.end;
            for (;;) {
                while (first != last && low[(size_t)(first - r.begin)]) ++first;
                if (first == last) break;
                --last;
                while (first != last && !low[(size_t)(last - r.begin)]) --last;
                if (first == last) break;
                std::swap(order_[first], order_[last]);
                ++first;
            }
            split = first;
        } else {
            auto first = order_.begin() + r.begin, last = order_.begin() + r.end;
            auto mid = std::partition(first, last, [&](int p) { return centre_[p][axis] < plane; });
            split = int(mid - order_.begin());
        }
        if (split == r.begin || split == r.end) split = (r.begin + r.end) / 2;
        return split;
    }

    // returns false when the range became a leaf
    bool split_node(const Range& r, int* split, int n_threads = 1) { return split_node(nodes_, r, split, n_threads); }
    bool split_node(std::vector<BVHNode>& nodes, const Range& r, int* split, int n_threads) {
        BVHNode& node = nodes[r.node];
        node.aabb = enclose(r.begin, r.end, n_threads);
        const int n = r.end - r.begin;
        if (n <= leaf_size_) {
            node.left_node = r.begin;
            node.count = n;
            return false;
        }
        const Bounds box = node.aabb;
        if (how_ == BVH_MIDDLE) {
            const int axis = box.longest_axis();
            *split = partition_at(r, axis, box.get_center()[axis], n_threads);
            return true;
        }
        int best_axis = -1;
        float best_plane = 0, best_cost = 1e30f;
        float cost_of[300]; // candidate (axis, k) -> cost, scanned below in the reference's order
        const bool binned = binned_costs(r.begin, r.end, box, n_threads, cost_of);
        if (!binned && n_threads > 1) {
            std::atomic<int> next_candidate{0};
            auto sweep = [&]() {
                for (;;) {
                    const int c = next_candidate.fetch_add(1);
                    if (c >= 300) return;
                    const int axis = c / 100, k = c % 100;
                    const float lo = box.bmin[axis], hi = box.bmax[axis];
                    if (lo == hi) continue;
                    cost_of[c] = sweep_cost(r.begin, r.end, axis, lo + k * ((hi - lo) / 100.f));
                }
            };
            std::vector<std::thread> pool;
            for (int i = 1; i < n_threads; i++) pool.emplace_back(sweep);
            sweep();
            for (auto& th : pool) th.join();
        }
        for (int axis = 0; axis < 3; axis++) {
            const float lo = box.bmin[axis], hi = box.bmax[axis];
            if (lo == hi) continue;
            const float step = (hi - lo) / 100.f;
            for (int k = 0; k < 100; k++) {
                const float plane = lo + k * step;
                const float cost = (binned || n_threads > 1) ? cost_of[axis * 100 + k] : sweep_cost(r.begin, r.end, axis, plane);
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
        *split = best_axis < 0 ? (r.begin + r.end) / 2 : partition_at(r, best_axis, best_plane, n_threads);
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
    int n_threads = (int)std::thread::hardware_concurrency();
    if (const char* e = std::getenv("OMP_NUM_THREADS")) n_threads = std::min(n_threads > 0 ? n_threads : 1, std::max(1, std::atoi(e)));
    if (const char* e = std::getenv("RT_BVH_THREADS")) n_threads = std::max(1, std::atoi(e));
    n_threads = std::max(1, std::min(n_threads, 64));
    // primitives below which a subtree is one thread's task: large tasks (the top of the tree, one node at a time, is the part that
    // scales worst) but at least two per thread (2 M boxes on 8 threads: 4096: 4.3 s, 16384: 2.5, 65536: 2.3, 131072: 1.8)
    int small = std::min(131072, std::max(16384, (int)(bounds.size() / (2 * (size_t)n_threads))));
    if (const char* e = std::getenv("RT_BVH_SMALL")) small = std::max(1024, std::atoi(e));
    if (strat == BVH_SAH && max_per_node == 1 && n_threads > 1 && (int)bounds.size() > 2 * small)
        b.run_threaded(n_threads, small);
    else if (n_threads > 1 && (int)bounds.size() > 2 * small)
        b.run_threaded_general(n_threads, small);
    else
        b.run();
    out.nodes = std::move(b.nodes_);
    out.indicies = std::move(b.order_);
    return out;
}

} // namespace rt
