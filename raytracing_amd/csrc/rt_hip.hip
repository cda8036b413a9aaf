// rt_hip.hip — wavefront path tracer for MI355X (gfx950) and its C-ABI
// (include/rt_hip.h).  Kernels, one per reference loop:
//
//   k_generate    hammersley_2d + Camera::get_ray          (Scene.h:61-75)
//   k_closest     TriangleMesh::intersect traversal        (Object.h:273-357)
//   k_classify    lists the bounce's hits per material type (no reference counterpart: replaces virtual dispatch)
//   k_shade_t<T>  get_ray_color body + shade_direct_NEE    (main.cpp:240-441), one launch per material type,
//                 + wave64 ballot/prefix-sum compaction of the next ray queue and of the shadow queue
//   k_shadow      trace_scene as a boolean                 (main.cpp:250,281)
//   k_nee_resolve radiance += throughput * direct          (main.cpp:430)
//   k_accumulate  NaN guard + per-pixel sum                (main.cpp:984-987)
//   k_resolve     add_to_buffer                            (main.cpp:865-877)
//
// Data layout in HBM: see DESIGN.md §3.  All queues are SoA float4 streams so
// every wave-instruction reads/writes 1 KiB contiguously; the BVH is read as
// aligned 64-byte child-pair records and 48-byte triangle records.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared
#include "rt_device.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

using namespace rtd;

#define RT_MAX_LIGHTS 15 // shadow slots per path = env + lights <= 16
#define RT_SHARDS 64      // sub-queues per queue: one reservation counter each (a single counter saturates at ~88 atomics/us)
#define RT_COUNTS (128 * RT_SHARDS)

// ------------------------------------------------------------------ errors
static thread_local std::string g_err;
static thread_local const char* g_err_lit = nullptr; // message of a failure reported without touching the heap
static int fail(int code, const std::string& msg) {
    g_err = msg;
    g_err_lit = nullptr;
    return code;
}
int rt_fail_message(int code, const std::string& msg) { return fail(code, msg); } // for rt_comm.hip (same library)
static int fail_lit(int code, const char* literal) { // for the entry point that must not allocate (std::string may)
    g_err_lit = literal;
    return code;
}
#define HIPCHK(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return fail(RT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                      \
    } while (0)

// ------------------------------------------------------------------ kernels

// Per-pass wavefront state.  cap = paths per pass.
struct Wave {
    // ray queues (double buffered), indexed by queue position.  The path's state TRAVELS WITH ITS RAY: a queue entry is
    // four float4 streams, gathered through `perm` (mostly ascending within a material's list) and written at the next
    // queue position in lane order.  (One 64-byte record per path id was a random HBM sector read and written per shaded
    // hit: a queue entry's path id is random after the first compaction.)
    float4* q_o[2];   // origin.xyz, path id (bits)
    float4* q_d[2];   // dir.xyz, random-number counter (bits)
    float4* q_T[2];   // throughput.xyz, scatter_pdf
    float4* q_p[2];   // prev.point.xyz (main.cpp:437), random-number key (bits)
    float4* hit;      // slot (bits), t, u, v
    float4* hit2;     // two-level scenes only: analytic primitive's oriented model-space normal, instance | front_face << 30 (bits)
    // radiance.xyz of every path, indexed by path id: touched only when something arrives (an emitter is hit, the path
    // leaves the scene under an environment, next-event estimation finds light)
    float4* rad;
    // Dense shadow-ray queue; the next-event-estimation records live in it, at the rays' own positions: a lane's shadow
    // rays are consecutive entries and lanes reserve in order, so every NEE store is a coalesced 1-KiB burst.  (Records
    // indexed by the shaded queue entry were scattered 16-byte writes in material order: with their resolve pass a third
    // of all shading time.)
    float4* s_o;      // origin.xyz, tmax
    float4* s_d;      // dir.xyz, -
    float4* s_c;      // direct contribution of this ray (before throughput), path id + 1 (bits) on a path's FIRST ray else 0
    float4* s_T;      // first ray only: throughput before this bounce's scatter, number of rays of the path (bits)
    uint8_t* vis;     // 1 = shadow ray unoccluded, written by k_shadow at the ray's position
    // counters: [b*S + s] live rays of shard s entering bounce b (bounce 0: k_generate deals the camera rays round the shards in groups of RT_BLOCK,
    // as the shading kernels do for later bounces), [64*S + b*S + s] shadow rays of shard s made at bounce b
    // Counter i lives at word i * RT_CSTRIDE (one 128-byte line each).
    uint32_t* counts;
    uint32_t* cursors; // same layout: rays of that queue shard handed out to traversal waves so far
    uint32_t* counts_dense; // RT_COUNTS words: the counters gathered for the host at the end of a pass
    uint32_t* type_counts;  // [(bounce * 16 + key) * RT_CSTRIDE]: queue entries of the bounce whose hit has material type key
    uint32_t* perm;         // [key * capq + i]: their queue indices
};

struct PassParams {
    rt_camera cam;
    int32_t width;
    int32_t x0, y0, tw, npix; // tile origin, tile width, tile pixel count
    int32_t spp_first;        // first sample index of this pass
    int32_t spp_pass;         // samples per pixel in this pass
    uint32_t spp_total;
    uint32_t seed;
    int32_t cap;
    int32_t K;                // shadow slots per path
    uint32_t n_shards;        // sub-queues per ray queue (bounce >= 1) and per shadow queue
    uint32_t shard_cap;       // ray-queue entries reserved per shard (multiple of RT_BLOCK)
    uint32_t capq;            // n_shards * shard_cap >= cap: range of queue indices
    int32_t has_env;
    // rt_render_tiles: the call's pixels are those of n_tiles disjoint rectangles, numbered tile after tile (row-major
    // within a tile); tile_tab[k] = (x0, y0, width, number of the tile's first pixel).  n_tiles == 0: the one rectangle above.
    const int4* tile_tab;
    int32_t n_tiles;
};

// pixel number within the call (0 .. npix) -> raster position
RT_DEV void pixel_of(const PassParams& pp, int pix, int* x, int* y) {
    if (pp.n_tiles == 0) {
        *x = pp.x0 + pix % pp.tw;
        *y = pp.y0 + pix / pp.tw;
        return;
    }
    int lo = 0, hi = pp.n_tiles - 1;
    while (lo < hi) { // the last tile whose first pixel is <= pix
        const int mid = (lo + hi + 1) >> 1;
        if (pp.tile_tab[mid].w <= pix) lo = mid;
        else hi = mid - 1;
    }
    const int4 t = pp.tile_tab[lo];
    const int local = pix - t.w;
    *x = t.x + local % t.z;
    *y = t.y + local / t.z;
}

// Camera rays.  Path p = (sample p / npix of the pass, pixel p % npix) is written to queue entry
//   shard * shard_cap + j * RT_BLOCK + lane,   (shard, j) = (group % n_shards, group / n_shards),   group = p / RT_BLOCK:
// the groups of RT_BLOCK consecutive paths go round the shards, as the entries the shading kernels write for later bounces do.
// (Shard s = the s-th consecutive piece of the dense array made every shard a band of the image; the band with the expensive
// pixels finished long after the others.)  A wave still gets 64 neighbouring pixels of one sample.
__global__ __launch_bounds__(RT_BLOCK) void k_generate(Wave w, PassParams pp) {
    const int p = blockIdx.x * RT_BLOCK + threadIdx.x;
    if ((uint32_t)p < pp.n_shards) {
        const uint32_t groups = (uint32_t)pp.cap / RT_BLOCK, rem = (uint32_t)pp.cap % RT_BLOCK, s = (uint32_t)p;
        w.counts[(size_t)p * RT_CSTRIDE] = (groups / pp.n_shards + (s < groups % pp.n_shards ? 1u : 0u)) * RT_BLOCK + (s == groups % pp.n_shards ? rem : 0u);
    }
    if (p >= pp.cap) return;
    const uint32_t q = (blockIdx.x % pp.n_shards) * pp.shard_cap + (blockIdx.x / pp.n_shards) * RT_BLOCK + threadIdx.x;
    const int s_local = p / pp.npix, pix = p - s_local * pp.npix;
    int x, y;
    pixel_of(pp, pix, &x, &y);
    V3 o, d;
    camera_ray(pp.cam, x, y, (uint32_t)(pp.spp_first + s_local), pp.spp_total, &o, &d);
    w.q_o[0][q] = pack(o, __int_as_float(p));
    w.q_d[0][q] = pack(d, __int_as_float(0));
    w.q_T[0][q] = make_float4(1.f, 1.f, 1.f, 1.f);
    w.q_p[0][q] = make_float4(0.f, 0.f, 0.f, __uint_as_float(rng_key((uint32_t)(y * pp.width + x), (uint32_t)(pp.spp_first + s_local), pp.seed)));
    w.rad[p] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// Copies the first RT_TOP_PAIRS child-pair records into LDS (whole workgroup), as the four part arrays load_pair reads,
// and returns how many bytes of the pair array that covers.
RT_DEV int stage_tree_top(const SceneView& sc, float4* s_top) {
    const int n_top = sc.n_pairs < RT_TOP_PAIRS ? sc.n_pairs : RT_TOP_PAIRS;
    for (int i = threadIdx.x; i < 4 * n_top; i += RT_TRAV_BLOCK) {
        const int rec = i >> 2, part = i & 3;
        const float4 v = sc.pairs[i];
        if (part < 3) s_top[part * RT_TOP_PAIRS + rec] = v;
        else ((float2*)(s_top + 3 * RT_TOP_PAIRS))[rec] = make_float2(v.x, v.y);
    }
    __syncthreads();
    return n_top * RT_PAIR_BYTES;
}

__global__ __launch_bounds__(RT_TRAV_BLOCK) void k_closest(SceneView sc, const float4* __restrict__ q_o, const float4* __restrict__ q_d,
                                                      const uint32_t* __restrict__ counts, uint32_t* __restrict__ cursors,
                                                      uint32_t n_shards, uint32_t shard_cap, float4* __restrict__ hit, int want_tri_id) {
    __shared__ int stack[RT_SHORT_STACK * RT_TRAV_BLOCK];
    __shared__ float4 s_top[RT_TOP_LDS_BYTES / 16];
    const int top_bytes = stage_tree_top(sc, s_top);
    trace_queue(sc, q_o, q_d, counts, cursors, n_shards, shard_cap, (lds_int*)stack + threadIdx.x, (lds_cf4*)s_top, top_bytes, [&](const RayState& r) {
        int slot = r.slot;
        if (want_tri_id && slot >= 0) slot = (int)(__float_as_uint(sc.tris[3 * (size_t)slot].w) & ~RT_TRI_LEAF_END);
        hit[r.q] = make_float4(__int_as_float(slot), r.t, r.u, r.v);
    });
}

// Dense shadow queue -> vis[target] = 1 when nothing blocks the segment.
__global__ __launch_bounds__(RT_TRAV_BLOCK) void k_shadow(SceneView sc, const float4* __restrict__ s_o, const float4* __restrict__ s_d,
                                                     const uint32_t* __restrict__ counts, uint32_t* __restrict__ cursors,
                                                     uint32_t n_shards, uint32_t shard_cap, uint8_t* __restrict__ vis) {
    __shared__ int stack[RT_SHORT_STACK * RT_TRAV_BLOCK];
    __shared__ float4 s_top[RT_TOP_LDS_BYTES / 16];
    const int top_bytes = stage_tree_top(sc, s_top);
    occlusion_queue(sc, s_o, s_d, counts, cursors, n_shards, shard_cap, (lds_int*)stack + threadIdx.x, (lds_cf4*)s_top, top_bytes,
                    [&](uint32_t q, uint32_t, bool occluded) { vis[q] = occluded ? 0 : 1; });
}

// Parity hook variant: rays + tmax in, occluded flag out (rt_trace_any).
__global__ __launch_bounds__(RT_TRAV_BLOCK) void k_any(SceneView sc, const float4* __restrict__ q_o, const float4* __restrict__ q_d,
                                                  const uint32_t* __restrict__ counts, uint32_t* __restrict__ cursors, uint32_t n_shards,
                                                  uint32_t shard_cap, uint8_t* __restrict__ occ) {
    __shared__ int stack[RT_SHORT_STACK * RT_TRAV_BLOCK];
    __shared__ float4 s_top[RT_TOP_LDS_BYTES / 16];
    const int top_bytes = stage_tree_top(sc, s_top);
    occlusion_queue(sc, q_o, q_d, counts, cursors, n_shards, shard_cap, (lds_int*)stack + threadIdx.x, (lds_cf4*)s_top, top_bytes,
                    [&](uint32_t q, uint32_t, bool occluded) { occ[q] = occluded ? 1 : 0; });
}

// ---- the occlusion kernels over quad records (rt_device.h, "four-wide occlusion walk"): launched instead of k_shadow /
// k_any when the scene has them
RT_DEV int stage_quad_top(const SceneView& sc, float4* s_top) {
    const int n_top = sc.n_quads < RT_QTOP ? sc.n_quads : RT_QTOP;
    for (int i = threadIdx.x; i < 4 * n_top; i += RT_TRAV_BLOCK) {
        const int rec = i >> 2, part = i & 3;
        const float4 v = sc.onodes[i];
        if (part < 3) s_top[part * RT_QTOP + rec] = v;
        else ((float2*)(s_top + 3 * RT_QTOP))[rec] = make_float2(v.x, v.y);
    }
    __syncthreads();
    return n_top * RT_ONODE_BYTES;
}
__global__ __launch_bounds__(RT_TRAV_BLOCK, 8) void k_shadow4(SceneView sc, const float4* __restrict__ s_o, const float4* __restrict__ s_d,
                                                      const uint32_t* __restrict__ counts, uint32_t* __restrict__ cursors,
                                                      uint32_t n_shards, uint32_t shard_cap, uint8_t* __restrict__ vis) {
    __shared__ int stack[RT_QSHORT_STACK * RT_TRAV_BLOCK];
    __shared__ float4 s_top[RT_QTOP_LDS_BYTES / 16];
    const int top_bytes = stage_quad_top(sc, s_top);
    occlusion_queue4(sc, s_o, s_d, counts, cursors, n_shards, shard_cap, (lds_int*)stack + threadIdx.x, (lds_cf4*)s_top, top_bytes,
                     [&](uint32_t q, uint32_t, bool occluded) { vis[q] = occluded ? 0 : 1; });
}
__global__ __launch_bounds__(RT_TRAV_BLOCK, 8) void k_any4(SceneView sc, const float4* __restrict__ q_o, const float4* __restrict__ q_d,
                                                   const uint32_t* __restrict__ counts, uint32_t* __restrict__ cursors, uint32_t n_shards,
                                                   uint32_t shard_cap, uint8_t* __restrict__ occ) {
    __shared__ int stack[RT_QSHORT_STACK * RT_TRAV_BLOCK];
    __shared__ float4 s_top[RT_QTOP_LDS_BYTES / 16];
    const int top_bytes = stage_quad_top(sc, s_top);
    occlusion_queue4(sc, q_o, q_d, counts, cursors, n_shards, shard_cap, (lds_int*)stack + threadIdx.x, (lds_cf4*)s_top, top_bytes,
                     [&](uint32_t q, uint32_t, bool occluded) { occ[q] = occluded ? 1 : 0; });
}

// ---- two-level scenes (rt_scene_create_instanced): the same persistent waves over the same sharded queues
#define RT_HIT2_FRONT 0x40000000
// waves per SIMD the two-level kernels are compiled for: 8 = two 1024-thread workgroups per CU (64 VGPRs; the closest walk
// then keeps 8 dwords in scratch).  Measured on config 5 two-level, one pass at a time: 4 waves (78 / 66 VGPRs) closest
// 181.8 ms, shadow 175.8 ms per 3 steps; 8 waves 170.2 and 129.3.
#ifndef RT_TL_WAVES
#define RT_TL_WAVES 8
#endif
__global__ __launch_bounds__(RT_TRAV_BLOCK, RT_TL_WAVES) void k_closest2(SceneView sc, const float4* __restrict__ q_o, const float4* __restrict__ q_d,
                                                       const uint32_t* __restrict__ counts, uint32_t* __restrict__ cursors, uint32_t n_shards,
                                                       uint32_t shard_cap, float4* __restrict__ hit, float4* __restrict__ hit2, int want_tri_id) {
    __shared__ int stack[RT_SHORT_STACK * RT_TRAV_BLOCK];
    __shared__ float4 s_top[RT_TOP_LDS_BYTES / 16];
    const int top_bytes = stage_tree_top(sc, s_top);
    __shared__ int hit_rec[7 * RT_TRAV_BLOCK]; // the closest hit's record beyond its distance (trace_queue2)
    trace_queue2<false>(sc, q_o, q_d, counts, cursors, n_shards, shard_cap, (lds_int*)stack + threadIdx.x, (lds_int*)hit_rec + threadIdx.x, (lds_cf4*)s_top, top_bytes,
                        [&](uint32_t q, const Hit2& h, bool) {
                            int slot = h.slot;
                            if (want_tri_id && slot != -1) slot = slot >= 0 ? (int)(__float_as_uint(sc.tris[3 * (size_t)slot].w) & ~RT_TRI_LEAF_END) : 0;
                            hit[q] = make_float4(__int_as_float(slot), h.t, h.u, h.v);
                            hit2[q] = pack(h.n, __int_as_float(h.inst < 0 ? -1 : (h.inst | (h.front ? RT_HIT2_FRONT : 0))));
                        });
}
// out[q] = on_hit when something blocks the segment (tmax = origin.w), 1 - on_hit otherwise
__global__ __launch_bounds__(RT_TRAV_BLOCK, RT_TL_WAVES) void k_shadow2(SceneView sc, const float4* __restrict__ s_o, const float4* __restrict__ s_d,
                                                      const uint32_t* __restrict__ counts, uint32_t* __restrict__ cursors, uint32_t n_shards,
                                                      uint32_t shard_cap, uint8_t* __restrict__ out, int on_hit) {
    __shared__ int stack[RT_SHORT_STACK * RT_TRAV_BLOCK];
    __shared__ float4 s_top[RT_TOP_LDS_BYTES / 16];
    const int top_bytes = stage_tree_top(sc, s_top);
    trace_queue2<true>(sc, s_o, s_d, counts, cursors, n_shards, shard_cap, (lds_int*)stack + threadIdx.x, nullptr, (lds_cf4*)s_top, top_bytes,
                       [&](uint32_t q, const Hit2&, bool occluded) { out[q] = (uint8_t)(occluded ? on_hit : 1 - on_hit); });
}

// wave64 compaction over the lanes that call it (under divergence: the lanes of the current branch): ONE atomic per
// wave reserves, for every calling lane, `per_lane` entries plus one more for lanes with `extra`; returns the calling
// lane's first entry.  Entries of one lane are contiguous, lanes in lane order.
RT_DEV uint32_t wave_reserve(uint32_t per_lane, bool extra, uint32_t* counter) {
    const unsigned long long all = __ballot(true);
    const unsigned long long ext = __ballot(extra);
    const uint32_t lane = __lane_id();
    const unsigned long long lt = (1ull << lane) - 1ull;
    const uint32_t before = per_lane * (uint32_t)__popcll(all & lt) + (uint32_t)__popcll(ext & lt);
    const uint32_t total = per_lane * (uint32_t)__popcll(all) + (uint32_t)__popcll(ext);
    const int leader = __ffsll((long long)all) - 1;
    uint32_t base = 0;
    if ((int)lane == leader && total) base = atomicAdd(counter, total);
    base = __shfl(base, leader);
    return base + before;
}

#ifdef RT_STATS // section timing of k_shade (debug builds only): cycles and lanes per section, spread over 256 rows
__device__ unsigned long long g_shade_stats[256 * 32];
#define SH_T0 unsigned long long sh_prev_ = __builtin_readcyclecounter();
#define SH_MARK(i)                                                                                         \
    do {                                                                                                   \
        const unsigned long long now_ = __builtin_readcyclecounter();                                      \
        const unsigned long long m_ = __ballot(true);                                                      \
        if ((int)__lane_id() == __ffsll((long long)m_) - 1) {                                              \
            atomicAdd(&g_shade_stats[(blockIdx.x & 255) * 32 + 2 * (i)], now_ - sh_prev_);                 \
            atomicAdd(&g_shade_stats[(blockIdx.x & 255) * 32 + 2 * (i) + 1], (unsigned long long)__popcll(m_)); \
        }                                                                                                  \
        sh_prev_ = now_;                                                                                   \
    } while (0)
#else
#define SH_T0
#define SH_MARK(i)
#endif

#define RT_LDS_CDF 2048 // emissive-triangle CDFs up to this many entries are staged in LDS (16 KB per workgroup)
#define RT_LDS_ENV 1024 // environment maps up to this many texels are staged in LDS (4 KB CDF + 12 KB texels per workgroup)

// One k_shade_t<KEY> launch per material type present in the scene (+ one for rays that left the scene): k_classify
// sorts the bounce's queue entries by the type of the material they hit, so every wave of a shading kernel runs ONE
// material's Sample_Eval / Eval code at full lane occupancy instead of all the types its 64 rays happened to hit one
// after the other (cornellbox: GGX walls, rough-dielectric and Disney-metal objects, emitters -> 4 code paths per wave).
#define RT_KEY_MISS 12
#define RT_N_KEYS 13

// perm[key * capq + i] = queue index of the i-th entry whose hit has material type `key` (RT_KEY_MISS: no hit);
// type_counts[(bounce * 16 + key) * RT_CSTRIDE] = how many.  Persistent workgroups: workgroup b takes every
// (gridDim/n_shards)-th RT_BLOCK group of queue shard b % n_shards, counts its entries per key, reserves its part of
// every key's list with ONE global atomic per key (a counter takes ~88 atomics/us: one per 256 entries was the whole
// cost of this kernel), then walks its groups again and writes the indices.
RT_DEV int classify_key(const SceneView& sc, const Wave& w, uint32_t q) {
    const int slot = __float_as_int(w.hit[q].x);
    if (sc.inst_q) return slot == -1 ? RT_KEY_MISS : (int)sc.inst_type[__float_as_int(w.hit2[q].w) & (RT_HIT2_FRONT - 1)];
    return slot < 0 ? RT_KEY_MISS : (int)sc.tri_type[slot];
}
// Wave priority of the memory-bound kernels (shading, classify, resolve): when they share a CU with the ALU-bound walks of
// another pass in flight, their few instructions between long memory waits should not queue behind the walks' arithmetic.
#ifndef RT_SHADE_PRIO
#define RT_SHADE_PRIO 3
#endif
RT_DEV void memory_bound_priority() {
    if (RT_SHADE_PRIO > 0) __builtin_amdgcn_s_setprio(RT_SHADE_PRIO);
}

__global__ __launch_bounds__(RT_BLOCK) void k_classify(SceneView sc, Wave w, PassParams pp, int bounce) {
    memory_bound_priority();
    __shared__ uint32_t s_tot[16]; // entries of this workgroup per key, then the next free index in the key's list
    const uint32_t lane = __lane_id();
    const uint32_t shard = blockIdx.x % pp.n_shards, sub = blockIdx.x / pp.n_shards, n_sub = gridDim.x / pp.n_shards;
    const uint32_t live = w.counts[(size_t)((uint32_t)bounce * pp.n_shards + shard) * RT_CSTRIDE];
    if (sub * RT_BLOCK >= live) return; // uniform
    const uint32_t q0 = shard * pp.shard_cap;
    if (threadIdx.x < 16) s_tot[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t base = sub * RT_BLOCK; base < live; base += n_sub * RT_BLOCK) {
        const uint32_t i = base + threadIdx.x;
        if (i < live) {
            const int key = classify_key(sc, w, q0 + i);
            for (;;) { // one ballot per key present in the wave; lanes leave as their key is handled
                const int k = __builtin_amdgcn_readfirstlane(key);
                const unsigned long long same = __ballot(key == k);
                if (key == k) {
                    if ((same & ((1ull << lane) - 1ull)) == 0ull) atomicAdd(&s_tot[k], (uint32_t)__popcll(same));
                    break;
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < RT_N_KEYS) {
        const uint32_t n = s_tot[threadIdx.x];
        s_tot[threadIdx.x] = n ? atomicAdd(&w.type_counts[(size_t)(bounce * 16 + (int)threadIdx.x) * RT_CSTRIDE], n) : 0u;
    }
    __syncthreads();
    for (uint32_t base = sub * RT_BLOCK; base < live; base += n_sub * RT_BLOCK) {
        const uint32_t i = base + threadIdx.x;
        if (i < live) {
            const int key = classify_key(sc, w, q0 + i);
            for (;;) {
                const int k = __builtin_amdgcn_readfirstlane(key);
                const unsigned long long same = __ballot(key == k);
                if (key == k) {
                    const uint32_t rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
                    uint32_t first = 0;
                    if (rank == 0) first = atomicAdd(&s_tot[k], (uint32_t)__popcll(same));
                    first = __shfl(first, __ffsll((long long)same) - 1);
                    w.perm[(size_t)k * pp.capq + first + rank] = q0 + i;
                    break;
                }
            }
        }
    }
}

// Shades queue entry q whose hit has material type KEY.  Runs under divergence; wave_reserve() compacts over the
// lanes that reach it, which is all a dense queue needs.  The workgroup appends to output shard `out_shard`.
template <int KEY, bool TL, class TB>
RT_DEV void shade_item(const SceneView& sc, const TB& tb, const Wave& w, const PassParams& pp, int bounce, int last_bounce, uint32_t q, uint32_t out_shard) {
    SH_T0
    const int in = bounce & 1, out = in ^ 1;
    const float4 o4 = w.q_o[in][q], d4 = w.q_d[in][q], tp = w.q_T[in][q], h4 = w.hit[q];
    const int p = __float_as_int(o4.w);
    const V3 ro(o4), rd(d4);
    const int slot = __float_as_int(h4.x);
    V3 T(tp);
    const float scatter_pdf = tp.w;
    SH_MARK(0); // queue entry loads

    if (KEY == RT_KEY_MISS) { // main.cpp:340-359
        if (pp.has_env) {
            float env_pdf;
            V3 bg = env_eval(sc, tb, rd, &env_pdf);
            float mw = 1.f;
            if (bounce > 0) mw = power_heuristic(scatter_pdf, env_pdf);
            w.rad[p] = pack(V3(w.rad[p]) + (T * bg) * mw, 0.f);
        }
        SH_MARK(1); // miss: environment lookup
        return;
    }
    Surf si;
    int mat_id;
    float hit_area; // world.instances[si.index].get_area() of main.cpp:376
    if (TL) { // Instance::intersect, Object.h:380-394: the Geometry's answer in model space, then point and normal to world
        const float4 g4 = w.hit2[q];
        const int iw = __float_as_int(g4.w);
        const float4* iq = sc.inst_q + (size_t)(iw & (RT_HIT2_FRONT - 1)) * RT_INST_QUADS;
        const Xf w2m = load_xf(iq), m2w = load_xf(iq + 4);
        const float4 meta = iq[3];
        const V3 mo = xf_point(w2m, ro), md = xf_model_vector(w2m, rd);
        V3 nm;
        if (slot >= 0) { // TriangleMesh::intersect tail, Object.h:350-355
            const float4* nq = sc.normals + 3 * (size_t)slot;
            V3 N = (1.f - h4.z - h4.w) * V3(nq[0]) + h4.z * V3(nq[1]) + h4.w * V3(nq[2]);
            N = normalize(N);
            si.front_face = dot(md, N) < 0.f;
            nm = si.front_face ? N : -N;
            si.u = si.v = 0.f; // TriangleMesh::intersect writes no si->u / si->v (Object.h:346-356)
        } else {
            nm = V3(g4);
            si.front_face = (iw & RT_HIT2_FRONT) != 0;
            si.u = h4.z; // the Rectangle's (Object.h:116-117); 0 for the other primitives
            si.v = h4.w;
        }
        si.point = xf_point(m2w, mo + md * h4.y);
        si.normal = normalize(xf_world_normal(w2m, nm));
        mat_id = __float_as_int(meta.y);
        hit_area = meta.w;
    } else {
        // TriangleMesh::intersect tail, Object.h:350-355
        // the device copy of the normals record carries the triangle's material and light index in its pad words, so
        // shading reads one 48-byte record per hit (every divergent load instruction costs 64 L1 look-ups)
        const float4* nq = sc.normals + 3 * (size_t)slot;
        const float4 n0 = nq[0], n1 = nq[1], n2 = nq[2];
        mat_id = __float_as_int(n0.w);
        const int light_id = __float_as_int(n1.w);
        si.point = ro + rd * h4.y;
        V3 N = (1.f - h4.z - h4.w) * V3(n0) + h4.z * V3(n1) + h4.w * V3(n2);
        N = normalize(N);
        si.front_face = dot(rd, N) < 0.f;
        si.normal = si.front_face ? N : -N;
        si.u = si.v = 0.f;
        hit_area = light_id >= 0 ? sc.lights[light_id].area : 0.f; // Geometry::area() default, Object.h:18-20
    }
    rt_material mat = sc.materials[mat_id];
    mat.type = KEY; // compile-time constant: the material switches below fold to this kernel's one case

    const float4 pv = w.q_p[in][q];
    Rng rng;
    rng.key = __float_as_uint(pv.w); // rng_key(pixel, sample, seed), stored by k_generate
    rng.ctr = __float_as_uint(d4.w);

    const V3 emitted = mat_emitted(mat);
    float weight = 1.f;
    if (bounce > 0 && dot(emitted, V3(1.0f)) > 0.1f) { // main.cpp:373-396
        float cos_light = fmax_sel(dot(rd, -si.normal), 0.f);
        float denom = hit_area * cos_light;
        if (denom <= 0.f || !si.front_face) {
            weight = 0.f;
        } else {
            float lpdf = len2(V3(pv) - si.point) / denom;
            weight = power_heuristic(scatter_pdf, lpdf);
        }
    }
    // radiance += weight * emitted * throughput (main.cpp:401).  Adding an exact zero leaves the radiance as it is (it is
    // never -0), so only emitters touch the path's radiance here; a NaN (0 * inf throughput) fails the test and is added.
    const V3 add = (weight * emitted) * T;
    if (!(add.x == 0.f && add.y == 0.f && add.z == 0.f)) w.rad[p] = pack(V3(w.rad[p]) + add, 0.f);
    SH_MARK(2); // surface, material fetch, emission

    Scatter scat = mat_sample(mat, si, rd, rng);
    SH_MARK(3); // Sample_Eval
    if (fabsf(scat.pdf) < 0.000001f) return; // main.cpp:410-412: the path ends BEFORE next-event estimation

    // shade_direct_NEE, main.cpp:240-317: contributions are computed here, the visibility
    // tests run in k_shadow over a dense queue, k_nee_resolve applies them in slot order.
    const V3 sh_origin = si.point + si.normal * 0.0001f;
    V3 env_dir(0.f), env_c(0.f);
    bool want_env = false;
    if (pp.has_env) {
        float env_pdf;
        V3 col = env_sample(sc, tb, rng.next(), &env_dir, &env_pdf);
        want_env = dot(env_dir, si.normal) > 0.f;
        if (want_env) {
            float bp;
            V3 f = mat_eval(mat, si, rd, env_dir, &bp);
            float mis = power_heuristic(env_pdf, bp);
            env_c = ((mis * f) * col) / env_pdf;
        }
    }
    SH_MARK(4); // environment NEE
    // one reservation for all of this wave's shadow rays: [env ray] + one per light, contiguous per lane.  A slot
    // without a ray (the environment direction below the surface) contributes an exact zero and is simply left out.
    const uint32_t n_rays = (uint32_t)sc.n_lights + (want_env ? 1u : 0u);
    uint32_t j = out_shard * (uint32_t)pp.K * pp.shard_cap +
                 wave_reserve((uint32_t)sc.n_lights, want_env, &w.counts[(size_t)(64u * pp.n_shards + (uint32_t)bounce * pp.n_shards + out_shard) * RT_CSTRIDE]);
    if (n_rays) w.s_T[j] = pack(T, __uint_as_float(n_rays)); // T before the scatter (main.cpp:430 uses it before 434 updates it)
    uint32_t first_tag = (uint32_t)p + 1u;
    if (want_env) {
        w.s_o[j] = pack(sh_origin, INFINITY);
        w.s_d[j] = pack(env_dir, 0.f);
        w.s_c[j] = pack(env_c, __uint_as_float(first_tag));
        first_tag = 0u;
        j++;
    }
    for (int i = 0; i < sc.n_lights; i++, j++) {
        const rt_light& L = sc.lights[__builtin_amdgcn_readfirstlane(i)]; // wave-uniform: read through the scalar cache
        V3 point, normal;
        light_sample(sc, tb, L, rng, &point, &normal);
        V3 ld = point - si.point;
        float length = len(ld);
        ld = ld / length;
        V3 c(0.f);
        float cos_light = fmax_sel(dot(-normal, ld), 0.f);
        float denom = cos_light * L.area;
        if (!(fabsf(denom) < 0.00001f)) {
            float lpdf = (length * length) / denom;
            float bp;
            V3 f = mat_eval(mat, si, rd, ld, &bp);
            float mis = power_heuristic(lpdf, bp);
            c = (((mis * f) / lpdf) * 1.f) * mat_emitted(sc.materials[L.material]);
        }
        // the reference traces this ray before it looks at denom (main.cpp:281-295)
        w.s_o[j] = pack(sh_origin, length - 0.001f);
        w.s_d[j] = pack(ld, 0.f);
        w.s_c[j] = pack(c, __uint_as_float(first_tag));
        first_tag = 0u;
    }
    SH_MARK(5); // light NEE

    T = (T * scat.f) / scat.pdf; // main.cpp:434
    if (!last_bounce) { // next-bounce ray queue, compacted over the surviving lanes; the path's state goes with the ray
        const uint32_t qi = out_shard * pp.shard_cap + wave_reserve(1u, false, &w.counts[(size_t)((uint32_t)(bounce + 1) * pp.n_shards + out_shard) * RT_CSTRIDE]);
        w.q_o[out][qi] = pack(scat.origin, __int_as_float(p));
        w.q_d[out][qi] = pack(scat.dir, __uint_as_float(rng.ctr));
        w.q_T[out][qi] = pack(T, scat.pdf);
        w.q_p[out][qi] = pack(si.point, pv.w);
    }
    SH_MARK(6); // path state + next ray
}


struct ShadeOrder { // the shading kernels of a bounce in launch order
    int32_t n;
    int32_t keys[RT_N_KEYS];
};

#ifndef RT_SHADE_WAVES
#define RT_SHADE_WAVES 1
#endif
#ifndef RT_SHADE_WAVES_TL
#define RT_SHADE_WAVES_TL 1
#endif
template <int KEY, bool TL>
__global__ __launch_bounds__(RT_BLOCK, TL ? RT_SHADE_WAVES_TL : RT_SHADE_WAVES) void k_shade_t(SceneView sc_in, Wave w, PassParams pp, ShadeOrder order, int bounce, int last_bounce) {
    // The sampling CDFs are walked by dependent binary searches (7 + 10 steps on cornellbox) and the environment is read
    // with four bilinear taps: from LDS when all the tables fit (wave-uniform choice between two instantiations).
    __shared__ float s_env_cdf[RT_LDS_ENV];
    __shared__ float s_env_rgb[RT_LDS_ENV * 3];
    __shared__ float s_emissive[RT_LDS_CDF * 2];
    memory_bound_priority();
    const uint32_t n = w.type_counts[(size_t)(bounce * 16 + KEY) * RT_CSTRIDE];
    if (blockIdx.x * RT_BLOCK >= n) return; // uniform: fewer entries than resident workgroups
    const SceneView& sc = sc_in;
    const int n_env = sc.env_cdf ? sc.env_w * sc.env_h : 0;
    const bool staged = n_env <= RT_LDS_ENV && sc.n_emissive <= RT_LDS_CDF;
    if (staged && KEY != RT_MAT_EMISSIVE) { // (emitters never reach next-event estimation and are never missed)
        for (int i = threadIdx.x; i < 3 * n_env; i += RT_BLOCK) s_env_rgb[i] = sc.env_rgb[i];
        if (KEY != RT_KEY_MISS) {
            for (int i = threadIdx.x; i < n_env; i += RT_BLOCK) s_env_cdf[i] = sc.env_cdf[i];
            const float* em = (const float*)sc.emissive;
            for (int i = threadIdx.x; i < 2 * sc.n_emissive; i += RT_BLOCK) s_emissive[i] = em[i];
        }
        __syncthreads();
    }
    // Output shards are filled round-robin by RT_BLOCK-entry groups numbered across all of the bounce's shading
    // kernels (this kernel's groups follow those of the kernels launched before it), so no shard takes more than
    // ceil(groups / n_shards) groups: at most (live entries / RT_BLOCK + one partial group per kernel) groups exist,
    // which the spare group per shard in shard_cap covers.
    uint32_t group0 = 0;
    for (int k = 0; k < order.n && order.keys[k] != KEY; k++)
        group0 += (w.type_counts[(size_t)(bounce * 16 + order.keys[k]) * RT_CSTRIDE] + RT_BLOCK - 1) / RT_BLOCK;
    const uint32_t* perm = w.perm + (size_t)KEY * pp.capq;
    if (staged) {
        LdsTables tb;
        tb.cdf = (lds_cfloat*)s_env_cdf;
        tb.rgb = (lds_cfloat*)s_env_rgb;
        tb.em = (lds_cfloat*)s_emissive;
        for (uint32_t base = blockIdx.x * RT_BLOCK; base < n; base += gridDim.x * RT_BLOCK) { // persistent, grid-stride
            const uint32_t i = base + threadIdx.x;
            const uint32_t out_shard = (group0 + base / RT_BLOCK) % pp.n_shards;
            if (i < n) shade_item<KEY, TL>(sc, tb, w, pp, bounce, last_bounce, perm[i], out_shard);
        }
    } else {
        GlobalTables tb;
        tb.cdf = sc.env_cdf;
        tb.rgb = sc.env_rgb;
        tb.em = sc.emissive;
        for (uint32_t base = blockIdx.x * RT_BLOCK; base < n; base += gridDim.x * RT_BLOCK) {
            const uint32_t i = base + threadIdx.x;
            const uint32_t out_shard = (group0 + base / RT_BLOCK) % pp.n_shards;
            if (i < n) shade_item<KEY, TL>(sc, tb, w, pp, bounce, last_bounce, perm[i], out_shard);
        }
    }
}

// radiance += throughput * direct (main.cpp:430), a path's slots summed in reference order (environment ray, then the
// lights).  One lane per shadow-queue entry; the lane of a path's FIRST ray does the path's sum (its rays are consecutive).
// Persistent workgroups, as k_classify: workgroup b takes every (gridDim / n_shards)-th RT_BLOCK group of the LIVE entries of
// shard b % n_shards (a launch at the queue's capacity — 33 M threads at 1080p x 8 spp x 2 rays — spent 0.12 ms per launch
// on workgroups that found nothing to do, 32 launches per step).
__global__ __launch_bounds__(RT_BLOCK) void k_nee_resolve(Wave w, PassParams pp, int bounce) {
    memory_bound_priority();
    const uint32_t shard = blockIdx.x % pp.n_shards, sub = blockIdx.x / pp.n_shards, n_sub = gridDim.x / pp.n_shards;
    const uint32_t shard_span = (uint32_t)pp.K * pp.shard_cap; // shadow-queue entries reserved per shard
    const uint32_t live = w.counts[(size_t)(64u * pp.n_shards + (uint32_t)bounce * pp.n_shards + shard) * RT_CSTRIDE];
    for (uint32_t i = sub * RT_BLOCK + threadIdx.x; i < live; i += n_sub * RT_BLOCK) {
        const uint32_t j = shard * shard_span + i;
        const float4 c0 = w.s_c[j];
        const int p = (int)__float_as_uint(c0.w) - 1;
        if (p < 0) continue; // not a path's first ray
        const float4 T4 = w.s_T[j];
        const uint32_t n = __float_as_uint(T4.w);
        V3 direct(0.f);
        if (w.vis[j]) direct = direct + V3(c0);
        for (uint32_t r = 1; r < n; r++)
            if (w.vis[j + r]) direct = direct + V3(w.s_c[j + r]);
        // nothing arrives (every ray blocked, or lights seen from behind): radiance + 0 is radiance (it is never -0: it starts
        // at +0 and the NaN / inf cases fail this test), so the path record — a random 64-byte sector — is left alone
        const V3 add = V3(T4) * direct;
        if (add.x == 0.f && add.y == 0.f && add.z == 0.f) continue;
        w.rad[p] = pack(V3(w.rad[p]) + add, 0.f);
    }
}

// pixel_total += s with the NaN guard of main.cpp:984-987, samples in index order
__global__ __launch_bounds__(RT_BLOCK) void k_accumulate(Wave w, PassParams pp, float* __restrict__ sum) {
    const int pix = blockIdx.x * RT_BLOCK + threadIdx.x;
    for (int i = pix; i < RT_COUNTS; i += gridDim.x * RT_BLOCK) w.counts_dense[i] = w.counts[(size_t)i * RT_CSTRIDE];
    if (pix >= pp.npix) return;
    int x, y;
    pixel_of(pp, pix, &x, &y);
    float* px = sum + ((size_t)y * pp.width + x) * 3;
    V3 total(px[0], px[1], px[2]);
    for (int s = 0; s < pp.spp_pass; s++) {
        V3 v(w.rad[(size_t)s * pp.npix + pix]);
        if (v.x != v.x) v = V3(0.f);
        total = total + v;
    }
    px[0] = total.x;
    px[1] = total.y;
    px[2] = total.z;
}

// Renderer::add_to_buffer, main.cpp:865-877 (+ the row flip of main.cpp:990)
__global__ __launch_bounds__(RT_BLOCK) void k_resolve(const float* __restrict__ sum, int w, int h, float scale, float inv_gamma,
                                                      uint8_t* __restrict__ rgb) {
    const int i = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= w * h) return;
    const int x = i % w, y = i / w;
    uint8_t* o = rgb + ((size_t)(h - y - 1) * w + x) * 3;
    for (int c = 0; c < 3; c++) {
        float v = sum[(size_t)i * 3 + c] * scale;
        v = powf(v, inv_gamma);
        o[c] = (uint8_t)(clampf(v, 0.f, 1.f) * 255.f);
    }
}

// ---- rt_trace_closest_noalloc: the scratch that rt_trace_closest allocates per call — the two
// float4 ray streams the walk reads, the shard counters and fetch cursors, the second hit record of two-level scenes —
// lives in ONE file-scope device array.  Every GPU the library runs on holds its own copy at its own address until the
// process ends; host code never uses the array's name as an address: scene_create resolves it per device with
// hipGetSymbolAddress and everything below goes through that pointer.
#define RT_NOALLOC_COUNT_BYTES (2 * RT_SHARDS * RT_CSTRIDE * 4) // counts, then cursors
#define RT_NOALLOC_BYTES ((size_t)RT_NOALLOC_COUNT_BYTES + (size_t)3 * 16 * RT_NOALLOC_MAX_RAYS)
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_BYTES];

// rt_ray[n] (24-byte records, device memory) [+ tmax[n]] -> the walk's two float4 streams; the first RT_SHARDS threads also set the
// shard counters and clear the fetch cursors (the array keeps the previous call's contents).
__global__ __launch_bounds__(RT_BLOCK) void k_stage_rays(const float* __restrict__ rays, const float* __restrict__ tmax, int64_t n, uint32_t chunk,
                                                         float4* __restrict__ o, float4* __restrict__ d, uint32_t* __restrict__ counts) {
    const int64_t i = (int64_t)blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i < RT_SHARDS) {
        counts[(size_t)i * RT_CSTRIDE] = dealt_count((uint64_t)n, chunk, RT_SHARDS, (uint32_t)i);
        counts[(size_t)(RT_SHARDS + i) * RT_CSTRIDE] = 0u;
    }
    if (i >= n) return;
    const float* r = rays + 6 * i;
    o[i] = make_float4(r[0], r[1], r[2], tmax ? tmax[i] : INFINITY);
    d[i] = make_float4(r[3], r[4], r[5], 0.f);
}

// -------------------------------------------------------------- host side
#define RT_MAX_SETS 4
#define RT_MAX_DEVICES 64
static void* g_scratch_ptr[RT_MAX_DEVICES] = {}; // device address of `scratch` on each GPU (set once per device by scene_create)
struct rt_scene {
    int device = 0;
    unsigned persistent_blocks = 16;  // resident traversal workgroups: 32 waves per CU, a multiple of RT_SHARDS waves in all
    hipStream_t stream = nullptr;
    SceneView view{};
    int64_t n_pairs = 0, n_tris = 0;
    int32_t n_materials = 0, n_lights = 0;
    std::vector<int> keys; // material types some triangle uses, most triangles first, then RT_KEY_MISS
    unsigned shade_blocks = 64; // resident workgroups of a shading launch (a multiple of RT_SHARDS)
    unsigned resolve_blocks = 64; // ... of a k_nee_resolve launch (a streaming kernel: 8 small workgroups per CU)
    std::vector<void*> owned; // scene arrays
    // wavefront buffers: two sets, so that consecutive sample passes of one rt_render call can run on two streams
    Wave wave[RT_MAX_SETS] = {};
    int64_t wave_cap = 0;
    int wave_K = 0;
    int wave_sets = 0;
    std::vector<void*> wave_owned;
    hipStream_t streams[RT_MAX_SETS] = {}; // streams[0] == stream
    hipEvent_t ev_fork = nullptr, ev_acc[RT_MAX_SETS] = {};
    int max_sets = 4; // sample passes in flight (measured on cornellbox 1080p: 2 sets 4.16, 3 sets 4.05, 4 sets 4.37 Grays/s)
    uint32_t* h_counts = nullptr; // pinned: per-pass counter read-back must not block the launching thread
    size_t h_counts_n = 0;
    unsigned pipe_blocks = 16; // traversal workgroups per launch while two passes share the GPU
    float* d_sum = nullptr;
    int64_t d_sum_n = 0;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    std::vector<hipEvent_t> ev_pool;
    unsigned char* scratch_ptr = nullptr; // this GPU's copy of `scratch` (rt_trace_closest_noalloc)
    // rt_render_tiles: device copy of the tile table and the host copy it was made from (an unchanged list is neither
    // checked nor uploaded again)
    int4* d_tiles = nullptr;
    size_t d_tiles_cap = 0;
    unsigned char* api_mem = nullptr; // rt_trace_*'s device buffers: one allocation, grown on demand, kept up to RT_API_KEEP_BYTES
    size_t api_cap = 0;
    std::vector<int32_t> tiles_host; // x0, y0, x1, y1 per tile, as the caller gave them
    int64_t tiles_npix = 0;
    int32_t tiles_w = 0, tiles_h = 0; // raster the cached list was checked against
};

template <typename T>
static int upload(rt_scene* s, const T* host, size_t n, const T** dev) {
    *dev = nullptr;
    if (n == 0) return RT_OK;
    void* d = nullptr;
    HIPCHK(hipMalloc(&d, n * sizeof(T)));
    s->owned.push_back(d);
    HIPCHK(hipMemcpy(d, host, n * sizeof(T), hipMemcpyHostToDevice));
    *dev = (const T*)d;
    return RT_OK;
}

static void free_wave(rt_scene* s) {
    for (void* p : s->wave_owned) (void)hipFree(p);
    s->wave_owned.clear();
    s->wave_cap = 0;
    s->wave_sets = 0;
}

template <typename T>
static int wave_alloc(rt_scene* s, T** out, size_t n) {
    void* d = nullptr;
    HIPCHK(hipMalloc(&d, n * sizeof(T)));
    s->wave_owned.push_back(d);
    *out = (T*)d;
    return RT_OK;
}

static int ensure_wave(rt_scene* s, int64_t cap, int K, int sets) {
    if (s->wave_cap >= cap && s->wave_K >= K && s->wave_sets >= sets) return RT_OK;
    free_wave(s);
    const size_t c = (size_t)cap, kc = (size_t)cap * (size_t)(K > 0 ? K : 1);
    {   // a caller who raises spp_per_pass (or renders a huge tile) gets an error code, not an allocation failure half-way:
        // per queue entry 8 + 1 (+1 two-level) + 1 float4 streams and 13 permutation words, per shadow slot 4 float4 + 1 byte
        const size_t per_set = c * (16 * (8 + 1 + (s->view.inst_q ? 1 : 0) + 1) + 4 * RT_N_KEYS) + kc * (4 * 16 + 1) + (size_t)4 * RT_COUNTS * (3 * RT_CSTRIDE + 1);
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && per_set * (size_t)sets > free_b)
            return fail(RT_ERR_LIMIT, "rt_render: the wavefront buffers of this call need " + std::to_string(per_set * (size_t)sets >> 20) + " MiB (" +
                                          std::to_string(sets) + " passes in flight x " + std::to_string(cap) + " queue entries), " +
                                          std::to_string(free_b >> 20) + " MiB are free: lower spp_per_pass or render the frame in tiles");
    }
    int rc;
    for (int set = 0; set < sets; set++) {
    Wave& w = s->wave[set];
    for (int i = 0; i < 2; i++) {
        if ((rc = wave_alloc(s, &w.q_o[i], c))) return rc;
        if ((rc = wave_alloc(s, &w.q_d[i], c))) return rc;
        if ((rc = wave_alloc(s, &w.q_T[i], c))) return rc;
        if ((rc = wave_alloc(s, &w.q_p[i], c))) return rc;
    }
    if ((rc = wave_alloc(s, &w.hit, c))) return rc;
    w.hit2 = nullptr;
    if (s->view.inst_q && (rc = wave_alloc(s, &w.hit2, c))) return rc;
    if ((rc = wave_alloc(s, &w.rad, c))) return rc;
    if ((rc = wave_alloc(s, &w.s_c, kc))) return rc;
    if ((rc = wave_alloc(s, &w.s_T, kc))) return rc;
    if ((rc = wave_alloc(s, &w.vis, kc))) return rc;
    if ((rc = wave_alloc(s, &w.s_o, kc))) return rc;
    if ((rc = wave_alloc(s, &w.s_d, kc))) return rc;
    if ((rc = wave_alloc(s, &w.counts, (size_t)3 * RT_COUNTS * RT_CSTRIDE))) return rc;
    w.cursors = w.counts + (size_t)RT_COUNTS * RT_CSTRIDE;
    w.type_counts = w.counts + (size_t)2 * RT_COUNTS * RT_CSTRIDE;
    if ((rc = wave_alloc(s, &w.perm, c * RT_N_KEYS))) return rc;
    if ((rc = wave_alloc(s, &w.counts_dense, (size_t)RT_COUNTS))) return rc;
    }
    s->wave_cap = cap;
    s->wave_K = K;
    s->wave_sets = sets;
    return RT_OK;
}

static void launch_shade(rt_scene* s, hipStream_t st, const Wave& wave, int key, const PassParams& pp, int bounce, int last_bounce) {
    const dim3 g(s->shade_blocks), t(RT_BLOCK);
    ShadeOrder order{};
    order.n = (int32_t)s->keys.size();
    for (size_t i = 0; i < s->keys.size(); i++) order.keys[i] = s->keys[i];
#define RT_SHADE_CASE(K)                                                                                           \
    case K:                                                                                                        \
        if (s->view.inst_q) hipLaunchKernelGGL((k_shade_t<K, true>), g, t, 0, st, s->view, wave, pp, order, bounce, last_bounce); \
        else hipLaunchKernelGGL((k_shade_t<K, false>), g, t, 0, st, s->view, wave, pp, order, bounce, last_bounce);  \
        break;
    switch (key) {
        RT_SHADE_CASE(0) RT_SHADE_CASE(1) RT_SHADE_CASE(2) RT_SHADE_CASE(3) RT_SHADE_CASE(4) RT_SHADE_CASE(5) RT_SHADE_CASE(6)
        RT_SHADE_CASE(7) RT_SHADE_CASE(8) RT_SHADE_CASE(9) RT_SHADE_CASE(10) RT_SHADE_CASE(11) RT_SHADE_CASE(12)
    }
#undef RT_SHADE_CASE
}

static inline unsigned blocks_for(int64_t n) { return (unsigned)((n + RT_BLOCK - 1) / RT_BLOCK); }

extern "C" {

const char* rt_last_error(void) { return g_err_lit ? g_err_lit : g_err.c_str(); }

int rt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// Depth (inner nodes on the longest root-to-leaf path) of the tree of pair records under `root`; -1 when a record is
// reached twice (a cycle, or subtrees shared between parents: the walks assume a tree).  A walk's stack holds at most one
// entry per inner node on the current path, so a tree within the reference's own stack depth (Object.h:275, Scene.cpp:19)
// cannot overrun it.  seen: one flag per record, shared by all the roots of a scene.
static int pair_tree_depth(const rt_bvh_pair* pairs, int64_t n_pairs, int32_t root, std::vector<uint8_t>& seen) {
    if (root < 0) return 0;
    int depth = 0;
    std::vector<std::pair<int32_t, int>> todo;
    todo.push_back({root, 1});
    while (!todo.empty()) {
        const auto [r, d] = todo.back();
        todo.pop_back();
        if (r < 0 || r >= n_pairs) continue; // (leaf; ranges were checked by the caller)
        if (seen[(size_t)r]) return -1;
        seen[(size_t)r] = 1;
        depth = std::max(depth, d);
        todo.push_back({pairs[r].lref, d + 1});
        todo.push_back({pairs[r].rref, d + 1});
    }
    return depth;
}

// The compressed four-wide collapse of the tree of pair records under `root` (rt_device.h, "four-wide occlusion walk").
// Quad records are appended to `out` (16 words each) in breadth-first order, so the top of the tree leads the array; the
// exact box of every leaf goes to leaf_box[6 * first slot].  Leaf refs are written as ~slot and resolved by the caller
// once the number of quad records is known.  Returns the byte offset of the root record, and false in *ok when the tree
// must stay on the pair walk:
//   * a box that does not lie inside its parent's box, or is not finite (the argument that the quad walk reaches the same
//     leaves needs nested boxes; BVH::build makes them so, a caller's own records need not),
//   * a worst-case stack beyond RT_QSTACK_TOTAL entries, or records beyond 2 GiB.
struct QuadChild {
    float mn[3], mx[3];
    int32_t ref; // pair-record encoding
};
// plane q of a quad's grid as the device decodes it (rt_device.h, quad_axis): ONE rounding
static inline float quad_plane(int q, float step, float org) { return std::fmaf((float)q, step, org); }
static int32_t build_quads(const rt_bvh_pair* pairs, int32_t root, std::vector<uint32_t>& out, std::vector<float>& leaf_box, bool* ok, int* stack_need) {
    *ok = true;
    *stack_need = 0;
    auto finite3 = [](const float* v) { return std::isfinite(v[0]) && std::isfinite(v[1]) && std::isfinite(v[2]); };
    auto inside = [&](const QuadChild& c, const QuadChild& parent) {
        for (int a = 0; a < 3; a++)
            if (!(c.mn[a] >= parent.mn[a] && c.mx[a] <= parent.mx[a])) return false;
        return true;
    };
    auto children_of = [&](int32_t pr, QuadChild* l, QuadChild* r) {
        const rt_bvh_pair& p = pairs[pr];
        std::memcpy(l->mn, p.lmin, 12);
        std::memcpy(l->mx, p.lmax, 12);
        std::memcpy(r->mn, p.rmin, 12);
        std::memcpy(r->mx, p.rmax, 12);
        l->ref = p.lref;
        r->ref = p.rref;
        auto ordered = [](const QuadChild& c) { return c.mn[0] <= c.mx[0] && c.mn[1] <= c.mx[1] && c.mn[2] <= c.mx[2]; };
        return finite3(l->mn) && finite3(l->mx) && finite3(r->mn) && finite3(r->mx) && ordered(*l) && ordered(*r);
    };
    auto area = [](const QuadChild& c) {
        const double x = (double)c.mx[0] - c.mn[0], y = (double)c.mx[1] - c.mn[1], z = (double)c.mx[2] - c.mn[2];
        return x * y + y * z + z * x;
    };
    const size_t first = out.size() / 16;
    std::vector<int32_t> todo{root};      // pair record each quad starts from; quad k of this tree = todo[k]
    std::vector<uint8_t> n_children;      // per quad
    std::vector<int32_t> child_quads;     // per quad, 4 entries: index (within this tree) of the child quad, or -1
    for (size_t k = 0; k < todo.size(); k++) {
        if ((first + todo.size() + 4) * RT_ONODE_BYTES > (size_t)0x7fffffff) {
            *ok = false;
            return 0;
        }
        QuadChild c[4];
        int n = 2;
        if (!children_of(todo[k], &c[0], &c[1])) *ok = false;
        while (*ok && n < 4) { // open the inner child with the largest surface
            int pick = -1;
            for (int i = 0; i < n; i++)
                if (c[i].ref >= 0 && (pick < 0 || area(c[i]) > area(c[pick]))) pick = i;
            if (pick < 0) break;
            QuadChild l, r;
            if (!children_of(c[pick].ref, &l, &r) || !inside(l, c[pick]) || !inside(r, c[pick])) *ok = false;
            c[pick] = l;
            c[n++] = r;
        }
        if (!*ok) return 0;
        uint32_t rec[16] = {};
        // the record's grid: origin = the children's common min (an exact fp32 value: plane 0 decodes to it), step = a power
        // of two with plane 255 on or beyond the common max AS DECODED
        uint32_t lo[3][4] = {}, hi[3][4] = {};
        for (int a = 0; a < 3; a++) {
            float org = c[0].mn[a], top = c[0].mx[a];
            for (int i = 1; i < n; i++) {
                org = std::min(org, c[i].mn[a]);
                top = std::max(top, c[i].mx[a]);
            }
            int e = -126;
            const double ext = (double)top - (double)org;
            if (ext > 0) e = std::max(-126, (int)std::ceil(std::log2(ext / 255.0)));
            for (;; e++) {
                if (e > 127) { // (extents near FLT_MAX: no grid step holds them)
                    *ok = false;
                    return 0;
                }
                const float step = std::ldexp(1.0f, e);
                if (!(quad_plane(255, step, org) >= top)) continue;
                if (!std::isfinite(quad_plane(255, step, org))) { // (boxes near FLT_MAX: a decoded plane must stay a number)
                    *ok = false;
                    return 0;
                }
                bool fits = true;
                for (int i = 0; i < n && fits; i++) {
                    // largest q with plane(q) <= min; smallest q with plane(q) >= max (plane() is non-decreasing in q)
                    int ql = (int)std::floor(((double)c[i].mn[a] - (double)org) / (double)step);
                    ql = std::max(0, std::min(255, ql));
                    while (ql > 0 && !(quad_plane(ql, step, org) <= c[i].mn[a])) ql--;
                    while (ql < 255 && quad_plane(ql + 1, step, org) <= c[i].mn[a]) ql++;
                    int qh = (int)std::ceil(((double)c[i].mx[a] - (double)org) / (double)step);
                    qh = std::max(0, std::min(255, qh));
                    while (qh < 255 && !(quad_plane(qh, step, org) >= c[i].mx[a])) qh++;
                    while (qh > 0 && quad_plane(qh - 1, step, org) >= c[i].mx[a]) qh--;
                    if (!(quad_plane(ql, step, org) <= c[i].mn[a]) || !(quad_plane(qh, step, org) >= c[i].mx[a])) fits = false;
                    lo[a][i] = (uint32_t)ql;
                    hi[a][i] = (uint32_t)qh;
                }
                if (fits) break;
            }
            std::memcpy(&rec[a], &org, 4);
            rec[3] |= (uint32_t)(e + 127) << (8 * a);
            for (int i = 0; i < n; i++) {
                rec[4 + a] |= lo[a][i] << (8 * i);
                rec[7 + a] |= hi[a][i] << (8 * i);
            }
        }
        for (int i = 0; i < 4; i++) {
            int32_t ref;
            if (i >= n) {
                ref = RT_REF_NONE;
                child_quads.push_back(-1);
            } else if (c[i].ref >= 0) {
                child_quads.push_back((int32_t)todo.size());
                ref = (int32_t)((first + todo.size()) * RT_ONODE_BYTES);
                todo.push_back(c[i].ref);
            } else {
                child_quads.push_back(-1);
                ref = c[i].ref; // ~slot: the caller turns it into the leaf record's ref
                float* lb = &leaf_box[(size_t)6 * (size_t)(~c[i].ref)];
                std::memcpy(lb, c[i].mn, 12);
                std::memcpy(lb + 3, c[i].mx, 12);
            }
            std::memcpy(&rec[10 + i], &ref, 4);
        }
        out.insert(out.end(), rec, rec + 16);
        n_children.push_back((uint8_t)n);
    }
    // worst-case stack: while a walk is below child i of a quad, up to (children - 1) siblings wait on the stack.  Children
    // have larger numbers than their parent (breadth-first), so one backward sweep does it.
    std::vector<int32_t> need(todo.size(), 0);
    for (size_t k = todo.size(); k-- > 0;) {
        int32_t deepest = 0;
        for (int i = 0; i < 4; i++)
            if (child_quads[4 * k + i] >= 0) deepest = std::max(deepest, need[(size_t)child_quads[4 * k + i]]);
        need[k] = (int32_t)n_children[k] - 1 + deepest;
    }
    *stack_need = need[0];
    if (need[0] > RT_QSTACK_TOTAL) *ok = false;
    return (int32_t)(first * RT_ONODE_BYTES);
}

// The whole onode buffer of a flat scene: quad records, then one leaf record per triangle slot.
static bool build_onodes(const rt_scene_desc* d, std::vector<uint32_t>& out, int32_t* qroot, int32_t* n_quads, int32_t* leaf_base, int* stack_need) {
    *stack_need = 0;
    if (d->root_ref < 0 || d->n_pairs <= 0) return false; // a one-leaf tree has nothing to collapse
    std::vector<float> leaf_box((size_t)6 * (size_t)d->n_tris, 0.f);
    bool ok = false;
    *qroot = build_quads(d->pairs, d->root_ref, out, leaf_box, &ok, stack_need);
    if (!ok) return false;
    const size_t nq = out.size() / 16;
    if ((nq + (size_t)d->n_tris + 1) * RT_ONODE_BYTES > (size_t)0x7fffffff) return false; // refs are 31-bit byte offsets
    *n_quads = (int32_t)nq;
    *leaf_base = (int32_t)(nq * RT_ONODE_BYTES);
    for (size_t k = 0; k < nq; k++)
        for (int i = 0; i < 4; i++) {
            int32_t ref;
            std::memcpy(&ref, &out[16 * k + 10 + i], 4);
            if (ref < 0 && ref != RT_REF_NONE) {
                ref = (int32_t)(*leaf_base + (int64_t)(~ref) * RT_ONODE_BYTES) | RT_REF_LEAF;
                std::memcpy(&out[16 * k + 10 + i], &ref, 4);
            }
        }
    out.resize((nq + (size_t)d->n_tris) * 16);
    for (int64_t t = 0; t < d->n_tris; t++) {
        uint32_t* rec = &out[(nq + (size_t)t) * 16];
        std::memcpy(rec, &leaf_box[(size_t)6 * (size_t)t], 24);
        std::memcpy(rec + 6, d->tris[t].v0, 12);
        std::memcpy(rec + 9, d->tris[t].e1, 12);
        std::memcpy(rec + 12, d->tris[t].e2, 12);
        rec[15] = (d->tris[t].tri_id & RT_TRI_LEAF_END) ? 1u : 0u;
    }
    return true;
}

// tl == nullptr: the flat scene of rt_scene_create; otherwise the two-level scene of rt_scene_create_instanced
static int scene_create(const rt_scene_desc* d, const rt_tlas_desc* tl, int device, rt_scene** out) {
    if (!d || !out) return fail(RT_ERR_INVALID, "rt_scene_create: null argument");
    *out = nullptr;
    if (!tl && d->n_tris <= 0) return fail(RT_ERR_INVALID, "rt_scene_create: scene has no triangles");
    if (d->n_tris < 0 || (d->n_tris > 0 && (!d->tris || !d->normals))) return fail(RT_ERR_INVALID, "rt_scene_create: bad triangle table");
    if (d->n_pairs < 0 || (d->n_pairs > 0 && !d->pairs)) return fail(RT_ERR_INVALID, "rt_scene_create: bad pair table");
    if (d->n_pairs > RT_MAX_PAIRS) return fail(RT_ERR_LIMIT, "rt_scene_create: more than 2^25 - 1 child-pair records");
    if (d->n_tris > (int64_t)0xffffffffu / 48) return fail(RT_ERR_LIMIT, "rt_scene_create: more than 89 million triangles");
    if (d->n_lights > RT_MAX_LIGHTS) return fail(RT_ERR_LIMIT, "rt_scene_create: more than 15 area lights");
    if (d->n_materials <= 0 || !d->materials) return fail(RT_ERR_INVALID, "rt_scene_create: no materials");
    // every reference a kernel will follow is checked here, on the host, so that a bad
    // description is an error code and not a GPU memory fault
    if (!tl && (d->root_ref >= 0 ? d->root_ref >= d->n_pairs : ~d->root_ref >= d->n_tris))
        return fail(RT_ERR_INVALID, "rt_scene_create: root_ref out of range");
    for (int64_t i = 0; i < d->n_pairs; i++) {
        const int32_t refs[2] = {d->pairs[i].lref, d->pairs[i].rref};
        for (int32_t r : refs)
            if (r >= 0 ? r >= d->n_pairs : ~r >= d->n_tris) return fail(RT_ERR_INVALID, "rt_scene_create: child ref out of range");
    }
    if (d->n_tris > 0 && !(d->tris[d->n_tris - 1].tri_id & RT_TRI_LEAF_END))
        return fail(RT_ERR_INVALID, "rt_scene_create: last triangle must end a leaf");
    {   // the walks' stacks: 64 entries (flat, the reference's own: Object.h:275), 128 in a two-level scene = 63 for the TLAS
        // (Scene.cpp:19 has 64) + the mesh's exit marker + 64 for the mesh
        std::vector<uint8_t> seen((size_t)d->n_pairs, 0);
        if (!tl) {
            const int depth = pair_tree_depth(d->pairs, d->n_pairs, d->root_ref, seen);
            if (depth < 0) return fail(RT_ERR_INVALID, "rt_scene_create: the pair records do not form a tree (a record is reached twice)");
            if (depth > RT_STACK_TOTAL)
                return fail(RT_ERR_LIMIT, "rt_scene_create: BVH deeper than 64 levels (" + std::to_string(depth) + "; the traversal stack of Object.h:275 holds 64 entries)");
        } else {
            std::vector<int32_t> roots;
            for (int32_t i = 0; i < tl->n_instances && tl->instances; i++)
                if (tl->instances[i].geometry == RT_GEO_MESH && tl->instances[i].blas_root >= 0 && tl->instances[i].blas_root < d->n_pairs)
                    roots.push_back(tl->instances[i].blas_root);
            std::sort(roots.begin(), roots.end());
            roots.erase(std::unique(roots.begin(), roots.end()), roots.end());
            for (int32_t r : roots) {
                const int depth = pair_tree_depth(d->pairs, d->n_pairs, r, seen);
                if (depth < 0) return fail(RT_ERR_INVALID, "rt_scene_create_instanced: the mesh pair records do not form trees (a record is reached twice)");
                if (depth > RT_STACK_TOTAL) return fail(RT_ERR_LIMIT, "rt_scene_create_instanced: a mesh BVH is deeper than 64 levels (Object.h:275)");
            }
        }
    }
    for (int64_t i = 0; i < d->n_tris && !tl; i++) {
        if (d->tris[i].material < 0 || d->tris[i].material >= d->n_materials)
            return fail(RT_ERR_INVALID, "rt_scene_create: triangle material out of range");
        if (d->tris[i].light >= d->n_lights) return fail(RT_ERR_INVALID, "rt_scene_create: triangle light out of range");
    }
    if (tl) {
        if (tl->n_instances <= 0 || !tl->instances) return fail(RT_ERR_INVALID, "rt_scene_create_instanced: no instances");
        if (tl->n_instances >= RT_HIT2_FRONT) return fail(RT_ERR_LIMIT, "rt_scene_create_instanced: more than 2^30 - 1 instances");
        if (tl->n_pairs < 0 || (tl->n_pairs > 0 && !tl->pairs) || tl->n_pairs > RT_MAX_PAIRS)
            return fail(RT_ERR_INVALID, "rt_scene_create_instanced: bad TLAS pair table");
        auto tl_ref_ok = [&](int32_t r) { return r >= 0 ? r < tl->n_pairs : ~r < tl->n_instances; };
        if (!tl_ref_ok(tl->root_ref)) return fail(RT_ERR_INVALID, "rt_scene_create_instanced: TLAS root_ref out of range");
        for (int64_t i = 0; i < tl->n_pairs; i++)
            if (!tl_ref_ok(tl->pairs[i].lref) || !tl_ref_ok(tl->pairs[i].rref))
                return fail(RT_ERR_INVALID, "rt_scene_create_instanced: TLAS child ref out of range");
        {
            std::vector<uint8_t> seen((size_t)tl->n_pairs, 0);
            const int depth = pair_tree_depth(tl->pairs, tl->n_pairs, tl->root_ref, seen);
            if (depth < 0) return fail(RT_ERR_INVALID, "rt_scene_create_instanced: the TLAS pair records do not form a tree");
            if (depth > RT_STACK_TOTAL - 1) return fail(RT_ERR_LIMIT, "rt_scene_create_instanced: TLAS deeper than 63 levels");
        }
        for (int32_t i = 0; i < tl->n_instances; i++) {
            const rt_instance& I = tl->instances[i];
            if (I.geometry < RT_GEO_MESH || I.geometry > RT_GEO_CYLINDER) return fail(RT_ERR_INVALID, "rt_scene_create_instanced: unknown geometry");
            if (I.material < 0 || I.material >= d->n_materials) return fail(RT_ERR_INVALID, "rt_scene_create_instanced: instance material out of range");
            if (I.geometry == RT_GEO_MESH && (I.blas_root >= 0 ? I.blas_root >= d->n_pairs : ~I.blas_root >= d->n_tris))
                return fail(RT_ERR_INVALID, "rt_scene_create_instanced: blas_root out of range");
        }
        for (int32_t i = 0; i < d->n_lights; i++)
            if (d->lights[i].kind != RT_LIGHT_RECT) return fail(RT_ERR_INVALID, "rt_scene_create_instanced: lights must be rectangles");
    }
    for (int32_t i = 0; i < d->n_materials; i++)
        if (d->materials[i].type < 0 || d->materials[i].type >= RT_KEY_MISS) return fail(RT_ERR_INVALID, "rt_scene_create: unknown material type");
    for (int32_t i = 0; i < d->n_lights; i++) {
        const rt_light& L = d->lights[i];
        if (L.material < 0 || L.material >= d->n_materials) return fail(RT_ERR_INVALID, "rt_scene_create: light material out of range");
        if (L.kind == RT_LIGHT_TRIMESH) {
            if (L.count <= 0 || L.first < 0 || L.first + L.count > d->n_emissive_tris)
                return fail(RT_ERR_INVALID, "rt_scene_create: light triangle range out of bounds");
        } else if (L.kind != RT_LIGHT_RECT) {
            return fail(RT_ERR_INVALID, "rt_scene_create: unknown light kind");
        }
    }
    for (int32_t i = 0; i < d->n_emissive_tris; i++)
        if (d->emissive_tris[i].slot < 0 || d->emissive_tris[i].slot >= d->n_tris)
            return fail(RT_ERR_INVALID, "rt_scene_create: emissive triangle slot out of range");
    const bool has_env = d->env_rgb && d->env_w > 0 && d->env_h > 0;
    if (has_env && !d->env_cdf) return fail(RT_ERR_INVALID, "rt_scene_create: environment without CDF");

    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(RT_ERR_NO_DEVICE, "rt_scene_create: no HIP device");
    if (device < 0 || device >= n) return fail(RT_ERR_INVALID, "rt_scene_create: device index out of range");
    HIPCHK(hipSetDevice(device));

    rt_scene* s = new rt_scene;
    s->device = device;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) {
            const unsigned wg_unit = RT_SHARDS * 64 / RT_TRAV_BLOCK; // workgroups per RT_SHARDS waves
            s->persistent_blocks = std::max(wg_unit, ((unsigned)prop.multiProcessorCount * (2048u / RT_TRAV_BLOCK)) / wg_unit * wg_unit);
            unsigned shade_mult = 4; // resident shading workgroups per CU
            if (const char* e = getenv("RT_TUNE_SHADE_MULT")) shade_mult = (unsigned)std::max(1, std::min(8, atoi(e)));
            s->shade_blocks = std::max((unsigned)RT_SHARDS, ((unsigned)prop.multiProcessorCount * shade_mult) / RT_SHARDS * RT_SHARDS);
            unsigned resolve_mult = 8;
            if (const char* e = getenv("RT_TUNE_RESOLVE_MULT")) resolve_mult = (unsigned)std::max(1, std::min(32, atoi(e)));
            s->resolve_blocks = std::max((unsigned)RT_SHARDS, ((unsigned)prop.multiProcessorCount * resolve_mult) / RT_SHARDS * RT_SHARDS);
            if (const char* e = getenv("RT_TUNE_PERSIST_EIGHTHS")) // traversal grid as eighths of full occupancy
                s->persistent_blocks = std::max(wg_unit, s->persistent_blocks * (unsigned)std::max(1, std::min(8, atoi(e))) / 8u / wg_unit * wg_unit);
            // several passes in flight: each pass's traversal launches take 3/8 of the wave slots, so the other pass's
            // kernels (memory-bound shading against VALU-bound traversal) find room beside them
            unsigned pipe_eighths = 3;
            if (const char* e = getenv("RT_TUNE_PIPE_EIGHTHS")) pipe_eighths = (unsigned)std::max(1, std::min(8, atoi(e)));
            s->pipe_blocks = std::max(wg_unit, s->persistent_blocks * pipe_eighths / 8u / wg_unit * wg_unit);
        }
    }
    int rc = RT_OK;
    auto bail = [&](int code) {
        rt_scene_destroy(s);
        return code;
    };
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(RT_ERR_HIP, "hipStreamCreate failed"));
    s->streams[0] = s->stream;
    for (int i = 1; i < RT_MAX_SETS; i++)
        if (hipStreamCreateWithFlags(&s->streams[i], hipStreamNonBlocking) != hipSuccess) return bail(fail(RT_ERR_HIP, "hipStreamCreate failed"));
    if (hipEventCreate(&s->ev_begin) != hipSuccess || hipEventCreate(&s->ev_end) != hipSuccess ||
        hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming) != hipSuccess)
        return bail(fail(RT_ERR_HIP, "hipEventCreate failed"));
    for (int i = 0; i < RT_MAX_SETS; i++)
        if (hipEventCreateWithFlags(&s->ev_acc[i], hipEventDisableTiming) != hipSuccess) return bail(fail(RT_ERR_HIP, "hipEventCreate failed"));
    if (const char* e = getenv("RT_TUNE_SETS")) s->max_sets = std::max(1, std::min(RT_MAX_SETS, atoi(e)));
    if (device < RT_MAX_DEVICES) { // this GPU's address of the allocation-free entry point's scratch array (device is current)
        static std::mutex mu;
        std::lock_guard<std::mutex> lock(mu);
        if (!g_scratch_ptr[device]) {
            void* ptr = nullptr;
            const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
            if (e != hipSuccess || !ptr) return bail(fail(RT_ERR_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e)));
            g_scratch_ptr[device] = ptr;
        }
        s->scratch_ptr = (unsigned char*)g_scratch_ptr[device];
    }
    const rt_bvh_pair* pairs;
    const rt_tri* tris;
    const rt_tri_normals* normals;
    // device layout of the pair records (rt_device.h): bounds regrouped for packed arithmetic, inner refs as byte offsets
    // (two-level: the TLAS records lead the array, every mesh's records follow with their refs moved up)
    const int64_t n_tl = tl ? tl->n_pairs : 0;
    if (n_tl + d->n_pairs > RT_MAX_PAIRS) return bail(fail(RT_ERR_LIMIT, "rt_scene_create: more than 2^25 - 1 child-pair records"));
    std::vector<rt_bvh_pair> dev_pairs((size_t)(n_tl + d->n_pairs));
    auto pack_pairs = [&](const rt_bvh_pair* src, int64_t n, int64_t at, int64_t ref_offset) {
        auto ref = [&](int32_t r) { return r >= 0 ? (int32_t)((r + ref_offset) * RT_PAIR_BYTES) : r; };
        for (int64_t i = 0; i < n; i++) {
            const rt_bvh_pair& p = src[i];
            float* f = (float*)&dev_pairs[(size_t)(at + i)];
            const float q[12] = {p.lmin[0], p.lmin[1], p.lmax[0], p.lmax[1], p.rmin[0], p.rmin[1], p.rmax[0], p.rmax[1],
                                 p.lmin[2], p.rmin[2], p.lmax[2], p.rmax[2]};
            std::memcpy(f, q, sizeof(q));
            const int32_t refs[4] = {ref(p.lref), ref(p.rref), 0, 0};
            std::memcpy(f + 12, refs, sizeof(refs));
        }
    };
    if (tl) pack_pairs(tl->pairs, n_tl, 0, 0);
    pack_pairs(d->pairs, d->n_pairs, n_tl, n_tl);
    if ((rc = upload(s, dev_pairs.data(), dev_pairs.size(), &pairs))) return bail(rc);
    if ((rc = upload(s, d->tris, (size_t)d->n_tris, &tris))) return bail(rc);
    if (d->n_tris > 0) {
        std::vector<rt_tri_normals> dev(d->normals, d->normals + d->n_tris);
        for (int64_t i = 0; i < d->n_tris; i++) { // pad words of the device copy: the triangle's material and light index
            std::memcpy(&dev[(size_t)i].pad0, &d->tris[i].material, 4);
            std::memcpy(&dev[(size_t)i].pad1, &d->tris[i].light, 4);
        }
        if ((rc = upload(s, dev.data(), (size_t)d->n_tris, &normals))) return bail(rc);
    } else {
        normals = nullptr;
    }
    if ((rc = upload(s, d->materials, (size_t)d->n_materials, &s->view.materials))) return bail(rc);
    if ((rc = upload(s, d->lights, (size_t)d->n_lights, &s->view.lights))) return bail(rc);
    if ((rc = upload(s, d->emissive_tris, (size_t)d->n_emissive_tris, &s->view.emissive))) return bail(rc);
    if (has_env) {
        if ((rc = upload(s, d->env_rgb, (size_t)d->env_w * d->env_h * 3, &s->view.env_rgb))) return bail(rc);
        if ((rc = upload(s, d->env_cdf, (size_t)d->env_w * d->env_h, &s->view.env_cdf))) return bail(rc);
        s->view.env_w = d->env_w;
        s->view.env_h = d->env_h;
        s->view.env_total = d->env_total;
        s->view.env_offset = d->env_offset;
    }
    if (tl) {
        // two-level scene: instances repacked as RT_INST_QUADS float4 (rt_device.h) + their parameters, k_classify's
        // sort key per instance
        std::vector<float4> iq((size_t)tl->n_instances * RT_INST_QUADS), ip((size_t)tl->n_instances * 3);
        std::vector<uint8_t> inst_type((size_t)tl->n_instances);
        int64_t per_type[RT_N_KEYS] = {};
        for (int32_t i = 0; i < tl->n_instances; i++) {
            const rt_instance& I = tl->instances[i];
            float4* q = &iq[(size_t)i * RT_INST_QUADS];
            for (int r = 0; r < 3; r++) {
                q[r] = make_float4(I.inv_m[r], I.inv_m[4 + r], I.inv_m[8 + r], I.inv_m[12 + r]);
                q[4 + r] = make_float4(I.m[r], I.m[4 + r], I.m[8 + r], I.m[12 + r]);
            }
            const int32_t root = I.geometry != RT_GEO_MESH ? 0 : (I.blas_root >= 0 ? (int32_t)((I.blas_root + n_tl) * RT_PAIR_BYTES) : I.blas_root);
            q[3] = make_float4(__builtin_bit_cast(float, I.geometry), __builtin_bit_cast(float, I.material), __builtin_bit_cast(float, root), I.area);
            q[7] = make_float4(0.f, 0.f, 0.f, 0.f);
            float p[12] = {};
            std::memcpy(p, I.p, sizeof(I.p));
            for (int r = 0; r < 3; r++) ip[(size_t)i * 3 + r] = make_float4(p[4 * r], p[4 * r + 1], p[4 * r + 2], p[4 * r + 3]);
            inst_type[(size_t)i] = (uint8_t)d->materials[I.material].type;
            per_type[inst_type[(size_t)i]]++;
        }
        if ((rc = upload(s, iq.data(), iq.size(), &s->view.inst_q))) return bail(rc);
        if ((rc = upload(s, ip.data(), ip.size(), &s->view.inst_p))) return bail(rc);
        if ((rc = upload(s, inst_type.data(), inst_type.size(), &s->view.inst_type))) return bail(rc);
        s->view.tl_root = tl->root_ref >= 0 ? tl->root_ref * RT_PAIR_BYTES : tl->root_ref;
        s->view.n_inst = tl->n_instances;
        for (int k = 0; k < RT_KEY_MISS; k++)
            if (per_type[k]) s->keys.push_back(k);
        std::sort(s->keys.begin(), s->keys.end(), [&](int a, int b) { return per_type[a] > per_type[b]; });
        s->keys.push_back(RT_KEY_MISS);
    } else {
        // k_classify's sort key per triangle slot, and the list of shading kernels this scene needs
        std::vector<uint8_t> tri_type((size_t)d->n_tris);
        int64_t per_type[RT_N_KEYS] = {};
        for (int64_t i = 0; i < d->n_tris; i++) {
            tri_type[(size_t)i] = (uint8_t)d->materials[d->tris[i].material].type;
            per_type[tri_type[(size_t)i]]++;
        }
        if ((rc = upload(s, tri_type.data(), tri_type.size(), &s->view.tri_type))) return bail(rc);
        for (int k = 0; k < RT_KEY_MISS; k++)
            if (per_type[k]) s->keys.push_back(k);
        std::sort(s->keys.begin(), s->keys.end(), [&](int a, int b) { return per_type[a] > per_type[b]; });
        s->keys.push_back(RT_KEY_MISS);
    }
    // The occlusion walks' compressed four-wide records of the same tree: built on request only (RT_TUNE_QUADS=1).  Same flags
    // bit for bit (tests/test_gpu_quads.py), 0.46 x the visits and 0.64 x the LDS instructions of the pair walk per ray — and
    // 1.55 x its vector instructions, which is what the walks turn out to be short of: k_shadow 37.6 -> 41.8 ms per step on
    // cornellbox, 27.1 -> 26.1 ms on the 9.6 M-triangle sphere field (profiles/r03_quad_walk.txt).
    if (!tl && getenv("RT_TUNE_QUADS")) {
        std::vector<uint32_t> onodes;
        int32_t qroot = 0, n_quads = 0, leaf_base = 0;
        int need = 0;
        const bool ok = build_onodes(d, onodes, &qroot, &n_quads, &leaf_base, &need);
        if (ok) {
            const uint32_t* dq = nullptr;
            if ((rc = upload(s, onodes.data(), onodes.size(), &dq))) return bail(rc);
            s->view.onodes = (const float4*)dq;
            s->view.qroot = qroot;
            s->view.n_quads = n_quads;
            s->view.leaf_base = leaf_base;
        }
        if (getenv("RT_TUNE_VERBOSE"))
            std::fprintf(stderr, "rt_scene_create: %lld pair records -> %d quad records (%s), worst-case stack %d\n", (long long)d->n_pairs, n_quads,
                         ok ? "used by the occlusion walks" : "NOT used: one leaf, boxes that do not nest, or a tree too deep / too large", need);
    }
    s->view.pairs = (const float4*)pairs;
    s->view.tris = (const float4*)tris;
    s->view.normals = (const float4*)normals;
    s->view.root_ref = d->root_ref >= 0 ? d->root_ref * RT_PAIR_BYTES : d->root_ref;
    s->view.refill_min = 12; // (round 2, after the visits got cheaper in scalar work: 16: k_shadow 38.2 ms per step, 12: 37.6, 8: 37.5 with k_closest +1 %)
    {
        // leaves are tested once this many lanes wait at one: 16 for one-triangle leaves; longer leaf turns (trees built with
        // max_per_node 2: 1.5 triangles per leaf on cornellbox) are worth waiting for more lanes (16: 7.02, 24: 7.09, 28: 7.13 Grays/s)
        int64_t leaves = 0;
        for (int64_t i = 0; i < d->n_tris; i++) leaves += (d->tris[i].tri_id & RT_TRI_LEAF_END) ? 1 : 0;
        s->view.leaf_min = (leaves > 0 && d->n_tris * 10 > leaves * 13) ? 24 : 16;
        // two-level scenes: a leaf turn also holds instance entries, exits and primitive tests (config 5, two triangles per
        // mesh leaf: 16: 2.45, 24: 2.66, 32: 2.75, 40: 2.71 Grays/s)
        if (tl) s->view.leaf_min = 32;
    }
    s->view.wave_min_rays = 0;
    s->view.fetch_chunk = 128;
    if (const char* e = getenv("RT_TUNE_REFILL_MIN")) s->view.refill_min = std::max(1, atoi(e));
    if (const char* e = getenv("RT_TUNE_LEAF_MIN")) s->view.leaf_min = std::max(1, atoi(e));
    if (const char* e = getenv("RT_TUNE_WAVE_MIN")) s->view.wave_min_rays = std::max(0, atoi(e));
    if (const char* e = getenv("RT_TUNE_FETCH_CHUNK")) s->view.fetch_chunk = std::max(64, atoi(e) / 64 * 64);
    s->view.n_lights = d->n_lights;
    s->view.n_emissive = tl ? 0 : d->n_emissive_tris;
    s->view.n_pairs = (int32_t)std::min<int64_t>(n_tl + d->n_pairs, 0x7fffffff);
    s->n_pairs = n_tl + d->n_pairs;
    s->n_tris = d->n_tris;
    s->n_materials = d->n_materials;
    s->n_lights = d->n_lights;
    *out = s;
    return RT_OK;
}

int rt_scene_create(const rt_scene_desc* d, int device, rt_scene** out) { return scene_create(d, nullptr, device, out); }

int rt_scene_create_instanced(const rt_scene_desc* d, const rt_tlas_desc* tlas, int device, rt_scene** out) {
    if (!tlas) return fail(RT_ERR_INVALID, "rt_scene_create_instanced: null TLAS description");
    return scene_create(d, tlas, device, out);
}

void rt_scene_destroy(rt_scene* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    for (int i = 1; i < RT_MAX_SETS; i++)
        if (s->streams[i]) (void)hipStreamSynchronize(s->streams[i]);
    free_wave(s);
    for (void* p : s->owned) (void)hipFree(p);
    if (s->d_sum) (void)hipFree(s->d_sum);
    if (s->d_tiles) (void)hipFree(s->d_tiles);
    if (s->api_mem) (void)hipFree(s->api_mem);
    if (s->h_counts) (void)hipHostFree(s->h_counts);
    for (hipEvent_t e : s->ev_pool) (void)hipEventDestroy(e);
    if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
    for (hipEvent_t e : s->ev_acc)
        if (e) (void)hipEventDestroy(e);
    for (int i = 1; i < RT_MAX_SETS; i++)
        if (s->streams[i]) (void)hipStreamDestroy(s->streams[i]);
    if (s->ev_begin) (void)hipEventDestroy(s->ev_begin);
    if (s->ev_end) (void)hipEventDestroy(s->ev_end);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

int rt_collapse_quads(const rt_bvh_pair* pairs, int64_t n_pairs, int32_t root_ref, const rt_tri* tris, int64_t n_tris, uint32_t* out_words,
                      int64_t capacity_words, int64_t* n_quads, int32_t* qroot, int32_t* stack_need) {
    if (!pairs || !tris || !n_quads || n_pairs <= 0 || n_tris <= 0) return fail(RT_ERR_INVALID, "rt_collapse_quads: bad argument");
    if (root_ref >= n_pairs || (root_ref < 0 && ~root_ref >= n_tris)) return fail(RT_ERR_INVALID, "rt_collapse_quads: root_ref out of range");
    for (int64_t i = 0; i < n_pairs; i++)
        for (int32_t r : {pairs[i].lref, pairs[i].rref})
            if (r >= 0 ? r >= n_pairs : ~r >= n_tris) return fail(RT_ERR_INVALID, "rt_collapse_quads: child ref out of range");
    {
        std::vector<uint8_t> seen((size_t)n_pairs, 0);
        const int depth = pair_tree_depth(pairs, n_pairs, root_ref, seen);
        if (depth < 0 || depth > RT_STACK_TOTAL) return fail(RT_ERR_INVALID, "rt_collapse_quads: the pair records do not form a tree of at most 64 levels");
    }
    rt_scene_desc d{};
    d.root_ref = root_ref;
    d.n_pairs = n_pairs;
    d.pairs = pairs;
    d.n_tris = n_tris;
    d.tris = tris;
    std::vector<uint32_t> onodes;
    int32_t root = 0, nq = 0, leaf_base = 0;
    int need = 0;
    const bool ok = build_onodes(&d, onodes, &root, &nq, &leaf_base, &need);
    if (stack_need) *stack_need = need;
    if (!ok) return fail(RT_ERR_INVALID, "rt_collapse_quads: this tree stays on the pair walk (one leaf, boxes that do not nest, or too deep / too large)");
    *n_quads = nq;
    if (qroot) *qroot = root;
    if (!out_words) return RT_OK;
    if ((int64_t)onodes.size() > capacity_words) return fail(RT_ERR_LIMIT, "rt_collapse_quads: out_words is too small");
    std::memcpy(out_words, onodes.data(), onodes.size() * sizeof(uint32_t));
    return RT_OK;
}

int64_t rt_scene_quad_records(const rt_scene* s) { return s && s->view.onodes ? s->view.n_quads : 0; }

int rt_scene_sync(rt_scene* s) {
    if (!s) return fail(RT_ERR_INVALID, "rt_scene_sync: null scene");
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipStreamSynchronize(s->stream));
    return RT_OK;
}

#define RT_API_KEEP_BYTES ((size_t)1 << 30)
static int trace_common(rt_scene* s, const rt_ray* rays, const float* tmax, int64_t n, rt_hit* hits, uint8_t* occ, int32_t* index = nullptr) {
    if (!s || !rays || n < 0 || (!hits && !occ)) return fail(RT_ERR_INVALID, "rt_trace: bad argument");
    if (n == 0) return RT_OK;
    if (n > 0x7fffffff) return fail(RT_ERR_LIMIT, "rt_trace: more than 2^31-1 rays in one call");
    HIPCHK(hipSetDevice(s->device));
    // The caller's records go to the device as they are (24-byte rays, 4-byte segment ends) and are put into the walk's two float4
    // streams there; hits come back straight into the caller's array (rt_hit is the kernels' 16-byte record).  No host loop over rays.
    static_assert(sizeof(rt_ray) == 24 && sizeof(rt_hit) == 16, "rt_ray / rt_hit are copied as raw records");
    const bool two_level = s->view.inst_q != nullptr;
    int rc = RT_OK;
    // One device allocation holds every buffer of the call; it stays with the scene for the next call (a query of a few rays costs a
    // launch, not seven hipMalloc / hipFree pairs) unless it is larger than RT_API_KEEP_BYTES.
    const size_t sz_count = 2 * RT_SHARDS * RT_CSTRIDE * sizeof(uint32_t), sz4 = ((size_t)n * sizeof(float4) + 255) & ~(size_t)255;
    const size_t sz_rays = ((size_t)n * sizeof(rt_ray) + 255) & ~(size_t)255, sz_f = ((size_t)n * sizeof(float) + 255) & ~(size_t)255;
    const size_t sz_b = ((size_t)n + 255) & ~(size_t)255;
    const size_t need = sz_count + sz_rays + (tmax ? sz_f : 0) + 2 * sz4 + (hits ? sz4 * (two_level ? 2 : 1) : sz_b);
    if (s->api_cap < need) {
        if (s->api_mem) (void)hipFree(s->api_mem);
        s->api_mem = nullptr;
        s->api_cap = 0;
        if (hipMalloc((void**)&s->api_mem, need) != hipSuccess) {
            (void)hipGetLastError();
            return fail(RT_ERR_HIP, "rt_trace: out of device memory for " + std::to_string(n) + " rays");
        }
        s->api_cap = need;
    }
    unsigned char* at = s->api_mem;
    auto take = [&](size_t bytes) {
        unsigned char* p = at;
        at += bytes;
        return p;
    };
    uint32_t* d_count = (uint32_t*)take(sz_count);
    float* d_rays = (float*)take(sz_rays);
    float* d_tmax = tmax ? (float*)take(sz_f) : nullptr;
    float4 *d_o = (float4*)take(sz4), *d_d = (float4*)take(sz4);
    float4* d_hit = hits ? (float4*)take(sz4) : nullptr;
    float4* d_hit2 = hits && two_level ? (float4*)take(sz4) : nullptr;
    uint8_t* d_occ = hits ? nullptr : (uint8_t*)take(sz_b);
    auto cleanup = [&]() {
        if (s->api_cap > RT_API_KEEP_BYTES) {
            (void)hipStreamSynchronize(s->stream);
            (void)hipFree(s->api_mem);
            s->api_mem = nullptr;
            s->api_cap = 0;
        }
    };
#define TRY(expr)                                                                  \
    do {                                                                           \
        hipError_t e_ = (expr);                                                    \
        if (e_ != hipSuccess) {                                                    \
            cleanup();                                                             \
            return fail(RT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
        }                                                                          \
    } while (0)
    TRY(hipMemcpyAsync(d_rays, rays, (size_t)n * sizeof(rt_ray), hipMemcpyHostToDevice, s->stream));
    if (tmax) TRY(hipMemcpyAsync(d_tmax, tmax, (size_t)n * sizeof(float), hipMemcpyHostToDevice, s->stream));
    // the caller's array is dealt out to RT_SHARDS shards in chunks, each shard with its own fetch cursor (rt_device.h, dealt_count);
    // k_stage_rays sets the shard counts and clears the cursors
    const int64_t piece = 0; // shard_cap == 0: dealt
    hipLaunchKernelGGL(k_stage_rays, dim3(blocks_for(std::max<int64_t>(n, RT_SHARDS))), dim3(RT_BLOCK), 0, s->stream, (const float*)d_rays, (const float*)d_tmax, n,
                       (uint32_t)s->view.fetch_chunk, d_o, d_d, d_count);
    TRY(hipGetLastError());
    const unsigned wg_unit = RT_SHARDS * 64 / RT_TRAV_BLOCK; // workgroups per RT_SHARDS waves
    const unsigned grid = std::min(((unsigned)((n + RT_TRAV_BLOCK - 1) / RT_TRAV_BLOCK) + wg_unit - 1) / wg_unit * wg_unit, s->persistent_blocks);
    if (hits) {
        std::vector<float4> h2;
        if (two_level) {
            hipLaunchKernelGGL(k_closest2, dim3(grid), dim3(RT_TRAV_BLOCK), 0, s->stream, s->view, d_o, d_d, d_count, d_count + RT_SHARDS * RT_CSTRIDE,
                               (uint32_t)RT_SHARDS, (uint32_t)piece, d_hit, d_hit2, 1);
            TRY(hipGetLastError());
            if (index) {
                h2.resize((size_t)n);
                TRY(hipMemcpyAsync(h2.data(), d_hit2, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost, s->stream));
            }
        } else {
            hipLaunchKernelGGL(k_closest, dim3(grid), dim3(RT_TRAV_BLOCK), 0, s->stream, s->view, d_o, d_d, d_count, d_count + RT_SHARDS * RT_CSTRIDE,
                               (uint32_t)RT_SHARDS, (uint32_t)piece, d_hit, 1);
            TRY(hipGetLastError());
        }
        TRY(hipMemcpyAsync(hits, d_hit, (size_t)n * sizeof(rt_hit), hipMemcpyDeviceToHost, s->stream));
        TRY(hipStreamSynchronize(s->stream));
        if (index) {
            for (int64_t i = 0; i < n; i++) {
                int32_t iw = hits[i].tri < 0 ? -1 : 0;
                if (two_level) {
                    std::memcpy(&iw, &h2[(size_t)i].w, 4);
                    if (iw >= 0) iw &= RT_HIT2_FRONT - 1;
                }
                index[i] = iw;
            }
        }
    } else {
        if (two_level)
            hipLaunchKernelGGL(k_shadow2, dim3(grid), dim3(RT_TRAV_BLOCK), 0, s->stream, s->view, d_o, d_d, d_count, d_count + RT_SHARDS * RT_CSTRIDE,
                               (uint32_t)RT_SHARDS, (uint32_t)piece, d_occ, 1);
        else if (s->view.onodes)
            hipLaunchKernelGGL(k_any4, dim3(grid), dim3(RT_TRAV_BLOCK), 0, s->stream, s->view, d_o, d_d, d_count, d_count + RT_SHARDS * RT_CSTRIDE,
                               (uint32_t)RT_SHARDS, (uint32_t)piece, d_occ);
        else
            hipLaunchKernelGGL(k_any, dim3(grid), dim3(RT_TRAV_BLOCK), 0, s->stream, s->view, d_o, d_d, d_count, d_count + RT_SHARDS * RT_CSTRIDE,
                               (uint32_t)RT_SHARDS, (uint32_t)piece, d_occ);
        TRY(hipGetLastError());
        TRY(hipMemcpyAsync(occ, d_occ, (size_t)n, hipMemcpyDeviceToHost, s->stream));
        TRY(hipStreamSynchronize(s->stream));
    }
#undef TRY
    cleanup();
    return rc;
}

int rt_trace_closest(rt_scene* s, const rt_ray* rays, int64_t n, rt_hit* hits) { return trace_common(s, rays, nullptr, n, hits, nullptr); }

int rt_trace_closest_indexed(rt_scene* s, const rt_ray* rays, int64_t n, rt_hit* hits, int32_t* index) {
    if (!hits) return fail(RT_ERR_INVALID, "rt_trace_closest_indexed: hits is null");
    return trace_common(s, rays, nullptr, n, hits, nullptr, index);
}

int rt_trace_any(rt_scene* s, const rt_ray* rays, const float* tmax, int64_t n, uint8_t* occluded) {
    if (!tmax) return fail(RT_ERR_INVALID, "rt_trace_any: tmax is null");
    return trace_common(s, rays, tmax, n, nullptr, occluded);
}

// rt_trace_closest for callers that cannot allocate: device pointers in and out, the walk's scratch in this GPU's copy of
// `scratch`.  No hipMalloc / hipFree / allocating library routine on this path (errors are literals, see fail_lit).
int rt_trace_closest_noalloc(rt_scene* s, const rt_ray* rays, int64_t n, rt_hit* hits) {
    if (!s || !rays || !hits || n < 0) return fail_lit(RT_ERR_INVALID, "rt_trace_closest_noalloc: bad argument");
    if (n > RT_NOALLOC_MAX_RAYS) return fail_lit(RT_ERR_LIMIT, "rt_trace_closest_noalloc: more than 524288 rays in one call (the static scratch holds no more)");
    if (((uintptr_t)rays & 3u) || ((uintptr_t)hits & 15u)) return fail_lit(RT_ERR_INVALID, "rt_trace_closest_noalloc: rays must be 4-byte and hits 16-byte aligned");
    if (!s->scratch_ptr) return fail_lit(RT_ERR_HIP, "rt_trace_closest_noalloc: no scratch address for this device");
    if (n == 0) return RT_OK;
    if (hipSetDevice(s->device) != hipSuccess) return fail_lit(RT_ERR_HIP, "rt_trace_closest_noalloc: hipSetDevice failed");
    for (const void* p : {(const void*)rays, (const void*)hits}) { // a host pointer here would be a GPU memory fault in the kernel
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, p) != hipSuccess || (attr.type != hipMemoryTypeDevice && attr.type != hipMemoryTypeManaged) ||
            (attr.type == hipMemoryTypeDevice && attr.device != s->device)) {
            (void)hipGetLastError();
            return fail_lit(RT_ERR_INVALID, "rt_trace_closest_noalloc: rays and hits must be device memory on the scene's GPU");
        }
    }
    uint32_t* d_count = (uint32_t*)s->scratch_ptr;
    float4* d_o = (float4*)(s->scratch_ptr + RT_NOALLOC_COUNT_BYTES);
    float4* d_d = d_o + RT_NOALLOC_MAX_RAYS;
    float4* d_hit2 = d_d + RT_NOALLOC_MAX_RAYS;
    const int64_t piece = 0; // shard_cap == 0: the array is dealt out in chunks (rt_device.h, dealt_count)
    const unsigned wg_unit = RT_SHARDS * 64 / RT_TRAV_BLOCK;
    const unsigned grid = std::min(((unsigned)((n + RT_TRAV_BLOCK - 1) / RT_TRAV_BLOCK) + wg_unit - 1) / wg_unit * wg_unit, s->persistent_blocks);
    hipLaunchKernelGGL(k_stage_rays, dim3(blocks_for(std::max<int64_t>(n, RT_SHARDS))), dim3(RT_BLOCK), 0, s->stream, (const float*)rays, (const float*)nullptr, n, (uint32_t)s->view.fetch_chunk, d_o, d_d, d_count);
    if (s->view.inst_q)
        hipLaunchKernelGGL(k_closest2, dim3(grid), dim3(RT_TRAV_BLOCK), 0, s->stream, s->view, d_o, d_d, d_count, d_count + RT_SHARDS * RT_CSTRIDE,
                           (uint32_t)RT_SHARDS, (uint32_t)piece, (float4*)hits, d_hit2, 1);
    else
        hipLaunchKernelGGL(k_closest, dim3(grid), dim3(RT_TRAV_BLOCK), 0, s->stream, s->view, d_o, d_d, d_count, d_count + RT_SHARDS * RT_CSTRIDE,
                           (uint32_t)RT_SHARDS, (uint32_t)piece, (float4*)hits, 1);
    if (hipGetLastError() != hipSuccess) return fail_lit(RT_ERR_HIP, "rt_trace_closest_noalloc: kernel launch failed");
    if (hipStreamSynchronize(s->stream) != hipSuccess) return fail_lit(RT_ERR_HIP, "rt_trace_closest_noalloc: hipStreamSynchronize failed");
    return RT_OK;
}

// Checks a tile list (rectangles inside the raster, not empty, pairwise disjoint) and puts its device table in place.
static int set_tiles(rt_scene* s, const rt_options* o, const int32_t* tiles, int32_t n_tiles) {
    // the list of the previous call, on the raster it was checked against (the same list on a smaller raster may leave it)
    if (s->tiles_host.size() == (size_t)4 * n_tiles && s->tiles_w == o->width && s->tiles_h == o->height && s->d_tiles &&
        std::memcmp(s->tiles_host.data(), tiles, sizeof(int32_t) * 4 * (size_t)n_tiles) == 0)
        return RT_OK;
    std::vector<int4> tab((size_t)n_tiles);
    int64_t first = 0;
    for (int32_t k = 0; k < n_tiles; k++) {
        const int32_t x0 = tiles[4 * k], y0 = tiles[4 * k + 1], x1 = tiles[4 * k + 2], y1 = tiles[4 * k + 3];
        if (x0 < 0 || y0 < 0 || x1 > o->width || y1 > o->height || x0 >= x1 || y0 >= y1) return fail(RT_ERR_INVALID, "rt_render_tiles: bad tile rectangle");
        tab[(size_t)k] = make_int4(x0, y0, x1 - x0, (int)first);
        first += (int64_t)(x1 - x0) * (y1 - y0);
        if (first > 0x7fffffff) return fail(RT_ERR_LIMIT, "rt_render_tiles: more than 2^31-1 pixels");
    }
    // two lanes adding to one pixel would race: overlapping tiles are refused (sorted by row, compared within a row band)
    std::vector<int32_t> order((size_t)n_tiles);
    for (int32_t k = 0; k < n_tiles; k++) order[(size_t)k] = k;
    std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return tiles[4 * a + 1] < tiles[4 * b + 1]; });
    for (int32_t i = 0; i < n_tiles; i++) {
        const int32_t* a = tiles + 4 * order[(size_t)i];
        for (int32_t j = i + 1; j < n_tiles; j++) {
            const int32_t* b = tiles + 4 * order[(size_t)j];
            if (b[1] >= a[3]) break; // starts below a's last row, and so do all later ones
            if (b[0] < a[2] && a[0] < b[2]) return fail(RT_ERR_INVALID, "rt_render_tiles: tiles overlap");
        }
    }
    if (s->d_tiles_cap < (size_t)n_tiles) {
        if (s->d_tiles) (void)hipFree(s->d_tiles);
        s->d_tiles = nullptr;
        s->d_tiles_cap = 0;
        HIPCHK(hipMalloc((void**)&s->d_tiles, sizeof(int4) * (size_t)n_tiles));
        s->d_tiles_cap = (size_t)n_tiles;
    }
    s->tiles_host.clear();
    HIPCHK(hipMemcpy(s->d_tiles, tab.data(), sizeof(int4) * (size_t)n_tiles, hipMemcpyHostToDevice));
    s->tiles_host.assign(tiles, tiles + 4 * (size_t)n_tiles);
    s->tiles_w = o->width;
    s->tiles_h = o->height;
    s->tiles_npix = first;
    return RT_OK;
}

static int render_common(rt_scene* s, const rt_camera* cam, const rt_options* o, const int32_t* tiles, int32_t n_tiles, float* radiance_sum,
                         rt_stats* stats) {
    if (!s || !cam || !o || !radiance_sum) return fail(RT_ERR_INVALID, "rt_render: null argument");
    if (o->width <= 0 || o->height <= 0) return fail(RT_ERR_INVALID, "rt_render: bad raster size");
    if (n_tiles == 0 && (o->x0 < 0 || o->y0 < 0 || o->x1 > o->width || o->y1 > o->height || o->x0 >= o->x1 || o->y0 >= o->y1))
        return fail(RT_ERR_INVALID, "rt_render: bad tile rectangle");
    if (o->spp_count <= 0 || o->spp_begin < 0 || o->spp_total < o->spp_begin + o->spp_count)
        return fail(RT_ERR_INVALID, "rt_render: bad sample window");
    if (o->max_depth < 1 || o->max_depth > 63) return fail(RT_ERR_LIMIT, "rt_render: max_depth must be in [1, 63]");
    HIPCHK(hipSetDevice(s->device));
    if (n_tiles > 0) {
        const int rc_tiles = set_tiles(s, o, tiles, n_tiles);
        if (rc_tiles) return rc_tiles;
    }

    const int tw = n_tiles ? 0 : o->x1 - o->x0, th = n_tiles ? 0 : o->y1 - o->y0;
    const int64_t npix = n_tiles ? s->tiles_npix : (int64_t)tw * th;
    int spp_pass = o->spp_per_pass;
    // auto: as many passes as can be in flight (max_sets), each up to 16 Mi paths (measured at 1080p, 32 spp per call:
    // 8 spp per pass 6.6 Grays/s, 4 spp 6.5, 2 spp 6.2; two passes of 8 spp in a 16-spp call only 5.7)
    if (spp_pass <= 0)
        spp_pass = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)(16 << 20) / npix, (o->spp_count + s->max_sets - 1) / s->max_sets));
    spp_pass = std::min(spp_pass, o->spp_count);
    const int64_t cap = npix * spp_pass;
    // entries per queue shard: its share of the paths plus one spare RT_BLOCK group (see k_shade_t)
    const int64_t shard_cap = ((cap + (int64_t)RT_SHARDS * RT_BLOCK - 1) / ((int64_t)RT_SHARDS * RT_BLOCK) + 1) * RT_BLOCK;
    const int64_t capq = shard_cap * RT_SHARDS;
    if (cap > (int64_t)1 << 30) return fail(RT_ERR_LIMIT, "rt_render: more than 2^30 paths per pass; lower spp_per_pass");
    const bool has_env = s->view.env_rgb != nullptr;
    const bool two_level = s->view.inst_q != nullptr;
    const int K = (has_env ? 1 : 0) + s->n_lights;
    const int n_pass = (o->spp_count + spp_pass - 1) / spp_pass;
    // Sample passes are independent until their sums are added to the frame: with two or more, pass i runs on
    // stream i % sets with its own buffer set, and only the accumulation kernels are ordered (pass order, by events).
    const int sets = (getenv("RT_TUNE_ONE_STREAM") || (o->flags & RT_RENDER_SERIAL)) ? 1 : std::min(n_pass, s->max_sets);
    int rc;
    if ((rc = ensure_wave(s, capq, K, sets))) return rc;
    if (s->h_counts_n < (size_t)n_pass * RT_COUNTS) {
        if (s->h_counts) (void)hipHostFree(s->h_counts);
        s->h_counts = nullptr;
        s->h_counts_n = 0;
        HIPCHK(hipHostMalloc((void**)&s->h_counts, (size_t)n_pass * RT_COUNTS * sizeof(uint32_t)));
        s->h_counts_n = (size_t)n_pass * RT_COUNTS;
    }

    // accumulation target
    const size_t sum_n = (size_t)o->width * o->height * 3;
    float* d_sum;
    const bool dev_out = (o->flags & RT_RENDER_DEVICE_OUT) != 0;
    if (dev_out) {
        d_sum = radiance_sum;
    } else {
        if ((size_t)s->d_sum_n < sum_n) {
            if (s->d_sum) (void)hipFree(s->d_sum);
            s->d_sum = nullptr;
            s->d_sum_n = 0;
            HIPCHK(hipMalloc((void**)&s->d_sum, sum_n * sizeof(float)));
            s->d_sum_n = (int64_t)sum_n;
        }
        d_sum = s->d_sum;
        if (!(o->flags & RT_RENDER_CLEAR)) HIPCHK(hipMemcpyAsync(d_sum, radiance_sum, sum_n * sizeof(float), hipMemcpyHostToDevice, s->stream));
    }
    if (o->flags & RT_RENDER_CLEAR) HIPCHK(hipMemsetAsync(d_sum, 0, sum_n * sizeof(float), s->stream));

    const bool timing = (o->flags & RT_RENDER_TIMING) != 0;
    // event pairs per pass: generate, accumulate, and closest/shade/shadow/resolve per bounce
    const size_t ev_per_pass = 2 * (2 + 4 * (size_t)o->max_depth);
    if (timing) {
        const size_t need = ev_per_pass * (size_t)n_pass;
        while (s->ev_pool.size() < need) {
            hipEvent_t e;
            HIPCHK(hipEventCreate(&e));
            s->ev_pool.push_back(e);
        }
    }
    size_t ev_next = 0;
    hipStream_t st = s->stream;
    auto mark = [&]() {
        if (timing) (void)hipEventRecord(s->ev_pool[ev_next++], st);
    };

    uint32_t* h_counts = s->h_counts;
    const auto host_t0 = std::chrono::steady_clock::now();
    HIPCHK(hipEventRecord(s->ev_begin, s->stream));
    if (sets > 1) { // the other streams start once the frame buffer is ready
        HIPCHK(hipEventRecord(s->ev_fork, s->stream));
        for (int i = 1; i < sets; i++) HIPCHK(hipStreamWaitEvent(s->streams[i], s->ev_fork, 0));
    }
    for (int pass = 0; pass < n_pass; pass++) {
        const int set = pass % sets;
        st = s->streams[set];
        const Wave& wave = s->wave[set];
        PassParams pp;
        pp.cam = *cam;
        pp.width = o->width;
        pp.x0 = o->x0;
        pp.y0 = o->y0;
        pp.tw = tw;
        pp.npix = (int32_t)npix;
        pp.spp_first = o->spp_begin + pass * spp_pass;
        pp.spp_pass = std::min(spp_pass, o->spp_count - pass * spp_pass);
        pp.spp_total = (uint32_t)o->spp_total;
        pp.seed = o->seed;
        pp.cap = (int32_t)(npix * pp.spp_pass);
        pp.K = K;
        pp.has_env = has_env ? 1 : 0;
        pp.tile_tab = n_tiles ? s->d_tiles : nullptr;
        pp.n_tiles = n_tiles;
        // queue geometry is that of the call's largest pass, so NEE slot strides stay valid for a short last pass
        pp.n_shards = RT_SHARDS;
        pp.shard_cap = (uint32_t)shard_cap;
        pp.capq = (uint32_t)capq;
        const unsigned grid0 = blocks_for(pp.cap);   // bounce-0 queue is dense
        const unsigned gridq = blocks_for(pp.capq);  // later queues are sharded
        const unsigned pgrid = sets > 1 ? s->pipe_blocks : s->persistent_blocks; // persistent traversal launches (multiple of RT_SHARDS waves)
        HIPCHK(hipMemsetAsync(wave.counts, 0, (size_t)3 * RT_COUNTS * RT_CSTRIDE * sizeof(uint32_t), st)); // counts, cursors, type counts
        mark();
        hipLaunchKernelGGL(k_generate, dim3(grid0), dim3(RT_BLOCK), 0, st, wave, pp);
        mark();
        for (int b = 0; b < o->max_depth; b++) {
            const int in = b & 1;
            uint32_t* ray_counts = wave.counts + (size_t)b * RT_SHARDS * RT_CSTRIDE;
            uint32_t* shadow_counts = wave.counts + ((size_t)64 * RT_SHARDS + (size_t)b * RT_SHARDS) * RT_CSTRIDE;
            mark();
            if (two_level)
                hipLaunchKernelGGL(k_closest2, dim3(pgrid), dim3(RT_TRAV_BLOCK), 0, st, s->view, wave.q_o[in], wave.q_d[in], ray_counts,
                                   ray_counts + (size_t)RT_COUNTS * RT_CSTRIDE, (uint32_t)RT_SHARDS, pp.shard_cap, wave.hit, wave.hit2, 0);
            else
                hipLaunchKernelGGL(k_closest, dim3(pgrid), dim3(RT_TRAV_BLOCK), 0, st, s->view, wave.q_o[in], wave.q_d[in],
                                   ray_counts, ray_counts + (size_t)RT_COUNTS * RT_CSTRIDE, (uint32_t)RT_SHARDS, pp.shard_cap, wave.hit, 0);
            mark();
            mark();
            hipLaunchKernelGGL(k_classify, dim3(s->shade_blocks), dim3(RT_BLOCK), 0, st, s->view, wave, pp, b);
            for (int key : s->keys) launch_shade(s, st, wave, key, pp, b, b == o->max_depth - 1 ? 1 : 0);
            mark();
            mark();
            if (K > 0 && two_level)
                hipLaunchKernelGGL(k_shadow2, dim3(pgrid), dim3(RT_TRAV_BLOCK), 0, st, s->view, wave.s_o, wave.s_d, shadow_counts,
                                   shadow_counts + (size_t)RT_COUNTS * RT_CSTRIDE, (uint32_t)RT_SHARDS, (uint32_t)K * pp.shard_cap, wave.vis, 0);
            else if (K > 0 && s->view.onodes)
                hipLaunchKernelGGL(k_shadow4, dim3(pgrid), dim3(RT_TRAV_BLOCK), 0, st, s->view, wave.s_o, wave.s_d, shadow_counts,
                                   shadow_counts + (size_t)RT_COUNTS * RT_CSTRIDE, (uint32_t)RT_SHARDS, (uint32_t)K * pp.shard_cap, wave.vis);
            else if (K > 0)
                hipLaunchKernelGGL(k_shadow, dim3(pgrid), dim3(RT_TRAV_BLOCK), 0, st, s->view, wave.s_o, wave.s_d, shadow_counts,
                                   shadow_counts + (size_t)RT_COUNTS * RT_CSTRIDE, (uint32_t)RT_SHARDS, (uint32_t)K * pp.shard_cap, wave.vis);
            mark();
            mark();
            if (K > 0) hipLaunchKernelGGL(k_nee_resolve, dim3(s->resolve_blocks), dim3(RT_BLOCK), 0, st, wave, pp, b);
            mark();
        }
        mark();
        if (sets > 1 && pass > 0) HIPCHK(hipStreamWaitEvent(st, s->ev_acc[(pass - 1) % sets], 0)); // frame adds in pass order
        hipLaunchKernelGGL(k_accumulate, dim3(blocks_for(npix)), dim3(RT_BLOCK), 0, st, wave, pp, d_sum);
        mark();
        if (sets > 1) HIPCHK(hipEventRecord(s->ev_acc[set], st));
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(&h_counts[(size_t)pass * RT_COUNTS], wave.counts_dense, RT_COUNTS * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    }
    if (sets > 1) HIPCHK(hipStreamWaitEvent(s->stream, s->ev_acc[(n_pass - 1) % sets], 0)); // join: it waited for all earlier ones
    HIPCHK(hipEventRecord(s->ev_end, s->stream));
    if (!dev_out) HIPCHK(hipMemcpyAsync(radiance_sum, d_sum, sum_n * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    const auto host_t1 = std::chrono::steady_clock::now();
    HIPCHK(hipStreamSynchronize(s->stream));
    if (getenv("RT_TUNE_VERBOSE"))
        std::fprintf(stderr, "rt_render: %d passes on %d streams enqueued in %.3f ms, finished %.3f ms after the first launch\n", n_pass, sets,
                     std::chrono::duration<double, std::milli>(host_t1 - host_t0).count(),
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t0).count());
    for (int i = 1; i < sets; i++) HIPCHK(hipStreamSynchronize(s->streams[i])); // their counter read-backs

    if (stats) {
        rt_stats st{};
        for (int pass = 0; pass < n_pass; pass++) {
            const uint32_t* c = &h_counts[(size_t)pass * RT_COUNTS];
            for (int sh = 0; sh < RT_SHARDS; sh++) st.paths += c[sh];
            for (int b = 0; b < o->max_depth; b++) {
                uint64_t nc = 0, ns = 0;
                for (int sh = 0; sh < RT_SHARDS; sh++) {
                    nc += c[b * RT_SHARDS + sh];
                    ns += c[64 * RT_SHARDS + b * RT_SHARDS + sh];
                }
                st.rays_closest += nc;
                st.rays_shadow += ns;
                if (nc) st.launches_closest++;
                if (ns) st.launches_shadow++;
            }
        }
        HIPCHK(hipEventElapsedTime(&st.ms_total, s->ev_begin, s->ev_end));
        if (timing) {
            size_t e = 0;
            auto span = [&]() {
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, s->ev_pool[e], s->ev_pool[e + 1]);
                e += 2;
                return ms;
            };
            for (int pass = 0; pass < n_pass; pass++) {
                st.ms_generate += span();
                for (int b = 0; b < o->max_depth; b++) {
                    st.ms_closest += span();
                    st.ms_shade += span();
                    st.ms_shadow += span();
                    st.ms_shade += span(); // NEE resolve is counted with shading
                }
                st.ms_accumulate += span();
            }
        }
        stats->rays_closest += st.rays_closest;
        stats->rays_shadow += st.rays_shadow;
        stats->paths += st.paths;
        stats->launches_closest += st.launches_closest;
        stats->launches_shadow += st.launches_shadow;
        stats->ms_total += st.ms_total;
        stats->ms_generate += st.ms_generate;
        stats->ms_closest += st.ms_closest;
        stats->ms_shade += st.ms_shade;
        stats->ms_shadow += st.ms_shadow;
        stats->ms_accumulate += st.ms_accumulate;
    }
    return RT_OK;
}

int rt_render(rt_scene* s, const rt_camera* cam, const rt_options* o, float* radiance_sum, rt_stats* stats) {
    return render_common(s, cam, o, nullptr, 0, radiance_sum, stats);
}

int rt_render_tiles(rt_scene* s, const rt_camera* cam, const rt_options* o, const int32_t* tiles_xyxy, int32_t n_tiles, float* radiance_sum,
                    rt_stats* stats) {
    if (!tiles_xyxy || n_tiles <= 0) return fail(RT_ERR_INVALID, "rt_render_tiles: empty tile list");
    if (n_tiles > (1 << 20)) return fail(RT_ERR_LIMIT, "rt_render_tiles: more than 2^20 tiles");
    return render_common(s, cam, o, tiles_xyxy, n_tiles, radiance_sum, stats);
}

#ifdef RT_STATS
// debug builds only: reads and clears the lane-occupancy counters ([0..7] closest walk, [8..15] occlusion walk;
// pairs of {wave-level steps, lanes active in them}: inner steps, leaf turns, -, rounds)
int rt_debug_lane_stats(unsigned long long* out16) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_rt_stats), 16 * sizeof(unsigned long long)) != hipSuccess) return RT_ERR_HIP;
    unsigned long long zero[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_rt_stats), zero, sizeof(zero)) != hipSuccess) return RT_ERR_HIP;
    return RT_OK;
}
// two-level walks: lanes per kind of leaf-turn work ([0..7] closest, [8..15] occlusion; rt_device.h, g_rt_stats2)
int rt_debug_lane_stats2(unsigned long long* out16) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_rt_stats2), 16 * sizeof(unsigned long long)) != hipSuccess) return RT_ERR_HIP;
    unsigned long long zero[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_rt_stats2), zero, sizeof(zero)) != hipSuccess) return RT_ERR_HIP;
    return RT_OK;
}
// where the walks' wave cycles went ([0..7] closest, [8..15] occlusion; sections: rt_device.h, g_rt_time)
int rt_debug_walk_times(unsigned long long* out16) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_rt_time), 16 * sizeof(unsigned long long)) != hipSuccess) return RT_ERR_HIP;
    unsigned long long zero[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_rt_time), zero, sizeof(zero)) != hipSuccess) return RT_ERR_HIP;
    return RT_OK;
}
// k_shade sections: out32[2*i] = summed wave cycles of section i, out32[2*i+1] = summed lanes that reached its end
int rt_debug_shade_stats(unsigned long long* out32) {
    static unsigned long long host[256 * 32];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_shade_stats), sizeof(host)) != hipSuccess) return RT_ERR_HIP;
    for (int i = 0; i < 32; i++) {
        out32[i] = 0;
        for (int r = 0; r < 256; r++) out32[i] += host[r * 32 + i];
    }
    std::memset(host, 0, sizeof(host));
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_shade_stats), host, sizeof(host)) != hipSuccess) return RT_ERR_HIP;
    return RT_OK;
}
#endif

int rt_resolve(rt_scene* s, const float* radiance_sum, int32_t width, int32_t height, int32_t spp, float gamma, uint8_t* rgb_out) {
    if (!s || !radiance_sum || !rgb_out || width <= 0 || height <= 0 || spp <= 0) return fail(RT_ERR_INVALID, "rt_resolve: bad argument");
    HIPCHK(hipSetDevice(s->device));
    const size_t n = (size_t)width * height * 3;
    float* d_in = nullptr;
    uint8_t* d_out = nullptr;
    HIPCHK(hipMalloc((void**)&d_in, n * sizeof(float)));
    if (hipMalloc((void**)&d_out, n) != hipSuccess) {
        (void)hipFree(d_in);
        return fail(RT_ERR_HIP, "rt_resolve: hipMalloc failed");
    }
    hipError_t e = hipMemcpyAsync(d_in, radiance_sum, n * sizeof(float), hipMemcpyHostToDevice, s->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_resolve, dim3(blocks_for((int64_t)width * height)), dim3(RT_BLOCK), 0, s->stream, d_in, width, height,
                           (float)(1.0 / spp), 1.f / gamma, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(rgb_out, d_out, n, hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(RT_ERR_HIP, std::string("rt_resolve: ") + hipGetErrorString(e));
    return RT_OK;
}

} // extern "C"
