// rt_device.h — device-side math, random stream, BVH traversal and material
// functions for the gfx950 wavefront path tracer.  Included only by rt_hip.hip.
//
// Arithmetic notes (bit-exactness of hit indices, SURVEY §7 "hard parts"):
//   * the file is compiled with -ffp-contract=off: no FMA is formed from a*b+c;
//   * fmin_sel/fmax_sel are the select forms of the reference's Math.h:17-24,
//     not v_min_f32/v_max_f32 (those differ when the SECOND operand is NaN);
//   * 1.f/dir is an IEEE division (hipcc's default correctly-rounded fp32
//     divide), vec3/float is a reciprocal multiply (Math.h:193-200).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_hip.h"

#define RT_BLOCK 256        // threads per workgroup = 4 wave64 (shading / streaming kernels)
#ifndef RT_TRAV_BLOCK
#define RT_TRAV_BLOCK 1024
#endif
#ifndef RT_TOP_PAIRS
#define RT_TOP_PAIRS 256
#endif
// RT_TRAV_BLOCK: threads per workgroup of the traversal kernels.  RT_TOP_PAIRS: child-pair records [0, RT_TOP_PAIRS) are
// read from LDS (16 KB per workgroup); the host numbers the top levels of the tree first.  On cornellbox half of all
// node visits fall into the first 64 records.  Two 1024-thread workgroups fill a CU (32 waves) and each stages ONE copy
// of the top for its 16 waves: measured 256 threads + 64 records 5.00, 512 + 128 5.07, 1024 + 256 5.15 Grays/s
// (512 or 768 records: no further gain).
#define RT_OCC_TOP_PAIRS RT_TOP_PAIRS // tree top of the occlusion kernels (one layout for all walks: load_pair)
#ifndef RT_SHORT_STACK
#define RT_SHORT_STACK 8
#endif
// RT_SHORT_STACK: traversal stack entries per lane kept in LDS (32 KB per 1024-thread workgroup)
#define RT_STACK_TOTAL 64   // reference stack capacity (Object.h:275)

// Queue counters and fetch cursors take atomics from every CU: each sits in its own 128-byte line (counter i at word
// i * RT_CSTRIDE), because atomics on one line are serviced one after the other wherever in the line they land.
#define RT_CSTRIDE 32

#define RT_DEV __device__ __forceinline__

namespace rtd {

constexpr float kPi = 3.1415926536f; // Math.h:9 narrowed to float

RT_DEV float fmax_sel(float a, float b) { return (a > b) ? a : b; }
RT_DEV float fmin_sel(float a, float b) { return (a < b) ? a : b; }
RT_DEV float clampf(float x, float lo, float hi) { return (x < lo) ? lo : ((x > hi) ? hi : x); }

struct V3 {
    float x, y, z;
    RT_DEV V3() {}
    RT_DEV V3(float a, float b, float c) : x(a), y(b), z(c) {}
    RT_DEV explicit V3(float e) : x(e), y(e), z(e) {}
    RT_DEV explicit V3(const float* p) : x(p[0]), y(p[1]), z(p[2]) {}
    RT_DEV explicit V3(const float4& q) : x(q.x), y(q.y), z(q.z) {}
};
RT_DEV V3 operator-(const V3& a) { return V3(-a.x, -a.y, -a.z); }
RT_DEV V3 operator-(const V3& a, const V3& b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV V3 operator+(const V3& a, const V3& b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV V3 operator*(const V3& a, const V3& b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_DEV V3 operator*(float t, const V3& v) { return V3(t * v.x, t * v.y, t * v.z); }
RT_DEV V3 operator*(const V3& v, float t) { return V3(v.x * t, v.y * t, v.z * t); }
RT_DEV V3 operator/(const V3& v, float t) { return (1.f / t) * v; }
RT_DEV float dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_DEV V3 cross(const V3& u, const V3& v) { return V3(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x); }
RT_DEV float len2(const V3& v) { return v.x * v.x + v.y * v.y + v.z * v.z; }
RT_DEV float len(const V3& v) { return sqrtf(len2(v)); }
RT_DEV V3 normalize(const V3& v) { return v / len(v); }
RT_DEV float luminance(const V3& c) { return 0.212671f * c.x + 0.715160f * c.y + 0.072169f * c.z; }
RT_DEV float4 pack(const V3& v, float w) { return make_float4(v.x, v.y, v.z, w); }

// ------------------------------------------------------------------ random
// Counter-based stream keyed by (pixel, sample, seed); specified in DESIGN.md
// and restated independently by the oracle.
RT_DEV uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
RT_DEV uint32_t rng_key(uint32_t pixel, uint32_t sample, uint32_t seed) {
    uint32_t k = mix32(pixel * 0x9E3779B1u + seed);
    return mix32(k ^ (sample * 0x85EBCA77u + 0x165667B1u));
}
struct Rng {
    uint32_t key, ctr;
    RT_DEV float next() { return (float)(mix32(key + (ctr++) * 0xC2B2AE3Du) >> 8) * 5.9604644775390625e-08f; }
};

// --------------------------------------------------------------- traversal
struct SceneView {
    const float4* pairs;   // 4 x float4 per rt_bvh_pair
    const float4* tris;    // 3 x float4 per rt_tri
    const float4* normals; // 3 x float4 per rt_tri_normals
    const rt_material* materials;
    const uint8_t* tri_type; // material type of each triangle slot (k_classify's sort key)
    const rt_light* lights;
    const rt_emissive_tri* emissive;
    const float* env_rgb;
    const float* env_cdf;
    int32_t root_ref;
    int32_t n_pairs;
    int32_t n_lights;
    int32_t n_emissive;
    int32_t env_w, env_h;
    float env_total, env_offset;
    int32_t refill_min; // refill a wave's idle lanes once this many are idle
    int32_t leaf_min;   // test pending leaves once this many lanes wait at one (or no lane has an inner node)
    int32_t wave_min_rays; // a shard of n rays is worked on by at most ceil(n / wave_min_rays) waves
    int32_t fetch_chunk;   // rays a wave takes from its shard per cursor atomic (multiple of 64)
    // legacy two-level scenes (inst_q != nullptr): instances, k_classify's sort key per instance
    // compressed four-wide records of the SAME tree for the occlusion walks (see "four-wide occlusion walk" below); nullptr:
    // the scene walks pair records everywhere (the default: see rt_scene_create; also boxes that do not nest, trees too deep for the quad stack)
    const float4* onodes;  // 4 x float4 per record: quad records, then one leaf record per triangle slot
    int32_t qroot;         // ref of the root quad record
    int32_t n_quads;
    int32_t leaf_base;     // byte offset of slot 0's leaf record
    const float4* inst_q;  // RT_INST_QUADS float4 per instance (see "two-level scenes" below); the TLAS records lead `pairs`
    const float4* inst_p;  // 3 float4 per instance: rt_instance.p
    const uint8_t* inst_type;
    int32_t tl_root;
    int32_t n_inst;
};

// ---- device layout of a child-pair record (64 B, repacked from rt_bvh_pair by rt_scene_create) -------------------
//   q0 = (lmin.x, lmin.y, lmax.x, lmax.y)   q1 = (rmin.x, rmin.y, rmax.x, rmax.y)
//   q2 = (lmin.z, rmin.z, lmax.z, rmax.z)   q3 = (lref, rref, -, -)
// so that the 24 subtract/multiply operations of the two slab tests are 6 v_pk_add_f32 + 6 v_pk_mul_f32 on aligned
// register pairs (each IEEE per component: bit-identical to the scalar form).  A child ref >= 0 is the BYTE offset of
// the child's own record (pair index * 64): it is the 32-bit offset operand of the load, no address arithmetic is left
// on the vector ALU.  A child ref < 0 is a leaf, ~ref = its first triangle slot (as in rt_bvh_pair).
#define RT_PAIR_BYTES 64
#define RT_MAX_PAIRS ((int64_t)0x7fffffff / RT_PAIR_BYTES)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) f4 lds_cf4;
typedef __attribute__((address_space(3))) int lds_int;

RT_DEV float vmin(float a, float b) { // written as instructions: through fminf/fmaxf hipcc adds a canonicalising v_max_f32 x, x, x per operand
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
RT_DEV float vmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
RT_DEV float vmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
RT_DEV float vmin3(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// A ray as the slab tests want it: origin and 1/dir (Math.h:756, true division) as register pairs.
struct RaySlab {
    f2 oxy, ozz, ixy, izz;
};

// Bounds::intersect (Math.h:755-770) on both children of a pair record.  EXACT = false: v_min_f32 / v_max3_f32 instead
// of compare+select chains.  IEEE min/max differ from the reference's select forms (Math.h:17-24) only when an operand
// is NaN (and in the sign of a zero result, which no comparison below can observe); a NaN needs 0 * inf, i.e. an
// infinite 1/dir, or a non-finite origin.  The walks take this path only while every ray in the wave has finite origin
// and finite 1/dir, so results are bit-identical.  EXACT = true: the select forms, in the reference's operand order.
template <bool EXACT>
RT_DEV void slab_pair(const f4& q0, const f4& q1, const f4& q2, const RaySlab& r, float& ln, bool& lh, float& rn, bool& rh) {
    const f2 a = (f2{q0.x, q0.y} - r.oxy) * r.ixy; // left  t(min) x, y
    const f2 b = (f2{q0.z, q0.w} - r.oxy) * r.ixy; // left  t(max) x, y
    const f2 c = (f2{q1.x, q1.y} - r.oxy) * r.ixy; // right t(min) x, y
    const f2 d = (f2{q1.z, q1.w} - r.oxy) * r.ixy; // right t(max) x, y
    const f2 e = (f2{q2.x, q2.y} - r.ozz) * r.izz; // t(min) z of left, right
    const f2 f = (f2{q2.z, q2.w} - r.ozz) * r.izz; // t(max) z of left, right
    float lf, rf;
    if (!EXACT) {
        ln = vmax3(vmin(a.x, b.x), vmin(a.y, b.y), vmin(e.x, f.x));
        lf = vmin3(vmax(a.x, b.x), vmax(a.y, b.y), vmax(e.x, f.x));
        rn = vmax3(vmin(c.x, d.x), vmin(c.y, d.y), vmin(e.y, f.y));
        rf = vmin3(vmax(c.x, d.x), vmax(c.y, d.y), vmax(e.y, f.y));
    } else {
        ln = fmax_sel(fmax_sel(fmin_sel(a.x, b.x), fmin_sel(a.y, b.y)), fmin_sel(e.x, f.x));
        lf = fmin_sel(fmin_sel(fmax_sel(a.x, b.x), fmax_sel(a.y, b.y)), fmax_sel(e.x, f.x));
        rn = fmax_sel(fmax_sel(fmin_sel(c.x, d.x), fmin_sel(c.y, d.y)), fmin_sel(e.y, f.y));
        rf = fmin_sel(fmin_sel(fmax_sel(c.x, d.x), fmax_sel(c.y, d.y)), fmax_sel(e.y, f.y));
    }
    lh = !(ln > lf || lf < 0.f);
    rh = !(rn > rf || rf < 0.f);
}

// One record: the top of the tree (every ray walks it) comes from LDS, which has its own bandwidth beside the vector
// L1 that the divergent 64-byte record reads saturate; the rest through the 32-bit-offset form of the global load.
// The LDS copy is kept as FOUR ARRAYS (all q0, all q1, all q2, all ref pairs), not as records: 64 lanes reading the same
// 16-byte part of 64 unrelated 64-byte records touch only banks 16 r .. 16 r + 3 (a quarter of the banks); part arrays
// spread the same reads over all of them.  Array strides are compile-time, so the three reads differ by an immediate.
#define RT_TOP_Q_BYTES (RT_TOP_PAIRS * 16)
#define RT_TOP_LDS_BYTES (RT_TOP_PAIRS * 56) // three float4 arrays + one array of (lref, rref)
RT_DEV void load_pair(const float4* __restrict__ pairs, lds_cf4* top, int top_bytes, int cur, f4& q0, f4& q1, f4& q2, int& lref, int& rref) {
    if (cur < top_bytes) {
        const __attribute__((address_space(3))) char* base = (const __attribute__((address_space(3))) char*)top;
        const uint32_t at = (uint32_t)cur >> 2; // record index * 16
        q0 = *(lds_cf4*)(base + at);
        q1 = *(lds_cf4*)(base + at + RT_TOP_Q_BYTES);
        q2 = *(lds_cf4*)(base + at + 2 * RT_TOP_Q_BYTES);
        const unsigned long long rr = *(const __attribute__((address_space(3))) unsigned long long*)(base + ((uint32_t)cur >> 3) + 3 * RT_TOP_Q_BYTES);
        lref = (int)(uint32_t)rr;
        rref = (int)(uint32_t)(rr >> 32);
#ifdef RT_PROBE_LDS
        { const int x_ = *(const __attribute__((address_space(3))) int*)(base + (at ^ 16u)); asm volatile("" ::"v"(x_)); }
#endif
    } else {
        const f4* p = (const f4*)((const char*)pairs + (uint32_t)cur);
        q0 = p[0];
        q1 = p[1];
        q2 = p[2];
        const unsigned long long rr = *(const unsigned long long*)(p + 3); // 56 of the record's 64 bytes are read
        lref = (int)(uint32_t)rr;
        rref = (int)(uint32_t)(rr >> 32);
#ifdef RT_PROBE_VMEM
        { const int x_ = *((const int*)p - 2); /* the record before: records beyond the top have cur > 0 */ asm volatile("" ::"v"(x_)); }
#endif
    }
}

// Timing probes (never in the shipped library; tools/ab.sh): N extra vector-ALU or scalar-ALU instructions per inner step (dependent
// chains), a sleep of N x 64 cycles per inner step (latency without instruction issue), one extra LDS read per record taken from
// the tree top, one extra load per record taken from global memory — to see what a walk is sensitive to.
#if defined(RT_PROBE_VALU)
#define RT_PROBE_STEP(x) do { float x_ = (x); _Pragma("unroll") for (int k_ = 0; k_ < RT_PROBE_VALU; k_++) asm volatile("v_add_f32 %0, %0, %0" : "+v"(x_)); asm volatile("" ::"v"(x_)); } while (0)
#elif defined(RT_PROBE_SALU)
#define RT_PROBE_STEP(x) do { int s_ = 0; _Pragma("unroll") for (int k_ = 0; k_ < RT_PROBE_SALU; k_++) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s_) : : "scc"); asm volatile("" ::"s"(s_)); } while (0)
#elif defined(RT_PROBE_SLEEP)
#define RT_PROBE_STEP(x) __builtin_amdgcn_s_sleep(RT_PROBE_SLEEP)
#else
#define RT_PROBE_STEP(x)
#endif
#ifndef RT_CLOSEST_STEPS
#define RT_CLOSEST_STEPS 2 // inner steps per round of the closest walk
#endif

#define RT_CUR_DONE ((int)0x80000000)   // lane has no ray
#define RT_CUR_FINISH ((int)0x80000001) // lane's ray is complete, result not yet written

// What a visit does to the lane's stack once both children are tested (Object.h:328-343): push the farther child when
// both are hit, take the next entry (or end the walk) when none is.  Entries 0 .. RT_SHORT_STACK-1 live in the lane's LDS
// column, deeper ones in private memory; the deep case is taken out of the way by ONE wave-uniform test, so the common
// path is two short predicated regions instead of a nest of five (fewer scalar instructions per visit: k_shadow 38.9 ->
// 38.1 ms, k_closest 33.2 -> 32.8 ms per step, 7.10 -> 7.24 Grays/s in flight).
template <int TOTAL>
RT_DEV void stack_step(lds_int* lds, int* spill, bool both, bool none, int far, int& cur, int& sp) {
    if (__ballot(sp >= RT_SHORT_STACK) == 0ull) {
        if (both) lds[sp * RT_TRAV_BLOCK] = far;
        const bool take = none && sp > 0;
        int next = RT_CUR_FINISH;
        if (take) next = lds[(sp - 1) * RT_TRAV_BLOCK];
        cur = none ? next : cur;
        sp += (both ? 1 : 0) - (take ? 1 : 0);
    } else if (both) {
        if (sp < RT_SHORT_STACK) lds[sp * RT_TRAV_BLOCK] = far;
        else if (sp < TOTAL) spill[sp - RT_SHORT_STACK] = far;
        if (sp < TOTAL) sp++; // (rt_scene_create refuses deeper trees; were one to get here, the entry is lost, never memory)
    } else if (none) {
        if (sp == 0) {
            cur = RT_CUR_FINISH;
        } else {
            sp--;
            if (sp < RT_SHORT_STACK) cur = lds[sp * RT_TRAV_BLOCK];
            else cur = spill[sp - RT_SHORT_STACK];
        }
    }
}

struct RayState {
    V3 o, d;
    float tmax;  // shrinking upper bound on accepted t (Object.h:314)
    float t;     // best hit so far, INF before the first (Object.h:281)
    float u, v;
    int slot;    // best hit's triangle slot, -1 = none
    uint32_t q;  // queue position of this ray
};

// IntersectTriRay2 (Object.h:245-264, edges pre-differenced by the host) followed by the acceptance test of
// TriangleMesh::intersect (Object.h:260, 309-315 with tmin = 0).  ONE definition for every caller (closest walk,
// occlusion walk), so that both accept exactly the same (ray, triangle, tmax) triples.
// `a > -1e-7 && a < 1e-7` is written in double in the reference; float(1e-7) is the smallest float above the double
// 1e-7, so the fp32 compare accepts the same set.
RT_DEV bool tri_accept(const V3& o, const V3& d, const float4& a, const float4& b, const float4& c, float tmax, float* t_out,
                       float* u_out, float* v_out) {
    const V3 e1(b.x, b.y, b.z), e2(c.x, c.y, c.z), v0(a.x, a.y, a.z);
    const V3 h = cross(d, e2);
    const float det = dot(e1, h);
    if (det > -1e-7f && det < 1e-7f) return false;
    const float f = 1.f / det;
    const V3 s = o - v0;
    const float u = f * dot(s, h);
    if (u < 0.f || u > 1.f) return false;
    const V3 qq = cross(s, e1);
    const float v = f * dot(d, qq);
    if (v < 0.f || u + v > 1.f) return false;
    const float tt = f * dot(e2, qq);
    if (!(tt > 0.0001f && !(tt > tmax) && !(tt < 0.f))) return false;
    *t_out = tt;
    *u_out = u;
    *v_out = v;
    return true;
}
// The three 16-byte parts of triangle slot `slot`, all requested before the first is used (one memory round trip per
// leaf, not one for the edges and a second for v0 behind the determinant test).
RT_DEV void load_tri(const float4* __restrict__ tris, int slot, float4& a, float4& b, float4& c) {
    const float4* tp = (const float4*)((const char*)tris + (uint32_t)slot * 48u);
    a = tp[0];
    b = tp[1];
    c = tp[2];
    asm volatile("" ::"v"(a.x), "v"(b.x), "v"(c.x));
}

#ifdef RT_STATS // lane-occupancy instrumentation of the traversal loops (debug builds only: tools/lane_stats.sh)
__device__ unsigned long long g_rt_stats[16];
#define RT_STAT_DECL unsigned st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define RT_STAT(i, pred) do { const unsigned long long m_ = __ballot(pred); if (m_) { st_[2 * (i)] += 1; st_[2 * (i) + 1] += (unsigned)__popcll(m_); } } while (0)
#define RT_STAT_FLUSH(base) do { if (__lane_id() == 0) for (int k_ = 0; k_ < 8; k_++) atomicAdd(&g_rt_stats[(base) + k_], (unsigned long long)st_[k_]); } while (0)
// ... and where a wave's cycles go (s_memtime deltas summed per wave; [0..7] closest walk, [8..15] occlusion): [0] refill,
// [1] from a visit's start until its record has arrived (a wait is forced there), [2] the rest of the visit incl. its pop,
// [3] leaf turns, [4] the round's own tests, [5] the cost of one mark (two marks back to back, once per round): every other
// section contains one.
__device__ unsigned long long g_rt_time[16];
#define RT_TDECL unsigned long long tm_[6] = {0, 0, 0, 0, 0, 0}; unsigned long long tprev_ = __builtin_readcyclecounter();
#define RT_TMARK(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); tm_[i] += now_ - tprev_; tprev_ = now_; } while (0)
#define RT_TWAIT asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#define RT_TFLUSH(base) do { if (__lane_id() == 0) for (int k_ = 0; k_ < 6; k_++) atomicAdd(&g_rt_time[(base) + k_], tm_[k_]); } while (0)
// ... and, for the two-level walk, lanes per kind of leaf-turn work: [0] mesh exits, [1] mesh leaves, [2] TLAS leaves (instance
// entries and primitive tests), [3] finished rays; pairs of {wave-level turns that had some, lanes}
__device__ unsigned long long g_rt_stats2[16];
#define RT_STAT2_DECL unsigned st2_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define RT_STAT2(i, pred) do { const unsigned long long m_ = __ballot(pred); if (m_) { st2_[2 * (i)] += 1; st2_[2 * (i) + 1] += (unsigned)__popcll(m_); } } while (0)
#define RT_STAT2_FLUSH(base) do { if (__lane_id() == 0) for (int k_ = 0; k_ < 8; k_++) atomicAdd(&g_rt_stats2[(base) + k_], (unsigned long long)st2_[k_]); } while (0)
#else
#define RT_STAT2_DECL
#define RT_STAT2(i, pred)
#define RT_STAT2_FLUSH(base)
#define RT_STAT_DECL
#define RT_STAT(i, pred)
#define RT_STAT_FLUSH(base)
#define RT_TDECL
#define RT_TMARK(i)
#define RT_TWAIT
#define RT_TFLUSH(base)
#endif

// Ray hand-out of the persistent traversal waves.  A queue is n_shards dense sub-queues (shard s = entries
// [s*shard_cap, s*shard_cap + counts[s]); shard_cap == 0: one dense array dealt out in chunks, see dealt_count) and every
// shard has its own fetch cursor; wave g works on shard g % n_shards
// (the launch has a multiple of n_shards waves).  A wave with idle lanes takes the next `need` rays of its shard with
// ONE atomic, so rays of a shard go to whichever of its waves has room: no wave is left holding a long static share
// while others have finished.  One cursor for the whole queue would saturate (~88 same-address atomics/us on MI355X);
// 64 shards keep every address far below that.  Rays handed to one wave are consecutive: the fetch is coalesced.
struct QueueShare {
    uint32_t off;      // queue index of a ray = off + its number within the shard (see `dealt`)
    uint32_t count;    // live rays in the shard
    uint32_t* cursor;  // rays of the shard handed out so far, in chunks
    uint32_t next, end; // this wave's current chunk: rays [next, end) of the shard are its to hand to its lanes
    uint32_t chunk;    // rays per cursor atomic (a multiple of 64): a returning atomic costs 1-3 us under load
    uint32_t shard, n_shards;
    bool dealt;        // shard_cap == 0: ONE dense array dealt out in chunks, chunk c belongs to shard c % n_shards
    bool exhausted;    // wave-uniform: the cursor has passed count
};
// A caller's dense array of n rays has no shards of its own.  Cutting it into n_shards consecutive pieces gives every shard
// one region of the caller's order — of an image, say — and the shard with the expensive region finishes long after the
// others (primary rays of cornellbox at 4K: 1.33 ms against 0.90 ms, tools/coherence_probe.py).  Dealing it out in chunks
// gives every shard a sample of the whole array, and a wave still gets consecutive rays.  dealt_count = rays shard s owns.
__host__ __device__ inline uint32_t dealt_count(uint64_t n, uint32_t chunk, uint32_t n_shards, uint32_t s) {
    const uint64_t full = n / chunk, rem = n % chunk;
    return (uint32_t)((full / n_shards + (s < full % n_shards ? 1u : 0u)) * chunk + (s == full % n_shards ? rem : 0u));
}
RT_DEV QueueShare queue_open(const uint32_t* __restrict__ counts, uint32_t* __restrict__ cursors, uint32_t n_shards, uint32_t shard_cap,
                             int wave_min_rays, int fetch_chunk) {
    const uint32_t gwave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (RT_TRAV_BLOCK / 64) + threadIdx.x / 64));
    const uint32_t shard = gwave % n_shards;
    QueueShare qs;
    qs.off = shard * shard_cap;
    qs.shard = shard;
    qs.n_shards = n_shards;
    qs.dealt = shard_cap == 0u;
    qs.count = counts[(size_t)shard * RT_CSTRIDE];
    qs.cursor = cursors + (size_t)shard * RT_CSTRIDE;
    qs.next = qs.end = 0;
    qs.chunk = (uint32_t)fetch_chunk;
    // short queues are left to the first waves of each shard: a wave that would get only a few rays costs more in its
    // drain (lanes idling while its last rays finish) than it adds in parallelism
    qs.exhausted = (uint64_t)(gwave / n_shards) * (uint64_t)wave_min_rays >= qs.count;
    return qs;
}
RT_DEV bool queue_has_more(const QueueShare& qs) { return qs.next < qs.end || !qs.exhausted; }
// All lanes call; lanes with `want` get the queue index of a fresh ray (true) or nothing (false: their turn comes with
// the wave's next chunk).
RT_DEV bool queue_fetch(QueueShare& qs, bool want, uint32_t* idx) {
    const unsigned long long idle = __ballot(want);
    const uint32_t need = (uint32_t)__popcll(idle);
    const uint32_t lane = __lane_id();
    if (qs.next == qs.end) { // wave-uniform
        uint32_t first = 0;
        if (lane == 0) first = atomicAdd(qs.cursor, qs.chunk);
        first = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
        qs.next = first < qs.count ? first : qs.count;
        qs.end = first + qs.chunk < qs.count ? first + qs.chunk : qs.count;
        if (first + qs.chunk >= qs.count) qs.exhausted = true;
        if (qs.dealt) qs.off = (first / qs.chunk * qs.n_shards + qs.shard) * qs.chunk - first; // wave-uniform; first is a multiple of chunk
    }
    const uint32_t i = qs.next + (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
    const bool got = want && i < qs.end;
    qs.next = qs.next + need < qs.end ? qs.next + need : qs.end;
    *idx = qs.off + i;
    return got;
}

// Origin and 1/dir of a fetched ray as register pairs; *exact = the ray needs the select-form slab test (non-finite
// 1/dir or origin: NaNs possible; also true for NaN inputs).
RT_DEV RaySlab ray_slab(const float4& o4, const float4& d4, bool* exact) {
    const float ix = 1.f / d4.x, iy = 1.f / d4.y, iz = 1.f / d4.z; // Math.h:756
    const float big = fmaxf(fmaxf(fabsf(ix), fabsf(iy)), fabsf(iz));
    const float org = fabsf(o4.x) + fabsf(o4.y) + fabsf(o4.z);
    *exact = !(big < INFINITY) || !(org < INFINITY);
    RaySlab r;
    r.oxy = f2{o4.x, o4.y};
    r.ozz = f2{o4.z, o4.z};
    r.ixy = f2{ix, iy};
    r.izz = f2{iz, iz};
    return r;
}

// When do the lanes waiting at a leaf get their turn?  Once leaf_min of them wait, once nobody is left walking, and (round 3) once
// they are as many as the lanes still walking: inner steps for a minority of the wave cost a full step each (stand-alone:
// k_closest -1.4 %, k_shadow -1 %, the two-level walks -2 % / -3 %; "half as many" is 4 % slower on the two-level walks).
RT_DEV bool keep_walking(unsigned long long at_leaf, unsigned long long walking, int leaf_min) {
    const int n_leaf = __popcll(at_leaf), n_walk = __popcll(walking);
    return n_leaf < leaf_min && n_walk != 0 && n_leaf < n_walk;
}

// Closest hit (TriangleMesh::intersect, Object.h:273-344) over a sharded ray queue, as PERSISTENT wave64s:
//   * every wave refills its idle lanes from its shard of the queue (queue_fetch), so lanes whose ray ended early do
//     not idle until the slowest ray of their wave is done;
//   * "while-while" order: all lanes walk inner nodes until few have one pending, then the pending leaves are tested
//     together — the two bodies are not interleaved per step under partial masks;
//   * the traversal stack keeps its first RT_SHORT_STACK entries in LDS (column lds[entry * RT_TRAV_BLOCK]: lanes of a
//     wave hit consecutive banks) and spills deeper entries to private memory, up to the reference's 64.
// Per ray the visit order, the culling rule (`dist < t` at push time, not re-tested at pop), right-child-first on
// ties and accept-on-equal-t are exactly the reference's, so results do not depend on how rays share a wave.
// sink(ray state) is called once per ray when it completes (RayState::q = its global queue index).  Exit: a shard's
// cursor only grows, so a wave stops fetching once it has passed `count` and leaves the loop when its last lane is done.
template <class Sink>
RT_DEV void trace_queue(const SceneView& sc, const float4* __restrict__ q_o, const float4* __restrict__ q_d,
                        const uint32_t* __restrict__ counts, uint32_t* __restrict__ cursors, uint32_t n_shards, uint32_t shard_cap,
                        lds_int* lds, lds_cf4* top, int top_bytes, Sink sink) {
    int spill[RT_STACK_TOTAL - RT_SHORT_STACK];
    RayState r;
    r.q = 0;
    r.slot = -1;
    r.t = INFINITY;
    r.tmax = INFINITY;
    r.u = r.v = 0.f;
    r.o = r.d = V3(0.f);
    RaySlab rs;
    rs.oxy = rs.ozz = rs.ixy = rs.izz = f2{0.f, 0.f};
    bool exact = false;
    int cur = RT_CUR_DONE;
    int sp = 0;
    bool wave_exact = false; // wave-uniform: some lane holds a ray that needs the select-form slab test
    const float4* __restrict__ pairs = sc.pairs;
    const float4* __restrict__ tris = sc.tris;
    RT_STAT_DECL
    RT_TDECL
    QueueShare qs = queue_open(counts, cursors, n_shards, shard_cap, sc.wave_min_rays, sc.fetch_chunk);
    for (;;) {
        if (queue_has_more(qs) && __popcll(__ballot(cur == RT_CUR_DONE)) >= sc.refill_min) {
            uint32_t idx;
            if (queue_fetch(qs, cur == RT_CUR_DONE, &idx)) {
                const float4 o4 = q_o[idx], d4 = q_d[idx];
                r.q = idx;
                r.o = V3(o4);
                r.d = V3(d4);
                rs = ray_slab(o4, d4, &exact);
                r.tmax = INFINITY;
                r.t = INFINITY;
                r.slot = -1;
                r.u = r.v = 0.f; // a miss reports (-1, inf, 0, 0), not the lane's previous hit
                sp = 0;
                cur = sc.root_ref;
            }
            wave_exact = __ballot(exact && cur != RT_CUR_DONE) != 0ull;
        }
        RT_TMARK(0);
        RT_TMARK(5);
        if (__ballot(cur != RT_CUR_DONE) == 0ull) break; // nothing fetched: the queue is exhausted
        RT_STAT(3, cur != RT_CUR_DONE); // rounds, lanes holding a ray

        // ---- inner-node steps (Object.h:322-343): two per round, so the wave-level bookkeeping above and below is
        // paid once per two node visits
#pragma unroll
        for (int rep = 0; rep < RT_CLOSEST_STEPS; rep++) {
            RT_STAT(0, cur >= 0); // inner steps
            RT_STAT(2, cur >= 0 && cur < top_bytes); // ... of which read from LDS
            if (cur >= 0) {
                f4 q0, q1, q2;
                int lref, rref;
                load_pair(pairs, top, top_bytes, cur, q0, q1, q2, lref, rref);
                RT_TWAIT;
                RT_TMARK(1);
                float ld, rd;
                bool lh, rh;
                if (!wave_exact) slab_pair<false>(q0, q1, q2, rs, ld, lh, rd, rh);
                else slab_pair<true>(q0, q1, q2, rs, ld, lh, rd, rh);
                RT_PROBE_STEP(ld);
                lh = lh && ld < r.t;
                rh = rh && rd < r.t;
                const bool left_first = ld < rd; // ties visit the right child first (Object.h:329-336)
                // next node when at least one child is hit: the nearer of two, else the one that is hit
                cur = (lh && (!rh || left_first)) ? lref : rref;
                // push the farther child / pop, or finish on an empty stack
                stack_step<RT_STACK_TOTAL>(lds, spill, lh && rh, !lh && !rh, left_first ? rref : lref, cur, sp);
                RT_TWAIT;
                RT_TMARK(2);
            }
        }

        // ---- leaves (Object.h:292-320) and completion: lanes that reached a leaf wait until enough of them have,
        // so the triangle test runs at a useful lane occupancy (wave-uniform decision)
        const unsigned long long at_leaf = __ballot(cur < 0 && cur != RT_CUR_DONE);
        if (keep_walking(at_leaf, __ballot(cur >= 0), sc.leaf_min)) {
            RT_TMARK(4);
            continue;
        }
        RT_TMARK(4);
        if (cur < 0 && cur != RT_CUR_DONE) {
            if (cur != RT_CUR_FINISH) {
                int slot = ~cur;
                bool last;
                RT_STAT(1, true); // leaf turns
                do {
                    float4 a, b, c;
                    load_tri(tris, slot, a, b, c);
                    last = (__float_as_uint(a.w) & RT_TRI_LEAF_END) != 0u;
                    float tt, u, v;
                    if (tri_accept(r.o, r.d, a, b, c, r.tmax, &tt, &u, &v)) {
                        r.tmax = tt;
                        r.t = tt;
                        r.slot = slot;
                        r.u = u;
                        r.v = v;
                    }
                    slot++;
                } while (!last);
                if (sp == 0) {
                    cur = RT_CUR_FINISH;
                } else {
                    sp--;
                    if (sp < RT_SHORT_STACK) cur = lds[sp * RT_TRAV_BLOCK];
                    else cur = spill[sp - RT_SHORT_STACK];
                }
            }
            if (cur == RT_CUR_FINISH) {
                sink(r);
                cur = RT_CUR_DONE;
                exact = false;
            }
        }
        RT_TWAIT;
        RT_TMARK(3);
    }
    RT_STAT_FLUSH(0);
    RT_TFLUSH(0);
}

#define RT_CUR_HIT ((int)0x80000002)    // occlusion_queue: an occluder was found
#ifndef RT_OCC_STEPS
#define RT_OCC_STEPS 2 // inner steps per round of the occlusion walk (1: 6.65, 2: 6.71, 3: 6.76, 4: 6.72 Grays/s — noise level)
#endif
// Boolean occlusion (trace_scene as used at main.cpp:250,281) over a sharded ray queue: the closest walk's loop — two
// inner steps per round for all lanes, then the lanes at a leaf test their triangles together; persistent waves,
// per-shard fetch cursors, LDS/private stack as in trace_queue — ending a ray at the first accepted triangle, as the
// reference's walk does (near child first, Object.h:328-337).  sink(queue index, tag = the ray's dir.w bits, occluded)
// is called once per ray.
//
// ONE step outside the reference's procedure: a child box whose entry distance exceeds the segment's end by more than
// a margin (dist >= tmax * 1.0001 + 1e-6) is not entered.  The reference enters it (its cull value `t` is infinite until a hit, Object.h:281,325-326) and then
// rejects every triangle in it whose computed t exceeds tmax (Object.h:309).  A triangle lies inside its leaf box, so
// its true distance is at least the box's entry distance; the answers can differ only if rounding makes a triangle's
// COMPUTED t fall at or below tmax while the computed entry distance of a box around it lies beyond tmax plus the
// margin (without the margin this happens at the ulp level, e.g. for tmax = the computed t of a triangle whose box is
// flat; with it, only for a ray grazing a large triangle so closely that its t is meaningless).  No such ray has been seen: the
// parity tests compare this walk with the oracle's reference-faithful walk, flag for flag (tests/test_gpu_trace.py,
// test_golden.py, test_gpu_scale.py: 2.4 M segments, random and light-directed).  It removes 19 % of the inner visits
// and 41 % of the triangle tests of the bench workload's shadow rays (they end 1e-3 in front of a tessellated light).
//
// (Until round 1's last day this was an order-free walk that parked hit leaves in a per-lane pending list in LDS and
// tested them in batches — 80 KB of LDS per workgroup.  With everything else in place the plain loop is 4 % faster alone
// and, at 48 KB, shares the CUs better with the other passes' kernels: 6.55 -> 6.75 Grays/s.)
template <class Sink>
RT_DEV void occlusion_queue(const SceneView& sc, const float4* __restrict__ q_o, const float4* __restrict__ q_d,
                               const uint32_t* __restrict__ counts, uint32_t* __restrict__ cursors, uint32_t n_shards, uint32_t shard_cap,
                               lds_int* lds, lds_cf4* top, int top_bytes, Sink sink) {
    int spill[RT_STACK_TOTAL - RT_SHORT_STACK];
    V3 o(0.f), d(0.f);
    RaySlab rs;
    rs.oxy = rs.ozz = rs.ixy = rs.izz = f2{0.f, 0.f};
    float tmax = 0.f, tcull = 0.f;
    uint32_t q = 0, tag = 0;
    bool exact = false, wave_exact = false;
    int cur = RT_CUR_DONE, sp = 0;
    const float4* __restrict__ pairs = sc.pairs;
    const float4* __restrict__ tris = sc.tris;
    RT_STAT_DECL
    RT_TDECL
    QueueShare qs = queue_open(counts, cursors, n_shards, shard_cap, sc.wave_min_rays, sc.fetch_chunk);
    for (;;) {
        if (queue_has_more(qs) && __popcll(__ballot(cur == RT_CUR_DONE)) >= sc.refill_min) {
            uint32_t idx;
            if (queue_fetch(qs, cur == RT_CUR_DONE, &idx)) {
                const float4 o4 = q_o[idx], d4 = q_d[idx];
                q = idx;
                tag = __float_as_uint(d4.w);
                o = V3(o4);
                d = V3(d4);
                rs = ray_slab(o4, d4, &exact);
                tmax = o4.w;
                tcull = (tmax > 0.f && tmax < INFINITY) ? tmax * 1.0001f + 0.000001f : (tmax <= 0.f ? tmax : INFINITY);
                sp = 0;
                cur = sc.root_ref;
            }
            wave_exact = __ballot(exact && cur != RT_CUR_DONE) != 0ull;
        }
        RT_TMARK(0);
        RT_TMARK(5);
        if (__ballot(cur != RT_CUR_DONE) == 0ull) break;
        RT_STAT(3, cur != RT_CUR_DONE);
#pragma unroll
        for (int rep = 0; rep < RT_OCC_STEPS; rep++) {
            RT_STAT(0, cur >= 0);
            RT_STAT(2, cur >= 0 && cur < top_bytes);
            if (cur >= 0) {
                f4 q0, q1, q2;
                int lref, rref;
                load_pair(pairs, top, top_bytes, cur, q0, q1, q2, lref, rref);
                RT_TWAIT;
                RT_TMARK(1);
                float ld, rd;
                bool lh, rh;
                if (!wave_exact) slab_pair<false>(q0, q1, q2, rs, ld, lh, rd, rh);
                else slab_pair<true>(q0, q1, q2, rs, ld, lh, rd, rh);
                RT_PROBE_STEP(ld);
                lh = lh && ld < tcull;
                rh = rh && rd < tcull;
                const bool left_first = ld < rd;
                cur = (lh && (!rh || left_first)) ? lref : rref;
                stack_step<RT_STACK_TOTAL>(lds, spill, lh && rh, !lh && !rh, left_first ? rref : lref, cur, sp);
                RT_TWAIT;
                RT_TMARK(2);
            }
        }
        const unsigned long long at_leaf = __ballot(cur < 0 && cur != RT_CUR_DONE);
        if (keep_walking(at_leaf, __ballot(cur >= 0), sc.leaf_min)) {
            RT_TMARK(4);
            continue;
        }
        RT_TMARK(4);
        if (cur < 0 && cur != RT_CUR_DONE) {
            if (cur != RT_CUR_FINISH) {
                int slot = ~cur;
                bool last, stop = false;
                RT_STAT(1, true);
                do {
                    float4 a, b, c;
                    load_tri(tris, slot, a, b, c);
                    last = (__float_as_uint(a.w) & RT_TRI_LEAF_END) != 0u;
                    float tt, u, v;
                    if (tri_accept(o, d, a, b, c, tmax, &tt, &u, &v)) stop = last = true;
                    slot++;
                } while (!last);
                if (stop) {
                    cur = RT_CUR_HIT;
                } else if (sp == 0) {
                    cur = RT_CUR_FINISH;
                } else {
                    sp--;
                    if (sp < RT_SHORT_STACK) cur = lds[sp * RT_TRAV_BLOCK];
                    else cur = spill[sp - RT_SHORT_STACK];
                }
            }
            if (cur == RT_CUR_FINISH || cur == RT_CUR_HIT) {
                sink(q, tag, cur == RT_CUR_HIT);
                cur = RT_CUR_DONE;
                exact = false;
            }
        }
        RT_TWAIT;
        RT_TMARK(3);
    }
    RT_STAT_FLUSH(8);
    RT_TFLUSH(8);
}

// ------------------------------------------------ four-wide occlusion walk
// The boolean answer of trace_scene (Scene.h:113-129 as used at main.cpp:250,281; the walk is Object.h:322-343) is "some
// triangle in a REACHED leaf is accepted", and the set of reached leaves does not depend on how the walk gets to them:
//   (1) a parent's box is the componentwise min / max of its children's boxes (rt_scene_create checks that every box lies
//       inside its parent's before it builds the records below);
//   (2) the slab test is MONOTONE in the box: fl(b - o) is non-decreasing in b and fl(y * inv) is monotone in y for a
//       finite non-zero inv, so for boxes C containing B (as fp32 numbers, per component) tnear(C) <= tnear(B) and
//       tfar(C) >= tfar(B) hold IN fp32 — whenever B passes `!(tnear > tfar || tfar < 0) && tnear < tcull`, C does;
//   (3) so a leaf whose own box passes has every ancestor passing: the binary walk reaches a leaf iff the leaf's OWN box
//       passes, whatever the order.
// The occlusion walks therefore use a four-wide collapse of the tree (the inner child with the largest surface is opened
// until a record holds four children) whose child boxes are CONSERVATIVE 8-bit boxes on a grid of the record's own extent
// (each decoded plane is verified on the host, with the device's arithmetic, to lie on or outside the exact plane): by (2)
// the walk reaches a superset of the leaves the reference reaches, and every leaf record carries its EXACT box — the one
// the reference tests before it enters the leaf (Object.h:325-326) — which is tested, with the binary walk's own
// arithmetic, before the leaf's triangles are.  By (3) that single test decides reached / not reached as the reference
// does: same flags, bit for bit, at a quarter of the bytes per child and less than half the dependent visits per ray.  (The
// measured reason for the format: the walks are bound by the LDS / vector-L1 transactions a visit makes — uncompressed
// 128-byte quad records halved the visits, doubled the cost of each, and changed nothing.)
// Rays that need the select-form slab test (`exact`: non-finite 1/dir or origin, where NaNs void (2)) walk the PAIR
// records in the same loop with the same stack.
//
// One buffer of 64-byte records ("onodes"), quad records first (top of the tree first), then one leaf record per triangle
// slot.  A ref is the record's byte offset, bit 0 set for a leaf record, bit 1 for "box already tested" (refs of the pair
// walk's lanes).
//   quad record (16 words):  [0..2] origin  [3] biased exponents of the grid steps, x | y << 8 | z << 16
//     [4] lo.x of children 0..3 (one byte each)  [5] lo.y  [6] lo.z  [7] hi.x  [8] hi.y  [9] hi.z
//     [10..13] child refs (RT_REF_NONE: no child; a record holds 2 to 4, empty slots last)  [14..15] spare
//     plane = fma(q, step, origin): lo planes <= the exact min, hi planes >= the exact max
//   leaf record (16 words):  [0..5] exact box min, max of the LEAF (first triangle's record only)
//     [6..8] v0  [9..11] e1  [12..14] e2 (Object.h:247-248)  [15] bit 0: last triangle of its leaf
// Quad records [0, RT_QTOP) are read from LDS, kept there as FOUR part arrays (see load_pair).
#define RT_ONODE_BYTES 64
#ifndef RT_QTOP
#define RT_QTOP 256
#endif
#define RT_QTOP_PART_BYTES (RT_QTOP * 16)
#define RT_QTOP_LDS_BYTES (RT_QTOP * 56)
#ifndef RT_QSHORT_STACK
#define RT_QSHORT_STACK 12  // stack entries per lane in LDS (a visit pushes up to three: 8: 43.1, 12: 41.8, 16: 41.9 ms per step)
#endif
#define RT_QSTACK_TOTAL 96  // rt_scene_create computes a tree's worst case and keeps a deeper one on the pair walk
#define RT_REF_NONE ((int)0x80000000)
#define RT_REF_LEAF 1
#define RT_REF_TESTED 2

typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) u4 lds_cu4;

RT_DEV void load_quad(const float4* __restrict__ onodes, lds_cf4* top, int top_bytes, int cur, u4& w0, u4& w1, u4& w2, int& r2, int& r3) {
    if (cur < top_bytes) {
        const __attribute__((address_space(3))) char* base = (const __attribute__((address_space(3))) char*)top;
        const uint32_t at = (uint32_t)cur >> 2; // record index * 16
        w0 = *(lds_cu4*)(base + at);
        w1 = *(lds_cu4*)(base + at + RT_QTOP_PART_BYTES);
        w2 = *(lds_cu4*)(base + at + 2 * RT_QTOP_PART_BYTES);
        const unsigned long long rr = *(const __attribute__((address_space(3))) unsigned long long*)(base + ((uint32_t)cur >> 3) + 3 * RT_QTOP_PART_BYTES);
        r2 = (int)(uint32_t)rr;
        r3 = (int)(uint32_t)(rr >> 32);
    } else {
        const u4* p = (const u4*)((const char*)onodes + (uint32_t)cur);
        w0 = p[0];
        w1 = p[1];
        w2 = p[2];
        const unsigned long long rr = *(const unsigned long long*)(p + 3);
        r2 = (int)(uint32_t)rr;
        r3 = (int)(uint32_t)(rr >> 32);
    }
}

// byte k of w as a float (v_cvt_f32_ubyte<k>)
template <int K>
RT_DEV float ubyte(uint32_t w) { return (float)((w >> (8 * K)) & 0xffu); }

// Child K of a quad record against the ray: its six decoded planes, then the slab arithmetic of Bounds::intersect
// (Math.h:757-770) in the form slab_pair gives it.  One child at a time: six plane words, twelve temporaries and the four
// entry distances are all a visit holds (the per-axis form kept 24 distances live and did not fit 64 registers).
struct QuadGrid {
    float sx, sy, sz, gx, gy, gz; // grid steps and origin
};
template <int K>
RT_DEV float quad_child(const u4& w1, const u4& w2, const QuadGrid& g, const V3& o, const V3& inv, float tcull, bool& hit) {
    const float ax = (__builtin_fmaf(ubyte<K>(w1.x), g.sx, g.gx) - o.x) * inv.x, bx = (__builtin_fmaf(ubyte<K>(w1.w), g.sx, g.gx) - o.x) * inv.x;
    const float ay = (__builtin_fmaf(ubyte<K>(w1.y), g.sy, g.gy) - o.y) * inv.y, by = (__builtin_fmaf(ubyte<K>(w2.x), g.sy, g.gy) - o.y) * inv.y;
    const float az = (__builtin_fmaf(ubyte<K>(w1.z), g.sz, g.gz) - o.z) * inv.z, bz = (__builtin_fmaf(ubyte<K>(w2.y), g.sz, g.gz) - o.z) * inv.z;
    const float tn = vmax3(vmin(ax, bx), vmin(ay, by), vmin(az, bz));
    const float tf = vmin3(vmax(ax, bx), vmax(ay, by), vmax(az, bz));
    hit = !(tn > tf || tf < 0.f) && tn < tcull;
    return tn;
}

// compare-exchange of two (distance, ref) pairs: afterwards ka <= kb
RT_DEV void order2(float& ka, int& ra, float& kb, int& rb) {
    const bool sw = kb < ka;
    const float k = sw ? kb : ka;
    const int r = sw ? rb : ra;
    kb = sw ? ka : kb;
    rb = sw ? ra : rb;
    ka = k;
    ra = r;
}

// The exact box of a leaf against the ray, as slab_pair<false> computes it for a child of a pair record (same operations
// on the same operands: v_max3 / v_min3 do not depend on operand order for non-NaN inputs).
RT_DEV bool leaf_box_passes(const f4& r0, const f4& r1, const V3& o, const V3& inv, float tcull) {
    const float ax = (r0.x - o.x) * inv.x, ay = (r0.y - o.y) * inv.y, az = (r0.z - o.z) * inv.z;
    const float bx = (r0.w - o.x) * inv.x, by = (r1.x - o.y) * inv.y, bz = (r1.y - o.z) * inv.z;
    const float tn = vmax3(vmin(ax, bx), vmin(ay, by), vmin(az, bz));
    const float tf = vmin3(vmax(ax, bx), vmax(ay, by), vmax(az, bz));
    return !(tn > tf || tf < 0.f) && tn < tcull;
}

// occlusion_queue over the onode records.  sink as there.  leaf_base: byte offset of slot 0's leaf record.
template <class Sink>
RT_DEV void occlusion_queue4(const SceneView& sc, const float4* __restrict__ q_o, const float4* __restrict__ q_d,
                             const uint32_t* __restrict__ counts, uint32_t* __restrict__ cursors, uint32_t n_shards, uint32_t shard_cap,
                             lds_int* lds, lds_cf4* top, int top_bytes, Sink sink) {
    int spill[RT_QSTACK_TOTAL - RT_QSHORT_STACK];
    V3 o(0.f), d(0.f), inv(0.f); // (origin and 1/dir as scalars: the pair walk's register pairs are rebuilt where the rare
                                 // select-form ray needs them)
    float tmax = 0.f, tcull = 0.f;
    uint32_t q = 0, tag = 0;
    bool exact = false;
    int cur = RT_CUR_DONE, sp = 0;
    const float4* __restrict__ pairs = sc.pairs;
    const float4* __restrict__ onodes = sc.onodes;
    const int leaf_base = sc.leaf_base;
    RT_STAT_DECL
    RT_TDECL
    auto push = [&](int v) {
        if (sp < RT_QSHORT_STACK) lds[sp * RT_TRAV_BLOCK] = v;
        else if (sp < RT_QSTACK_TOTAL) spill[sp - RT_QSHORT_STACK] = v;
        if (sp < RT_QSTACK_TOTAL) sp++; // (rt_scene_create keeps deeper trees off this walk; were one to get here, the entry is lost, never memory)
    };
    auto pop = [&]() {
        if (sp == 0) return RT_CUR_FINISH;
        sp--;
        return sp < RT_QSHORT_STACK ? (int)lds[sp * RT_TRAV_BLOCK] : spill[sp - RT_QSHORT_STACK];
    };
    // a pair-walk ref (exact rays) as an onode ref: leaves become leaf records whose box the pair walk has tested
    auto pair_ref = [&](int r) { return r >= 0 ? r : (leaf_base + (~r) * RT_ONODE_BYTES) | RT_REF_LEAF | RT_REF_TESTED; };
    QueueShare qs = queue_open(counts, cursors, n_shards, shard_cap, sc.wave_min_rays, sc.fetch_chunk);
    for (;;) {
        if (queue_has_more(qs) && __popcll(__ballot(cur == RT_CUR_DONE)) >= sc.refill_min) {
            uint32_t idx;
            if (queue_fetch(qs, cur == RT_CUR_DONE, &idx)) {
                const float4 o4 = q_o[idx], d4 = q_d[idx];
                q = idx;
                tag = __float_as_uint(d4.w);
                o = V3(o4);
                d = V3(d4);
                const RaySlab rs = ray_slab(o4, d4, &exact);
                inv = V3(rs.ixy.x, rs.ixy.y, rs.izz.x);
                tmax = o4.w;
                tcull = (tmax > 0.f && tmax < INFINITY) ? tmax * 1.0001f + 0.000001f : (tmax <= 0.f ? tmax : INFINITY);
                sp = 0;
                cur = exact ? pair_ref(sc.root_ref) : sc.qroot;
            }
        }
        RT_TMARK(0);
        RT_TMARK(5);
        if (__ballot(cur != RT_CUR_DONE) == 0ull) break;
        RT_STAT(3, cur != RT_CUR_DONE);
#pragma unroll
        for (int rep = 0; rep < RT_OCC_STEPS; rep++) {
            const bool inner = cur >= 0 && !(cur & RT_REF_LEAF);
            RT_STAT(0, inner);
            RT_STAT(2, inner && cur < top_bytes && !exact);
            if (inner && !exact) {
                u4 w0, w1, w2;
                int r0, r1, r2, r3;
                load_quad(onodes, top, top_bytes, cur, w0, w1, w2, r2, r3);
                RT_TWAIT;
                RT_TMARK(1);
                r0 = (int)w2.z;
                r1 = (int)w2.w;
                QuadGrid g;
                g.sx = __uint_as_float((w0.w & 0xffu) << 23);
                g.sy = __uint_as_float((w0.w & 0xff00u) << 15);
                g.sz = __uint_as_float((w0.w & 0xff0000u) << 7);
                g.gx = __uint_as_float(w0.x);
                g.gy = __uint_as_float(w0.y);
                g.gz = __uint_as_float(w0.z);
                bool h0, h1, h2, h3;
                float k0 = quad_child<0>(w1, w2, g, o, inv, tcull, h0);
                float k1 = quad_child<1>(w1, w2, g, o, inv, tcull, h1);
                float k2 = quad_child<2>(w1, w2, g, o, inv, tcull, h2);
                float k3 = quad_child<3>(w1, w2, g, o, inv, tcull, h3);
                h2 = h2 && r2 != RT_REF_NONE;
                h3 = h3 && r3 != RT_REF_NONE;
                RT_PROBE_STEP(k0);
                const int n = (h0 ? 1 : 0) + (h1 ? 1 : 0) + (h2 ? 1 : 0) + (h3 ? 1 : 0);
                k0 = h0 ? k0 : INFINITY; // (a hit has k < tcull <= inf: misses sort behind every hit)
                k1 = h1 ? k1 : INFINITY;
                k2 = h2 ? k2 : INFINITY;
                k3 = h3 ? k3 : INFINITY;
                order2(k0, r0, k1, r1);
                order2(k2, r2, k3, r3);
                order2(k0, r0, k2, r2);
                order2(k1, r1, k3, r3);
                order2(k1, r1, k2, r2);
                // nearest child next, the others on the stack with the nearer ones on top (near first as Object.h:328-337
                // has it; the answer does not depend on the order)
                if (__ballot(sp > RT_QSHORT_STACK - 3) == 0ull) { // wave-uniform: nobody can leave the LDS part
                    if (n == 4) lds[sp * RT_TRAV_BLOCK] = r3;
                    if (n >= 3) lds[(sp + n - 3) * RT_TRAV_BLOCK] = r2;
                    if (n >= 2) lds[(sp + n - 2) * RT_TRAV_BLOCK] = r1;
                    const bool take = n == 0 && sp > 0;
                    int next = RT_CUR_FINISH;
                    if (take) next = lds[(sp - 1) * RT_TRAV_BLOCK];
                    cur = n == 0 ? next : r0;
                    sp += (n > 0 ? n - 1 : 0) - (take ? 1 : 0);
                } else {
                    if (n == 4) push(r3);
                    if (n >= 3) push(r2);
                    if (n >= 2) push(r1);
                    cur = n == 0 ? pop() : r0;
                }
                RT_TWAIT;
                RT_TMARK(2);
            } else if (inner) { // select-form rays: the binary walk of occlusion_queue on the pair records
                f4 q0, q1, q2;
                int lref, rref;
                load_pair(pairs, top, 0, cur, q0, q1, q2, lref, rref);
                float ld, rd;
                bool lh, rh;
                RaySlab rs;
                rs.oxy = f2{o.x, o.y};
                rs.ozz = f2{o.z, o.z};
                rs.ixy = f2{inv.x, inv.y};
                rs.izz = f2{inv.z, inv.z};
                slab_pair<true>(q0, q1, q2, rs, ld, lh, rd, rh);
                lh = lh && ld < tcull;
                rh = rh && rd < tcull;
                const bool left_first = ld < rd;
                cur = pair_ref((lh && (!rh || left_first)) ? lref : rref);
                if (lh && rh) push(pair_ref(left_first ? rref : lref));
                else if (!lh && !rh) cur = pop();
            }
        }
        const unsigned long long at_leaf = __ballot(cur >= 0 ? (cur & RT_REF_LEAF) != 0 : cur == RT_CUR_FINISH);
        if (keep_walking(at_leaf, __ballot(cur >= 0 && !(cur & RT_REF_LEAF)), sc.leaf_min)) {
            RT_TMARK(4);
            continue;
        }
        RT_TMARK(4);
        if (cur >= 0 ? (cur & RT_REF_LEAF) != 0 : cur == RT_CUR_FINISH) {
            if (cur != RT_CUR_FINISH) {
                bool test_box = !(cur & RT_REF_TESTED);
                uint32_t at = (uint32_t)cur & ~3u;
                bool last, stop = false;
                RT_STAT(1, true);
                do {
                    const f4* p = (const f4*)((const char*)onodes + at);
                    const f4 r0 = p[0], r1 = p[1], r2 = p[2], r3 = p[3];
                    asm volatile("" ::"v"(r0.x), "v"(r1.x), "v"(r2.x), "v"(r3.x)); // (all four requested before the first is used)
                    last = (__float_as_uint(r3.w) & 1u) != 0u;
                    if (test_box && !leaf_box_passes(r0, r1, o, inv, tcull)) break; // the reference never enters this leaf
                    test_box = false;
                    float tt, u, v;
                    if (tri_accept(o, d, make_float4(r1.z, r1.w, r2.x, 0.f), make_float4(r2.y, r2.z, r2.w, 0.f), make_float4(r3.x, r3.y, r3.z, 0.f), tmax, &tt, &u, &v))
                        stop = last = true;
                    at += RT_ONODE_BYTES;
                } while (!last);
                cur = stop ? RT_CUR_HIT : pop();
            }
            if (cur == RT_CUR_FINISH || cur == RT_CUR_HIT) {
                sink(q, tag, cur == RT_CUR_HIT);
                cur = RT_CUR_DONE;
                exact = false;
            }
        }
        RT_TWAIT;
        RT_TMARK(3);
    }
    RT_STAT_FLUSH(8);
    RT_TFLUSH(8);
}

// ------------------------------------------------------ two-level scenes
// Scene::closest_hit over Scene::instances (Scene.cpp:17-77): the TLAS walk; at a leaf Instance::intersect
// (Object.h:380-394) moves the ray into model space and runs the Geometry's own test — an analytic primitive
// (Object.h:26-243) or, for a TriangleMesh, the walk of ITS BVH (Object.h:273-344).
//
// Device layout.  TLAS records and every mesh's records share ONE pair array, TLAS first: a stack entry is a byte offset
// whatever its level, and the LDS copy of the array's head holds the whole TLAS plus the top of the first mesh.  An
// instance is eight float4 (rt_scene_create_instanced repacks rt_instance): [0..2] rows of world-to-model (row i =
// inv_m[i], inv_m[4+i], inv_m[8+i], inv_m[12+i]), [3] geometry, material, blas_root (bits), area, [4..6] rows of
// model-to-world, [7] spare; its eleven geometry parameters are three more float4 in a second array, read only for
// analytic primitives.  (Divergent loads cost per instruction: 4 x 16 bytes per instance visit instead of 30 dwords.)
#define RT_INST_QUADS 8
#define RT_TL_STACK_TOTAL 128 // 64 for the TLAS (Scene.cpp:19) + 64 for the mesh (Object.h:275)
#define RT_CUR_EXIT ((int)0x80000003) // stack marker beneath a mesh instance's entries: popping it ends the mesh walk

struct Xf { // three rows of a 3x4 transform
    float4 r0, r1, r2;
};
RT_DEV Xf load_xf(const float4* q) {
    Xf x;
    x.r0 = q[0];
    x.r1 = q[1];
    x.r2 = q[2];
    return x;
}
RT_DEV V3 xf_point(const Xf& m, const V3& p) { // mat4 * vec4(p, 1), Math.h:368-390: same terms in the same order
    return V3(m.r0.x * p.x + m.r0.y * p.y + m.r0.z * p.z + m.r0.w * 1.f, m.r1.x * p.x + m.r1.y * p.y + m.r1.z * p.z + m.r1.w * 1.f,
              m.r2.x * p.x + m.r2.y * p.y + m.r2.z * p.z + m.r2.w * 1.f);
}
RT_DEV V3 xf_model_vector(const Xf& im, const V3& v) { // Transform::to_model_vector, Math.h:675-683 (im = world-to-model)
    return V3(im.r0.x * v.x + im.r0.y * v.y + im.r0.z * v.z, im.r1.x * v.x + im.r1.y * v.y + im.r1.z * v.z, im.r2.x * v.x + im.r2.y * v.y + im.r2.z * v.z);
}
RT_DEV V3 xf_world_normal(const Xf& im, const V3& n) { // Transform::to_world_normal, Math.h:667-674: transpose of world-to-model
    return V3(im.r0.x * n.x + im.r1.x * n.y + im.r2.x * n.z, im.r0.y * n.x + im.r1.y * n.y + im.r2.y * n.z, im.r0.z * n.x + im.r1.z * n.y + im.r2.z * n.z);
}

struct Hit2 {
    float t, u, v;
    int slot;     // >= 0: triangle slot of a mesh instance; -2: analytic primitive; -1: nothing
    int inst;
    V3 n;         // analytic primitive: model-space normal after set_face_normal / set_forward_face
    bool front;   // ... and its front_face
};

RT_DEV float comp(const V3& v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

// Geometry::intersect of the analytic primitives with the ray in model space.  (o, d, tmin, tmax) as the reference
// passes them; on a hit t, n (oriented), front, u, v are set.  Comparisons the reference writes against double
// constants are made in double.  p = the instance's parameters (rt_instance.p).
template <class Store>
RT_DEV bool prim_intersect(int geometry, const float* p, const V3& o, const V3& d, float tmin, float tmax, float* t_out, Store&& store) {
    switch (geometry) {
    case RT_GEO_SPHERE: { // Object.h:34-55
        const float radius = p[0];
        const float a = dot(d, d);
        const float halfb = dot(o, d);
        const float c = dot(o, o) - radius * radius;
        const float disc = halfb * halfb - a * c;
        if (disc < 0.f) return false;
        const float sq = sqrtf(disc);
        float root = (-halfb - sq) / a;
        if (root < tmin || root > tmax) {
            root = (-halfb + sq) / a;
            if (root < tmin || root > tmax) return false;
        }
        const V3 out = (o + d * root) / radius;
        const bool front = dot(d, out) < 0.f;
        *t_out = root;
        store(0.f, 0.f, front ? out : -out, front);
        return true;
    }
    case RT_GEO_RECTANGLE: { // Object.h:93-120
        const V3 N(p), T(p + 3), B(p + 6);
        const float denom = dot(N, d);
        if ((double)fabsf(denom) < 0.01) return false;
        const float dist = dot(o, N);
        const float t = -dist / denom;
        if (t < tmin || t > tmax) return false;
        const V3 pi = o + d * t;
        const float ud = dot(pi, T);
        if (ud < -p[9] || ud > p[9]) return false;
        const float vd = dot(pi, B);
        if (vd < -p[10] || vd > p[10]) return false;
        const bool front = dot(d, N) < 0.f;
        *t_out = t;
        store(ud / p[9], vd / p[10], front ? N : -N, front);
        return true;
    }
    case RT_GEO_DISK: { // Object.h:141-162
        const V3 N(p);
        const float denom = dot(N, d);
        if ((double)fabsf(denom) < 0.01) return false;
        const float dist = dot(o, N);
        const float t = -dist / denom;
        if (t < tmin || t > tmax) return false;
        const float r = len(o + d * t);
        if (r > p[3] || r < p[4]) return false;
        const bool front = dot(d, N) < 0.f;
        *t_out = t;
        store(0.f, 0.f, front ? N : -N, front);
        return true;
    }
    case RT_GEO_BOX: { // Object.h:175-204
        float gm = -INFINITY;
        int idx = 0;
        for (int a = 0; a < 3; a++) {
            const float oa = comp(o, a), da = comp(d, a);
            const float x0 = (p[a] - oa) / da, x1 = (p[3 + a] - oa) / da;
            const float t0 = fminf(x0, x1), t1 = fmaxf(x0, x1);
            tmin = fmaxf(t0, tmin);
            tmax = fminf(t1, tmax);
            if (tmax <= tmin) return false;
            if (tmin > gm) {
                gm = tmin;
                idx = a;
            }
        }
        const float s = comp(d, idx) > 0.f ? -1.f : 1.f;
        const V3 out(idx == 0 ? s : 0.f, idx == 1 ? s : 0.f, idx == 2 ? s : 0.f);
        const bool front = dot(d, out) < 0.f;
        *t_out = gm;
        store(0.f, 0.f, front ? out : -out, front);
        return true;
    }
    case RT_GEO_CYLINDER: { // Object.h:216-239
        const float radius = p[0], ymin = p[1], ymax = p[2];
        const float a = d.x * d.x + d.z * d.z;
        const float b = 2.f * (o.x * d.x + o.z * d.z);
        const float c = o.x * o.x + o.z * o.z - radius * radius;
        const float disc = b * b - 4.f * a * c;
        if (disc < 0.f) return false;
        const float sq = sqrtf(disc);
        float root = (-b - sq) / (2.f * a);
        V3 pt = o + d * root;
        if (root < tmin || root > tmax || pt.y < ymin || pt.y > ymax) {
            root = (-b + sq) / (2.f * a);
            pt = o + d * root;
            if (root < tmin || root > tmax || pt.y < ymin || pt.y > ymax) return false;
        }
        const V3 out = V3(pt.x, 0.f, pt.z) / radius;
        *t_out = root;
        store(0.f, 0.f, dot(d, out) < 0.f ? out : -out, true); // set_forward_face, Def.h:35-39
        return true;
    }
    default: return false;
    }
}

// The two-level walk over a sharded ray queue, as persistent wave64s with lane refill (the scheme of trace_queue).
// One loop serves both levels.  A lane holds its CURRENT ray: the world ray in the TLAS, the model-space ray inside a
// mesh instance.  Entering a mesh pushes RT_CUR_EXIT beneath the mesh's own entries; popping it means the mesh is
// finished: the world ray is read back from the queue and the TLAS walk carries on.  Cull values as the reference has
// them: the TLAS compares box distances with the closest hit over all instances (Scene.cpp:50-51), a mesh walk with
// the closest hit found IN THIS MESH so far (Object.h:281,325-326: its local t starts infinite), while triangles are
// accepted against the shrinking tmax handed down (Object.h:312-314).  Visit order, right-first on ties, accept on
// equal t: the reference's, per ray, whatever rays share the wave.
// ANY = true answers Scene::trace_scene as a boolean (Scene.h:113-129) for the segment [0, origin.w]: the walk ends at
// the first accepted hit, and boxes (of either level) entered beyond the segment's end plus the margin of
// occlusion_queue are skipped — the reference loops over ALL instances instead (no TLAS on that path); an instance
// whose world box the segment misses cannot be hit by it, up to the rounding of the box test.
// sink(queue index, hit record, found) once per ray.
template <bool ANY, class Sink>
RT_DEV void trace_queue2(const SceneView& sc, const float4* __restrict__ q_o, const float4* __restrict__ q_d,
                         const uint32_t* __restrict__ counts, uint32_t* __restrict__ cursors, uint32_t n_shards, uint32_t shard_cap,
                         lds_int* lds, lds_int* hrec, lds_cf4* top, int top_bytes, Sink sink) {
    int spill[RT_TL_STACK_TOTAL - RT_SHORT_STACK];
    V3 o(0.f), d(0.f); // the lane's current ray
    RaySlab rs;
    rs.oxy = rs.ozz = rs.ixy = rs.izz = f2{0.f, 0.f};
    // The closest hit's record beyond its distance (u, v, slot, instance | front, normal) is written when a hit is accepted
    // and read once, when the ray completes: it lives in LDS (hrec: seven words per lane, column layout like the stack),
    // not in seven registers of a walk that is compiled for 64 (the register version spilled 31 of them to scratch).
    // Occlusion queries (ANY) keep no record at all.
    auto hit_store = [&](float u, float v, int slot, int inst_front, const V3& n) {
        if (ANY) return;
        hrec[0 * RT_TRAV_BLOCK] = __float_as_int(u);
        hrec[1 * RT_TRAV_BLOCK] = __float_as_int(v);
        hrec[2 * RT_TRAV_BLOCK] = slot;
        hrec[3 * RT_TRAV_BLOCK] = inst_front;
        hrec[4 * RT_TRAV_BLOCK] = __float_as_int(n.x);
        hrec[5 * RT_TRAV_BLOCK] = __float_as_int(n.y);
        hrec[6 * RT_TRAV_BLOCK] = __float_as_int(n.z);
    };
    bool found = false;
    float t_best = INFINITY, t_local = INFINITY, tcull = INFINITY;
    uint32_t q = 0;
    bool exact = false, wave_exact = false;
    int cur = RT_CUR_DONE, sp = 0;
    int in_inst = -1; // >= 0: inside this mesh instance
    const float4* __restrict__ pairs = sc.pairs;
    const float4* __restrict__ tris = sc.tris;
    auto push = [&](int v) {
        if (sp < RT_SHORT_STACK) lds[sp * RT_TRAV_BLOCK] = v;
        else if (sp < RT_TL_STACK_TOTAL) spill[sp - RT_SHORT_STACK] = v;
        if (sp < RT_TL_STACK_TOTAL) sp++; // (deeper than the reference's own stacks: the entry is lost, never memory)
    };
    auto pop = [&]() {
        if (sp == 0) return RT_CUR_FINISH;
        sp--;
        return sp < RT_SHORT_STACK ? (int)lds[sp * RT_TRAV_BLOCK] : spill[sp - RT_SHORT_STACK];
    };
    RT_STAT_DECL
    RT_STAT2_DECL
    QueueShare qs = queue_open(counts, cursors, n_shards, shard_cap, sc.wave_min_rays, sc.fetch_chunk);
    for (;;) {
        if (queue_has_more(qs) && __popcll(__ballot(cur == RT_CUR_DONE)) >= sc.refill_min) {
            uint32_t idx;
            if (queue_fetch(qs, cur == RT_CUR_DONE, &idx)) {
                const float4 o4 = q_o[idx], d4 = q_d[idx];
                q = idx;
                o = V3(o4);
                d = V3(d4);
                rs = ray_slab(o4, d4, &exact);
                const float tmax_in = ANY ? o4.w : INFINITY;
                t_best = tmax_in; // closest_so_far (Scene.h:117) / Scene::closest_hit's t (Scene.cpp:25)
                t_local = INFINITY;
                tcull = (tmax_in > 0.f && tmax_in < INFINITY) ? tmax_in * 1.0001f + 0.000001f : (tmax_in <= 0.f ? tmax_in : INFINITY);
                found = false;
                sp = 0;
                in_inst = -1;
                cur = sc.n_inst > 0 ? sc.tl_root : RT_CUR_FINISH;
            }
            wave_exact = __ballot(exact && cur != RT_CUR_DONE) != 0ull;
        }
        if (__ballot(cur != RT_CUR_DONE) == 0ull) break;
        RT_STAT(3, cur != RT_CUR_DONE);

#pragma unroll
        for (int rep = 0; rep < RT_CLOSEST_STEPS; rep++) { // inner nodes of either level (Scene.cpp:47-68 / Object.h:322-343)
            RT_STAT(0, cur >= 0);
            RT_STAT(2, cur >= 0 && cur < top_bytes);
            if (cur >= 0) {
                f4 q0, q1, q2;
                int lref, rref;
                load_pair(pairs, top, top_bytes, cur, q0, q1, q2, lref, rref);
                float ld, rd;
                bool lh, rh;
                if (!wave_exact) slab_pair<false>(q0, q1, q2, rs, ld, lh, rd, rh);
                else slab_pair<true>(q0, q1, q2, rs, ld, lh, rd, rh);
                RT_PROBE_STEP(ld);
                // inside a mesh the reference culls by the mesh's own closest hit only (t_local).  Boxes entered beyond
                // the closest hit over ALL instances hold no triangle that can be accepted (t <= t_best, Object.h:312):
                // they are skipped too, with occlusion_queue's margin against the rounding of the box test — same hits.
                const float cull = ANY ? tcull : (in_inst >= 0 ? fminf(t_local, t_best * 1.0001f + 0.000001f) : t_best);
                lh = lh && ld < cull;
                rh = rh && rd < cull;
                const bool left_first = ld < rd;
                cur = (lh && (!rh || left_first)) ? lref : rref;
                if (lh && rh) push(left_first ? rref : lref);
                else if (!lh && !rh) cur = pop();
            }
        }

        const unsigned long long at_leaf = __ballot(cur < 0 && cur != RT_CUR_DONE);
        if (keep_walking(at_leaf, __ballot(cur >= 0), sc.leaf_min)) continue;
        RT_STAT(1, cur < 0 && cur != RT_CUR_DONE);
        RT_STAT2(0, cur == RT_CUR_EXIT);
        RT_STAT2(1, cur < 0 && cur > RT_CUR_EXIT && in_inst >= 0);
        RT_STAT2(2, cur < 0 && cur > RT_CUR_EXIT && in_inst < 0);
        RT_STAT2(3, cur == RT_CUR_FINISH);
        if (cur < 0 && cur != RT_CUR_DONE) {
            if (cur == RT_CUR_EXIT) { // the mesh is finished: the world ray comes back (Object.h:384 returns)
                const float4 o4 = q_o[q], d4 = q_d[q];
                o = V3(o4);
                d = V3(d4);
                rs = ray_slab(o4, d4, &exact);
                in_inst = -1;
                cur = pop();
            } else if (cur != RT_CUR_FINISH && in_inst >= 0) { // a leaf of the mesh's BVH (Object.h:292-320)
                int slot = ~cur;
                bool last, stop = false;
                do {
                    float4 a, b, c;
                    load_tri(tris, slot, a, b, c);
                    last = (__float_as_uint(a.w) & RT_TRI_LEAF_END) != 0u;
                    float tt, u, v;
                    if (tri_accept(o, d, a, b, c, t_best, &tt, &u, &v)) {
                        t_best = tt;
                        t_local = tt;
                        found = true;
                        hit_store(u, v, slot, in_inst | 0x40000000, V3(0.f));
                        if (ANY) stop = last = true;
                    }
                    slot++;
                } while (!last);
                cur = stop ? RT_CUR_HIT : pop();
            } else if (cur != RT_CUR_FINISH) { // a TLAS leaf: one Instance (Scene.cpp:33-44, Object.h:380-394)
                const int k = ~cur;
                const float4* iq = sc.inst_q + (size_t)k * RT_INST_QUADS;
                const Xf w2m = load_xf(iq);
                const float4 meta = iq[3];
                const int geometry = __float_as_int(meta.x);
                // the lane is in the TLAS, so (o, d) is the world ray: Transform::to_model_ray, Math.h:692-694
                const V3 mo = xf_point(w2m, o), md = xf_model_vector(w2m, d);
                if (geometry == RT_GEO_MESH) {
                    push(RT_CUR_EXIT);
                    in_inst = k;
                    o = mo;
                    d = md;
                    rs = ray_slab(pack(mo, 0.f), pack(md, 0.f), &exact);
                    t_local = INFINITY;
                    cur = __float_as_int(meta.z); // (a one-leaf mesh: its root is never box-tested, Object.h:276-290)
                } else {
                    const float4* pq = sc.inst_p + (size_t)k * 3;
                    const float4 p0 = pq[0], p1 = pq[1], p2 = pq[2];
                    const float p[12] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w, p2.x, p2.y, p2.z, p2.w};
                    // (the hit's record is stored where a geometry's test ends: nothing but t is carried out of the switch)
                    float t_prim;
                    bool stop = false;
                    if (prim_intersect(geometry, p, mo, md, 0.f, t_best, &t_prim,
                                       [&](float u, float v, const V3& n, bool front) { hit_store(u, v, -2, k | (front ? 0x40000000 : 0), n); })) {
                        t_best = t_prim;
                        found = true;
                        stop = ANY;
                    }
                    cur = stop ? RT_CUR_HIT : pop();
                }
            }
            if (cur == RT_CUR_FINISH || cur == RT_CUR_HIT) {
                Hit2 h;
                h.slot = h.inst = -1;
                h.t = INFINITY;
                h.u = h.v = 0.f;
                h.n = V3(0.f);
                h.front = true;
                if (!ANY && found) {
                    h.t = t_best;
                    h.u = __int_as_float(hrec[0 * RT_TRAV_BLOCK]);
                    h.v = __int_as_float(hrec[1 * RT_TRAV_BLOCK]);
                    h.slot = hrec[2 * RT_TRAV_BLOCK];
                    const int inst_front = hrec[3 * RT_TRAV_BLOCK];
                    h.inst = inst_front & 0x3fffffff;
                    h.front = (inst_front & 0x40000000) != 0;
                    h.n = V3(__int_as_float(hrec[4 * RT_TRAV_BLOCK]), __int_as_float(hrec[5 * RT_TRAV_BLOCK]), __int_as_float(hrec[6 * RT_TRAV_BLOCK]));
                }
                sink(q, h, ANY ? cur == RT_CUR_HIT : found);
                cur = RT_CUR_DONE;
                exact = false;
            }
            wave_exact = __ballot(exact && cur != RT_CUR_DONE) != 0ull;
        } else {
            wave_exact = __ballot(exact && cur != RT_CUR_DONE) != 0ull; // (lanes of this wave may have changed their ray above)
        }
    }    RT_STAT_FLUSH(ANY ? 8 : 0);
    RT_STAT2_FLUSH(ANY ? 8 : 0);
}

// ---------------------------------------------------------------- shading
RT_DEV void onb(const V3& N, V3& T, V3& B) { // Utils.h:53-59
    V3 up(0.f, 0.f, 1.f);
    if (fabsf(N.z) > 0.999f) up = V3(1.f, 0.f, 0.f);
    T = normalize(cross(up, N));
    B = cross(N, T);
}
RT_DEV V3 reflect(const V3& v, const V3& n) { return v - (2.f * dot(v, n)) * n; } // Utils.h:85-88
RT_DEV V3 refract(const V3& uv, const V3& n, float eta) {                         // Utils.h:89-95
    float theta = fmin_sel(dot(-uv, n), 1.0f);
    V3 perp = eta * (uv + theta * n);
    V3 par = (-sqrtf(fabsf(1.0f - len2(perp)))) * n;
    return perp + par;
}
// pow(x, 5) of the reference (Material.cpp:39,116; Utils.h:100) as a multiply chain (<= 2 ulp from libm's pow)
RT_DEV float pow5(float x) {
    const float x2 = x * x;
    return (x2 * x2) * x;
}
RT_DEV float schlick_reflectance(float cosine, float ref_idx) { // Utils.h:96-101
    float r0 = (1.f - ref_idx) / (1.f + ref_idx);
    r0 = r0 * r0;
    return r0 + (1.f - r0) * pow5(1.f - cosine);
}
RT_DEV float power_heuristic(float a, float b) { // Utils.h:130-136
    float denom = a * a + b * b;
    if (fabsf(denom) < 0.0001f) return 0.f;
    return (a * a) / denom;
}
RT_DEV V3 cosine_sample(Rng& r) { // Utils.h:70-83
    float r1 = r.next();
    float r2 = r.next();
    float phi = 2.f * kPi * r1;
    float ct = sqrtf(1.f - r2);
    float st = sqrtf(r2);
    float sp, cp;
    sincosf(phi, &sp, &cp);
    return V3(cp * st, sp * st, ct);
}

struct Surf { // Intersection, Def.h:11-40
    V3 point, normal;
    bool front_face;
    float u, v; // texture coordinates: a Rectangle's si->u, si->v (Object.h:116-117); 0 where the reference's Geometry writes none
};
struct Scatter {
    V3 origin, dir, f;
    float pdf;
};

// ---- GGX / Beckmann Microfacet (Material.cpp:4-109; Beckmann D is new)
RT_DEV float ggx_D(const V3& N, const V3& H, float alpha) {
    float ndh = fmax_sel(dot(H, N), 0.f);
    if (ndh <= 0.f) return 0.f;
    float a2 = alpha * alpha;
    float n2 = ndh * ndh;
    // pow(tan(acos(NdotH)), 2) (Material.cpp:8-10) = (1 - c^2) / c^2, without the two transcendentals
    float inner = a2 + (1.f - n2) / n2;
    float denom = kPi * (n2 * n2) * (inner * inner);
    if (denom <= 0.f) return 0.f;
    return a2 / denom;
}
RT_DEV float beckmann_D(const V3& N, const V3& H, float alpha) {
    float ndh = fmax_sel(dot(H, N), 0.f);
    if (ndh <= 0.f) return 0.f;
    float c2 = ndh * ndh;
    float a2 = alpha * alpha;
    float t2 = (1.f - c2) / c2;
    float denom = kPi * a2 * c2 * c2;
    if (denom <= 0.f) return 0.f;
    return expf(-t2 / a2) / denom;
}
RT_DEV V3 microfacet_sample_h(const V3& N, float alpha, bool beckmann, Rng& r) { // Material.cpp:14-30
    float r1 = r.next();
    float r2 = r.next();
    // theta_m = atan(x) (Material.cpp:18); only its sine and cosine are used: cos = 1/sqrt(1+x^2), sin = x cos
    const float x = beckmann ? sqrtf(-(alpha * alpha) * logf(1.f - r1)) : (alpha * sqrtf(r1)) / sqrtf(1.f - r1);
    float phi = 2.f * kPi * r2;
    const float ct = 1.f / sqrtf(1.f + x * x);
    const float st = x * ct;
    float sp, cp;
    sincosf(phi, &sp, &cp);
    V3 m(cp * st, sp * st, ct);
    V3 T, B;
    onb(N, T, B);
    return T * m.x + B * m.y + N * m.z;
}
RT_DEV float microfacet_pdf(const V3& V, const V3& H, const V3& N, float alpha, bool beckmann) { // Material.cpp:31-34
    float D = beckmann ? beckmann_D(N, H, alpha) : ggx_D(N, H, alpha);
    return D * fmax_sel(dot(H, N), 0.f) / (4.f * fmax_sel(dot(V, H), 0.f) + 0.0000001f);
}
RT_DEV V3 fresnel_schlick(float vdh, const V3& F0) { // Material.cpp:37-40
    float p = pow5(clampf(1.f - vdh, 0.f, 1.f));
    return F0 + (V3(1.f) - F0) * p;
}
RT_DEV float schlick_ggx_G1(float ndv, float rough) { // Material.cpp:42-48
    float k = rough * rough * 0.5f;
    float denom = ndv * (1.f - k) + k;
    if (denom == 0.f) return 0.f;
    return ndv / denom;
}
RT_DEV V3 microfacet_eval(const V3& n, const V3& V, const V3& H, const V3& L, float rough, bool beckmann) { // Material.cpp:57-66
    float alpha = rough * rough;
    float D = beckmann ? beckmann_D(n, H, alpha) : ggx_D(n, H, alpha);
    V3 F = fresnel_schlick(fmax_sel(dot(V, H), 0.f), V3(0.8f));
    float G = schlick_ggx_G1(fmax_sel(dot(n, V), 0.f), rough) * schlick_ggx_G1(fmax_sel(dot(n, L), 0.f), rough);
    float denom = 4.0f * fmax_sel(dot(n, V), 0.f) * fmax_sel(dot(n, L), 0.f) + 0.0000001f;
    return (fmax_sel(dot(n, L), 0.f) * ((D * F) * G)) / denom;
}

// ---- Disney lobes (Material.cpp:114-722)
RT_DEV float diffuse_fresnel(float ndw, float f90) { return 1.f + (f90 - 1.f) * pow5(1.f - ndw); }

RT_DEV V3 disney_diffuse_core(const V3& albedo, float rough, float subsurface, float hdo, float ndo, float ndi) {
    float fd90 = 0.5f + 2.f * rough * hdo * hdo;
    V3 base = ((albedo / kPi) * diffuse_fresnel(ndi, fd90)) * diffuse_fresnel(ndo, fd90) * ndo;
    float fss90 = rough * hdo * hdo;
    float ss = diffuse_fresnel(ndi, fss90) * diffuse_fresnel(ndo, fss90) * (1.f / (ndi + ndo) - 0.5f) + 0.5f;
    V3 sub = ((1.25f * albedo) / kPi) * ss * ndo;
    return (1.f - subsurface) * base + subsurface * sub;
}
RT_DEV float tr_D(const V3& H, float ax, float ay) { // Material.cpp:160-164
    float hx = H.x / ax, hy = H.y / ay;
    float inner = hx * hx + hy * hy + H.z * H.z;
    float denom = (kPi * ax * ay) * (inner * inner);
    return 1.f / denom;
}
RT_DEV float smith_G1(const V3& w, float ax, float ay) { // Material.cpp:166-171
    float inner = 1.f + (w.x * ax * w.x * ax + w.y * ay * w.y * ay) / (w.z * w.z);
    float denom = 1.f + (sqrtf(1.f + inner) - 1.f) / 2.f;
    return 1.f / denom;
}
RT_DEV V3 sample_vndf(const V3& Wi, float ax, float ay, Rng& r) { // Material.cpp:178-199
    float r1 = r.next();
    float r2 = r.next();
    V3 Vh = normalize(V3(ax * Wi.x, ay * Wi.y, Wi.z));
    float l2 = Wi.x * Wi.x + Wi.y * Wi.y;
    V3 T1 = l2 > 0.f ? V3(-Vh.y, Vh.x, 0.f) / sqrtf(l2) : V3(1.f, 0.f, 0.f);
    V3 T2 = cross(T1, Vh);
    float rr = sqrtf(r1);
    float phi = 2.0f * kPi * r2;
    float sp, cp;
    sincosf(phi, &sp, &cp);
    float t1 = rr * cp;
    float t2 = rr * sp;
    float s = 0.5f * (1.f + Vh.z);
    t2 = (1.f - s) * sqrtf(1.f - t1 * t1) + s * t2;
    V3 Nh = t1 * T1 + t2 * T2 + sqrtf(fmax_sel(1.f - t1 * t1 - t2 * t2, 0.f)) * Vh;
    return normalize(V3(ax * Nh.x, ay * Nh.y, fmax_sel(0.f, Nh.z)));
}
RT_DEV void aniso_alphas(float rough, float aniso, float& ax, float& ay) { // Material.cpp:216-218
    float aspect = sqrtf(1.0f - aniso * 0.9f);
    ax = fmax_sel(0.001f, rough * rough / aspect);
    ay = fmax_sel(0.001f, rough * rough * aspect);
}
RT_DEV V3 disney_metal_eval(const V3& V, const V3& L, const V3& H, float ax, float ay, const V3& base, float* pdf) { // Material.cpp:201-212
    V3 F = fresnel_schlick(fmax_sel(dot(V, H), 0.f), base);
    float D = tr_D(H, ax, ay);
    float g1 = smith_G1(V, ax, ay);
    float g2 = smith_G1(L, ax, ay);
    float G = g1 * g2;
    *pdf = D * g1 / (4.f * V.z);
    return ((F * D) * G) / (4.f * V.z);
}
RT_DEV float r0_from_ior(float ior) { return ((ior - 1.f) * (ior - 1.f)) / ((ior + 1.f) * (ior + 1.f)); } // Material.cpp:256-259
RT_DEV float clearcoat_D(float ag, float hdn) {                                                            // Material.cpp:260-265
    float ag2 = ag * ag;
    float denom = kPi * logf(ag2) * (1.f + (ag2 - 1.f) * hdn * hdn);
    return (ag2 - 1.f) / denom;
}
RT_DEV V3 clearcoat_sample(float ag, Rng& r) { // Material.cpp:267-280
    float ag2 = ag * ag;
    float r1 = r.next();
    float r2 = r.next();
    float phi = 2.f * kPi * r2;
    float ct = sqrtf((1.f - powf(ag2, 1.f - r1)) / (1.f - ag2));
    float st = clampf(sqrtf(1.f - (ct * ct)), 0.f, 1.f);
    float sp, cp;
    sincosf(phi, &sp, &cp);
    return V3(st * cp, st * sp, ct);
}
RT_DEV V3 clearcoat_eval(const V3& V, const V3& L, const V3& H, float gloss, float* pdf) { // Material.cpp:282-294
    float r0 = r0_from_ior(1.5f);
    float ag = (1.f - gloss) * 0.1f + gloss * 0.001f;
    float Fs = r0 + (1.f - r0) * pow5(1.f - fmax_sel(dot(H, L), 0.f));
    float D = clearcoat_D(ag, H.z);
    float G = smith_G1(V, 0.25f, 0.25f) * smith_G1(L, 0.25f, 0.25f);
    *pdf = D / (4.f * fmax_sel(dot(H, L), 0.f) + 0.00000001f);
    return ((V3(Fs) * D) * G) / (4.f * V.z);
}
RT_DEV float dielectric_fresnel3(float ndi, float ndt, float eta) { // Material.cpp:343-349
    float rs = (ndi - eta * ndt) / (ndi + eta * ndt);
    float rp = (ndi * eta - ndt) / (ndi * eta + ndt);
    return (rs * rs + rp * rp) / 2.f;
}
RT_DEV float dielectric_fresnel(float ndi, float eta) { // Material.cpp:350-357
    float ndt2 = 1.f - (1.f - ndi * ndi) / (eta * eta);
    if (ndt2 < 0.f) return 1.f;
    return dielectric_fresnel3(fabsf(ndi), sqrtf(ndt2), eta);
}
RT_DEV V3 disney_refraction_eval(const V3& V, const V3& L, const V3& H, float eta, float ax, float ay, const V3& base, float* pdf) { // Material.cpp:360-377
    float D = tr_D(H, ax, ay);
    float hdi = dot(H, V);
    float F = dielectric_fresnel(hdi, 1.f / eta);
    float g1 = smith_G1(V, ax, ay);
    float g2 = smith_G1(L, ax, ay);
    float G = g1 * g2;
    float hdo = fabsf(dot(H, L));
    float sd = hdi + eta * hdo;
    V3 out = (((((base * (1.f - F)) * D) * G) * eta) * eta) * fabsf(hdo * hdi) / (fabsf(V.z) * sd * sd);
    float dh = eta * eta * hdo / (sd * sd);
    *pdf = (1.f - F) * D * g1 * fabsf(dh * hdi / V.z);
    return out;
}
RT_DEV V3 disney_reflection_eval(const V3& V, const V3& L, const V3& H, float eta, float ax, float ay, const V3& base, float* pdf) { // Material.cpp:379-392
    float hdi = dot(H, V);
    float F = dielectric_fresnel(hdi, eta);
    float D = tr_D(H, ax, ay);
    float g1 = smith_G1(V, ax, ay);
    float g2 = smith_G1(L, ax, ay);
    float G = g1 * g2;
    *pdf = F * D * g1 / (4.f * V.z);
    return (((base * F) * D) * G) / (4.f * V.z);
}

struct UberP { // DisneyUber fields, Material.h:183-196
    V3 base;
    float spec_trans, metallic, subsurface, specular, rough, spec_tint, aniso, clearcoat, ior;
};
RT_DEV UberP uber_params(const rt_material& m) {
    UberP u;
    u.base = V3(m.c0);
    u.spec_trans = m.s[0]; u.metallic = m.s[1]; u.subsurface = m.s[2]; u.specular = m.s[3]; u.rough = m.s[4];
    u.spec_tint = m.s[5]; u.aniso = m.s[6]; u.clearcoat = m.s[9]; u.ior = m.s[11];
    return u;
}
RT_DEV V3 uber_diffuse(const UberP& u, const V3& V, const V3& L, const V3& H, float* pdf) { // Material.cpp:571-589
    *pdf = fmax_sel(L.z, 0.f) / kPi;
    V3 d = disney_diffuse_core(u.base, u.rough, u.subsurface, fmax_sel(dot(H, L), 0.f), fmax_sel(L.z, 0.f), fmax_sel(V.z, 0.f));
    return (d * (1.f - u.spec_trans)) * (1.f - u.metallic);
}
RT_DEV V3 uber_metal(const UberP& u, const V3& V, const V3& L, const V3& H, float eta, float ax, float ay, float* pdf) { // Material.cpp:591-609
    V3 c_tint = u.base / luminance(u.base);
    V3 ks = V3(1.f - u.spec_tint) + u.spec_tint * c_tint;
    V3 C0 = ((u.specular * r0_from_ior(eta)) * (1.f - u.metallic)) * ks + u.metallic * u.base;
    float p = pow5(1.f - fmax_sel(dot(L, H), 0.f));
    V3 F = C0 + (V3(1.f) - C0) * p;
    float D = tr_D(H, ax, ay);
    float g1 = smith_G1(V, ax, ay);
    float g2 = smith_G1(L, ax, ay);
    float G = g1 * g2;
    *pdf = D * g1 / (4.f * V.z);
    V3 f = ((F * D) * G) / (4.f * V.z);
    return f * (1.f - u.spec_trans * (1.f - u.metallic));
}

RT_DEV V3 mat_emitted(const rt_material& m) { return m.type == RT_MAT_EMISSIVE ? V3(m.c0) : V3(0.f); }

// albedo->sample(si.u, si.v, si.point) of MatteMaterial::Eval (Material.h:84): ConstantTexture (Material.h:13-24) or,
// when s0 = its grid (PI / the constructor's argument) is set, CheckeredTexture::sample (Material.h:29-36)
RT_DEV float modulo_f(float a, float b) { // Math.h:39-42
    const float r = fmodf(a, b);
    return (r < 0.f) ? r + b : r;
}
RT_DEV V3 matte_albedo(const rt_material& m, const Surf& si) {
    const float grid = m.s[0];
    if (grid == 0.f) return V3(m.c0);
    const bool odd = ((double)modulo_f(si.u, grid) < (double)grid * 0.5) ^ ((double)modulo_f(si.v, grid) < (double)grid * 0.5);
    // selected by VALUE: `odd ? V3(m.c1) : V3(m.c0)` becomes a select of two addresses, which keeps the whole record in
    // scratch memory — its type field with it, so the per-type kernels stop folding their switch (two-level shading
    // kernels carried every material's code: 246 VGPRs, two waves per SIMD)
    const float e0 = m.c0[0], e1 = m.c0[1], e2 = m.c0[2], o0 = m.c1[0], o1 = m.c1[1], o2 = m.c1[2];
    return V3(odd ? o0 : e0, odd ? o1 : e1, odd ? o2 : e2);
}

// Material::Eval
RT_DEV V3 mat_eval(const rt_material& m, const Surf& si, const V3& in_dir, const V3& out_dir, float* pdf) {
    const V3 n = si.normal;
    switch (m.type) {
    case RT_MAT_MATTE: { // Material.h:81-88
        float c = fmax_sel(dot(n, out_dir), 0.f);
        *pdf = c / kPi;
        return (matte_albedo(m, si) * c) / kPi;
    }
    case RT_MAT_MICROFACET_GGX:
    case RT_MAT_MICROFACET_BECKMANN: { // Material.cpp:87-98
        const bool beck = m.type == RT_MAT_MICROFACET_BECKMANN;
        V3 Wo = -in_dir;
        V3 Wh = normalize(Wo + out_dir);
        *pdf = microfacet_pdf(Wo, Wh, n, m.s[0] * m.s[0], beck);
        return microfacet_eval(n, Wo, Wh, out_dir, m.s[0], beck);
    }
    case RT_MAT_DISNEY_DIFFUSE: { // Material.cpp:128-153
        *pdf = fmax_sel(dot(n, out_dir), 0.f) / kPi;
        V3 Wh = normalize(-in_dir + out_dir);
        return disney_diffuse_core(V3(m.c0), m.s[0], m.s[1], fmax_sel(dot(Wh, out_dir), 0.f), fmax_sel(dot(n, out_dir), 0.f),
                                   fmax_sel(dot(n, -in_dir), 0.f));
    }
    case RT_MAT_DISNEY_SHEEN: { // Material.cpp:554-567
        *pdf = fmax_sel(dot(n, out_dir), 0.f) / kPi;
        V3 half = normalize(-in_dir + out_dir);
        V3 base(m.c0);
        float denom = luminance(base);
        if (denom < 0.f) denom = 1.f;
        V3 tint = base / denom;
        V3 c = V3(1.f - m.s[0]) + m.s[0] * tint;
        return (c * pow5(1.f - fabsf(dot(half, out_dir)))) * dot(n, out_dir);
    }
    case RT_MAT_EMISSIVE:
    case RT_MAT_GLASS:
        *pdf = 0.f;
        return V3(0.f);
    default: break;
    }
    // tangent-space lobes
    V3 T, B;
    onb(n, T, B);
    const V3 V(dot(-in_dir, T), dot(-in_dir, B), dot(-in_dir, n));
    const V3 L(dot(out_dir, T), dot(out_dir, B), dot(out_dir, n));
    switch (m.type) {
    case RT_MAT_DISNEY_METAL: { // Material.cpp:234-248
        float ax, ay;
        aniso_alphas(m.s[0], m.s[1], ax, ay);
        return disney_metal_eval(V, L, normalize(V + L), ax, ay, V3(m.c0), pdf);
    }
    case RT_MAT_DISNEY_CLEARCOAT: // Material.cpp:314-325
        return clearcoat_eval(V, L, normalize(V + L), m.s[0], pdf);
    case RT_MAT_DISNEY_GLASS: { // Material.cpp:432-456
        const bool refl = dot(n, out_dir) >= 0.f;
        const float eta = si.front_face ? 1.f / m.s[2] : m.s[2];
        float ax, ay;
        aniso_alphas(m.s[0], m.s[1], ax, ay);
        if (refl) return disney_reflection_eval(V, L, normalize(V + L), eta, ax, ay, V3(m.c0), pdf);
        return disney_refraction_eval(V, L, normalize(V + L * eta), eta, ax, ay, V3(m.c0), pdf);
    }
    case RT_MAT_ROUGH_DIELECTRIC: { // Material.cpp:494-541
        const bool refl = dot(n, out_dir) <= 0.f; // as written in the reference
        const float eta = si.front_face ? 1.f / m.s[0] : m.s[0];
        V3 hv = refl ? normalize(-in_dir + out_dir) : normalize(-in_dir + out_dir * eta);
        const V3 lh(dot(hv, T), dot(hv, B), dot(hv, n));
        if (hv.z < 0.f) hv = -hv; // dot(half_vec, vec3(0,0,1)) < 0
        const float hdi = dot(hv, -in_dir);
        const float F = dielectric_fresnel(hdi, eta);
        const float alpha = m.s[1] * m.s[1];
        const float D = tr_D(lh, alpha, alpha);
        const float g1 = smith_G1(V, alpha, alpha);
        const float g2 = smith_G1(L, alpha, alpha);
        const float G = g1 * g2;
        if (refl) {
            *pdf = (F * D * g1) / (4.f * fabsf(V.z));
            return (V3(m.c0) * (F * D * G)) / (4.f * fabsf(V.z));
        }
        const float hdo = fabsf(dot(hv, out_dir));
        const float sd = hdi + eta * hdo;
        V3 out = (((((V3(m.c1) * (1.f - F)) * D) * G) * eta) * eta) * fabsf(hdo * hdi) / (fabsf(V.z) * sd * sd);
        const float dh = eta * eta * hdo / (sd * sd);
        *pdf = (1.f - F) * D * g1 * fabsf(dh * hdi / V.z);
        return out;
    }
    case RT_MAT_DISNEY_UBER: { // Material.cpp:678-722
        const UberP u = uber_params(m);
        const V3 H = normalize(V + L);
        float dw = (1.f - u.metallic) * (1.f - u.spec_trans);
        float mw = (1.f - u.spec_trans * (1.f - u.metallic));
        float gw = (1.f - u.metallic) * u.spec_trans;
        float cw = 0.25f * u.clearcoat;
        if (!si.front_face) dw = mw = cw = 0.f;
        const float sum = dw + mw + gw + cw;
        dw /= sum;
        mw /= sum;
        float ax, ay;
        aniso_alphas(u.rough, u.aniso, ax, ay);
        const float eta = si.front_face ? 1.f / u.ior : u.ior;
        float p, total = 0.f;
        V3 f(0.f);
        if (dw > 0.f) {
            f = f + uber_diffuse(u, V, L, H, &p);
            total += p * dw;
        }
        if (mw > 0.f) {
            f = f + uber_metal(u, V, L, H, eta, ax, ay, &p);
            total += p * mw;
        }
        *pdf = total;
        return f;
    }
    default:
        *pdf = 0.f;
        return V3(0.f);
    }
}

// Material::Sample_Eval
RT_DEV Scatter mat_sample(const rt_material& m, const Surf& si, const V3& in_dir, Rng& r) {
    Scatter sc;
    sc.origin = V3(0.f);
    sc.dir = V3(1.f, 0.f, 0.f); // Ray(), Math.h:624
    sc.f = V3(0.f);
    sc.pdf = 0.f;
    const V3 n = si.normal;
    switch (m.type) {
    case RT_MAT_MATTE:          // Material.h:71-79
    case RT_MAT_DISNEY_DIFFUSE: // Material.cpp:119-127
    case RT_MAT_DISNEY_SHEEN: { // Material.cpp:545-553
        V3 d = cosine_sample(r);
        V3 T, B;
        onb(n, T, B);
        d = d.x * T + d.y * B + d.z * n;
        sc.origin = si.point + n * 0.001f;
        sc.dir = d;
        sc.f = mat_eval(m, si, in_dir, d, &sc.pdf);
        return sc;
    }
    case RT_MAT_MICROFACET_GGX:
    case RT_MAT_MICROFACET_BECKMANN: { // Material.cpp:68-86
        const bool beck = m.type == RT_MAT_MICROFACET_BECKMANN;
        const float alpha = m.s[0] * m.s[0];
        V3 Wo = -in_dir;
        V3 Wh = microfacet_sample_h(n, alpha, beck, r);
        V3 Wi = -reflect(-in_dir, Wh);
        sc.origin = si.point + n * 0.0001f;
        sc.dir = Wi;
        sc.pdf = microfacet_pdf(Wo, Wh, n, alpha, beck);
        if (sc.pdf != sc.pdf) sc.pdf = 0.f;
        sc.f = microfacet_eval(n, Wo, Wh, Wi, m.s[0], beck);
        return sc;
    }
    case RT_MAT_GLASS: { // Material.h:248-266
        const float ratio = si.front_face ? (1.0f / m.s[0]) : m.s[0];
        const float ct = fmin_sel(dot(-in_dir, n), 1.0f);
        const float st = sqrtf(1.f - ct * ct);
        bool refl = ratio * st > 1.f;
        if (!refl) refl = schlick_reflectance(ct, ratio) > r.next(); // the draw is skipped on total internal reflection
        V3 d = refl ? reflect(in_dir, n) : refract(in_dir, n, ratio);
        sc.origin = si.point + n * (refl ? 0.001f : -0.001f);
        sc.dir = normalize(d);
        sc.pdf = 1.f;
        sc.f = V3(1.f);
        return sc;
    }
    case RT_MAT_EMISSIVE: return sc; // Material.h:56-59
    default: break;
    }
    V3 T, B;
    onb(n, T, B);
    const V3 V(dot(-in_dir, T), dot(-in_dir, B), dot(-in_dir, n));
    switch (m.type) {
    case RT_MAT_DISNEY_METAL: { // Material.cpp:214-233
        float ax, ay;
        aniso_alphas(m.s[0], m.s[1], ax, ay);
        V3 H = sample_vndf(V, ax, ay, r);
        V3 L = reflect(-V, H);
        sc.f = disney_metal_eval(V, L, H, ax, ay, V3(m.c0), &sc.pdf);
        sc.dir = L.x * T + L.y * B + L.z * n;
        sc.origin = si.point + n * 0.00001f;
        return sc;
    }
    case RT_MAT_DISNEY_CLEARCOAT: { // Material.cpp:297-313
        float ag = (1.f - m.s[0]) * 0.1f + m.s[0] * 0.001f;
        V3 H = clearcoat_sample(ag, r);
        V3 L = reflect(-V, H);
        sc.f = clearcoat_eval(V, L, H, m.s[0], &sc.pdf);
        sc.dir = L.x * T + L.y * B + L.z * n;
        sc.origin = si.point + n * 0.00001f;
        return sc;
    }
    case RT_MAT_DISNEY_GLASS: { // Material.cpp:396-431
        const float eta = si.front_face ? (1.f / m.s[2]) : m.s[2];
        float ax, ay;
        aniso_alphas(m.s[0], m.s[1], ax, ay);
        V3 H = sample_vndf(V, ax, ay, r);
        const float F = dielectric_fresnel(dot(H, V), 1.f / eta);
        const bool do_reflect = F > r.next();
        if (do_reflect) {
            V3 L = normalize(reflect(-V, H));
            sc.dir = L.x * T + L.y * B + L.z * n;
            sc.origin = si.point + n * 0.00001f;
            sc.f = disney_reflection_eval(V, L, H, eta, ax, ay, V3(m.c0), &sc.pdf);
        } else {
            V3 L = normalize(refract(-V, H, eta));
            sc.dir = L.x * T + L.y * B + L.z * n;
            sc.origin = si.point - n * 0.00001f;
            sc.f = disney_refraction_eval(V, L, H, eta, ax, ay, V3(m.c0), &sc.pdf);
        }
        return sc;
    }
    case RT_MAT_ROUGH_DIELECTRIC: { // Material.cpp:463-493
        const float eta = si.front_face ? 1.f / m.s[0] : m.s[0];
        const float alpha = m.s[1] * m.s[1];
        V3 lm = sample_vndf(V, alpha, alpha, r);
        V3 wh = lm.x * T + lm.y * B + lm.z * n;
        const float F = dielectric_fresnel(dot(wh, -in_dir), eta);
        const float r1 = r.next();
        if (r1 <= F) {
            sc.dir = normalize(reflect(in_dir, wh));
            sc.origin = si.point + n * 0.00001f;
        } else {
            sc.dir = normalize(refract(in_dir, wh, eta));
            sc.origin = si.point + n * -0.00001f;
        }
        sc.f = mat_eval(m, si, in_dir, sc.dir, &sc.pdf);
        return sc;
    }
    case RT_MAT_DISNEY_UBER: { // Material.cpp:611-677
        const UberP u = uber_params(m);
        float dw = (1.f - u.metallic) * (1.f - u.spec_trans);
        float mw = (1.f - u.spec_trans * (1.f - u.metallic));
        float gw = (1.f - u.metallic) * u.spec_trans;
        float cw = 0.25f * u.clearcoat;
        if (!si.front_face) dw = mw = cw = 0.f;
        const float c0 = dw, c1 = c0 + mw, c2 = c1 + gw, c3 = c2 + cw;
        const float sum = c3;
        const float lobe = r.next() * sum;
        float ax, ay;
        aniso_alphas(u.rough, u.aniso, ax, ay);
        const float eta = si.front_face ? 1.f / u.ior : u.ior;
        if (lobe < c0) {
            V3 L = cosine_sample(r);
            sc.dir = L.x * T + L.y * B + L.z * n;
            sc.origin = si.point + n * 0.001f;
            sc.f = uber_diffuse(u, V, L, normalize(V + L), &sc.pdf);
            sc.pdf = sc.pdf * (dw / sum);
        } else if (lobe < c1) {
            V3 H = sample_vndf(V, ax, ay, r);
            V3 L = reflect(-V, H);
            sc.dir = L.x * T + L.y * B + L.z * n;
            sc.origin = si.point + n * 0.001f;
            sc.f = uber_metal(u, V, L, H, eta, ax, ay, &sc.pdf);
            sc.pdf *= (mw / sum);
        } // glass / clearcoat lobes are unwritten in the reference: f = 0, pdf = 0 (Material.cpp:665-673)
        return sc;
    }
    default: return sc;
    }
}

// ------------------------------------------------------ environment light
RT_DEV int imod(int a, int b) { // Math.h:44-47
    int r = a % b;
    return (r < 0) ? r + b : r;
}
// The three sampling tables of the shading kernels (environment CDF, environment texels, emissive-triangle CDF), read
// either from global memory or from the workgroup's LDS copy.  Two types, not one generic pointer: a generic pointer
// turns every step of the CDF searches into a flat load that queues behind the kernel's divergent global traffic.
typedef const __attribute__((address_space(3))) float lds_cfloat;
struct GlobalTables {
    const float* cdf;
    const float* rgb;
    const rt_emissive_tri* em;
    RT_DEV float env_cdf(int i) const { return cdf[i]; }
    RT_DEV V3 texel(int i) const { return V3(rgb + 3 * (size_t)i); }
    RT_DEV float em_cdf(int i) const { return em[i].cdf; }
    RT_DEV int em_slot(int i) const { return em[i].slot; }
};
struct LdsTables {
    lds_cfloat* cdf;
    lds_cfloat* rgb;
    lds_cfloat* em; // rt_emissive_tri as two words: slot bits, cdf
    RT_DEV float env_cdf(int i) const { return cdf[i]; }
    RT_DEV V3 texel(int i) const { return V3(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]); }
    RT_DEV float em_cdf(int i) const { return em[2 * i + 1]; }
    RT_DEV int em_slot(int i) const { return __float_as_int(em[2 * i]); }
};
template <class TB>
RT_DEV V3 env_texel(const SceneView& sc, const TB& tb, int x, int y) { return tb.texel(sc.env_w * y + x); }
template <class TB>
RT_DEV V3 env_lookup(const SceneView& sc, const TB& tb, float u, float v) { // Image.h:46-64
    const int w = sc.env_w, h = sc.env_h;
    u = u * w - 0.5f;
    v = v * h - 0.5f;
    int ufi = imod((int)u, w);
    int vfi = imod((int)v, h);
    int uci = imod(ufi + 1, w);
    int vci = imod(vfi + 1, h);
    float ip;
    float uo = modff(u - ufi, &ip);
    float vo = modff(v - vfi, &ip);
    V3 ff = env_texel(sc, tb, ufi, vfi), fc = env_texel(sc, tb, ufi, vci), cf = env_texel(sc, tb, uci, vfi), cc = env_texel(sc, tb, uci, vci);
    return (ff * (1.f - uo)) * (1.f - vo) + (fc * (1.f - uo)) * (vo) + (cf * (uo)) * (1.f - vo) + (cc * (uo)) * (vo);
}
RT_DEV float env_pdf_of(const SceneView& sc, const V3& color, float st) { // st = sin(theta)
    float pdf = luminance(color) / sc.env_total;
    if (st == 0.f) return 0.f;
    return pdf * ((float)(sc.env_w * sc.env_h) / (2.f * kPi * kPi * st));
}
template <class TB>
RT_DEV V3 env_eval(const SceneView& sc, const TB& tb, const V3& dir, float* pdf) { // main.cpp:144-164
    float theta = acosf(dir.y);
    float phi = atan2f(dir.z, dir.x);
    float u = (kPi + phi) / kPi * 0.5f + sc.env_offset;
    float v = theta / kPi;
    V3 color = env_lookup(sc, tb, u, v);
    *pdf = env_pdf_of(sc, color, sinf(theta));
    return color;
}
template <class TB>
RT_DEV V3 env_sample(const SceneView& sc, const TB& tb, float rnd, V3* dir, float* pdf) { // main.cpp:121-143,165-184
    const int w = sc.env_w, h = sc.env_h;
    int lo = 0, hi = h - 1;
    while (lo < hi) {
        int mid = (lo + hi) / 2;
        if (rnd < tb.env_cdf((mid + 1) * w - 1)) hi = mid;
        else lo = mid + 1;
    }
    const int y = lo;
    lo = 0;
    hi = w - 1;
    while (lo < hi) {
        int mid = (lo + hi) / 2;
        if (rnd < tb.env_cdf(y * w + mid)) hi = mid;
        else lo = mid + 1;
    }
    const int x = lo;
    float ux = (float)x / w, uy = (float)y / h;
    V3 color = env_lookup(sc, tb, ux, uy);
    ux -= sc.env_offset;
    float phi = ux * 2.f * kPi;
    float theta = uy * kPi;
    float st, ct, sp, cp;
    sincosf(theta, &st, &ct);
    sincosf(phi, &sp, &cp);
    *dir = V3(-st * cp, ct, -st * sp);
    *pdf = env_pdf_of(sc, color, st);
    return color;
}

// Instance::sample_geometry (Object.h:396-403) for a RECT light, area-CDF
// triangle pick for a TRIMESH light (extension).
template <class TB>
RT_DEV void light_sample(const SceneView& sc, const TB& tb, const rt_light& L, Rng& r, V3* point, V3* normal) {
    if (L.kind == RT_LIGHT_RECT) {
        float r1 = -1.f + 2.f * r.next(); // random_float(-1,1), Utils.h:12-15
        float r2 = -1.f + 2.f * r.next();
        V3 p = (V3(L.t) * L.hu) * r1 + (V3(L.b) * L.hv) * r2;
        const float* m = L.m;
        *point = V3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12] * 1.f, m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13] * 1.f,
                    m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14] * 1.f);
        const float* im = L.inv_m;
        const V3 N(L.n);
        *normal = V3(im[0] * N.x + im[1] * N.y + im[2] * N.z, im[4] * N.x + im[5] * N.y + im[6] * N.z,
                     im[8] * N.x + im[9] * N.y + im[10] * N.z);
        return;
    }
    float r0 = r.next();
    float b1 = r.next();
    float b2 = r.next();
    int lo = 0, hi = L.count - 1;
#ifdef RT_PROBE_NOSEARCH // timing probe: no CDF search (wrong distribution, same memory traffic otherwise)
    lo = (int)(r0 * (float)L.count);
    lo = lo > hi ? hi : lo;
#else
    while (lo < hi) {
        int mid = (lo + hi) / 2;
        if (r0 < tb.em_cdf(L.first + mid)) hi = mid;
        else lo = mid + 1;
    }
#endif
    const int slot = tb.em_slot(L.first + lo);
    const float4* tp = sc.tris + 3 * (size_t)slot;
    const float4* np = sc.normals + 3 * (size_t)slot;
    float su = sqrtf(b1);
    float u = 1.f - su;
    float v = b2 * su;
#ifdef RT_PROBE_NOGATHER // timing probe: no triangle / normal fetch for the sampled emitter
    *point = V3(0.1f * u, 2.f + 0.1f * v, 0.f);
    *normal = V3(0.f, -1.f, 0.f);
    (void)tp;
    (void)np;
#else
    *point = V3(tp[0]) + u * V3(tp[1]) + v * V3(tp[2]);
    V3 N = (1.f - u - v) * V3(np[0]) + u * V3(np[1]) + v * V3(np[2]);
    *normal = normalize(N);
#endif
}

// hammersley_2d + Camera::get_ray + Filter::sample (Utils.h:138-151, Scene.h:61-75, Filter.h)
RT_DEV void camera_ray(const rt_camera& cam, int x, int y, uint32_t i, uint32_t n, V3* o, V3* d) {
    float hx = (float)i / (float)n;
    float hy = (float)__brev(i) * 2.3283064365386963e-10f;
    float rx = hx + (float)x;
    float ry = hy + (float)y;
    float fx = floorf(rx), fy = floorf(ry);
    float dx = rx - fx;
    float dy = ry - fy;
    float ox, oy;
    if (cam.filter == RT_FILTER_GAUSSIAN) {
        float rad = cam.filter_param * sqrtf(-2.f * logf(fmax_sel(dx, 1e-8f)));
        float sp, cp;
        sincosf(2.f * kPi * dy, &sp, &cp);
        ox = rad * cp;
        oy = rad * sp;
    } else {
        ox = (2.f * dx - 1.f) * (cam.filter_param * 0.5f);
        oy = (2.f * dy - 1.f) * (cam.filter_param * 0.5f);
    }
    float px = (fx + 0.5f) + ox;
    float py = (fy + 0.5f) + oy;
    const float* m = cam.raster_to_world;
    V3 pos(m[0] * px + m[4] * py + m[8] * 0.f + m[12] * 1.f, m[1] * px + m[5] * py + m[9] * 0.f + m[13] * 1.f,
           m[2] * px + m[6] * py + m[10] * 0.f + m[14] * 1.f);
    *o = V3(cam.pos);
    *d = normalize(pos - *o);
}

} // namespace rtd
