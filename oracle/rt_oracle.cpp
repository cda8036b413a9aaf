// rt_oracle.cpp — CPU restatement of the reference's per-sample hot path.
//
// TEST INFRASTRUCTURE ONLY (see rt_oracle.h).  PARITY UNPINNED against the
// compiled reference: the reference cannot be built in this image and ships
// no golden vectors; every function below cites the reference lines it
// restates so the arithmetic can be audited by reading.
//
// Arithmetic conventions restated from the reference:
//   * fp32 everywhere, compiled with -ffp-contract=off;
//   * min/max are the select forms of Math.h:17-24 (NaN-asymmetric);
//   * vec3 / float is a reciprocal multiply (Math.h:193-196), float / vec3 and
//     vec3 / vec3 are true divisions (Math.h:185-188,197-200);
//   * unqualified abs() is the float overload (what MSVC gives the author and
//     what `using std::abs` gives under g++);
//   * transcendentals (log, sin, cos, tan, atan, acos, atan2, pow) are evaluated
//     by the double libm function and narrowed, as g++ resolves the reference's
//     unqualified calls; MSVC's float overloads differ by ULPs.
//   * rand() (Utils.h:7-10) is replaced by a counter-based stream keyed by
//     (pixel, sample, seed) — the one specified deviation, shared with the GPU
//     path so that paths can be compared one to one.
#include "rt_oracle.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace {

const float kPi = 3.1415926536; // Math.h:9

inline float fmax_sel(float a, float b) { return (a > b) ? a : b; } // Math.h:17-20
inline float fmin_sel(float a, float b) { return (a < b) ? a : b; } // Math.h:21-24
inline float clampf(float x, float lo, float hi) { return (x < lo) ? lo : ((x > hi) ? hi : x); } // Utils.h:16-19

struct V3 {
    float x, y, z;
    V3() : x(0), y(0), z(0) {}
    V3(float a, float b, float c) : x(a), y(b), z(c) {}
    explicit V3(float e) : x(e), y(e), z(e) {}
    explicit V3(const float* p) : x(p[0]), y(p[1]), z(p[2]) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    void store(float* p) const { p[0] = x; p[1] = y; p[2] = z; }
};
inline V3 operator-(const V3& a) { return V3(-a.x, -a.y, -a.z); }
inline V3 operator-(const V3& a, const V3& b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator+(const V3& a, const V3& b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator*(const V3& a, const V3& b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator*(float t, const V3& v) { return V3(t * v.x, t * v.y, t * v.z); } // Math.h:189-192
inline V3 operator*(const V3& v, float t) { return V3(v.x * t, v.y * t, v.z * t); } // vec3 * vec3(t)
inline V3 operator/(const V3& u, float t) { return (1 / t) * u; }                    // Math.h:193-196
inline V3 rdiv(float t, const V3& u) { return V3(t / u.x, t / u.y, t / u.z); }       // Math.h:197-200
inline float dot(const V3& u, const V3& v) { return u.x * v.x + u.y * v.y + u.z * v.z; }
inline V3 cross(const V3& u, const V3& v) {
    return V3(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x);
}
inline float len2(const V3& v) { return v.x * v.x + v.y * v.y + v.z * v.z; }
inline float len(const V3& v) { return std::sqrt(len2(v)); }
inline V3 normalize(const V3& v) { return v / len(v); } // Math.h:211-214
inline V3 vmin(const V3& a, const V3& b) { return V3(fmin_sel(a.x, b.x), fmin_sel(a.y, b.y), fmin_sel(a.z, b.z)); }
inline V3 vmax(const V3& a, const V3& b) { return V3(fmax_sel(a.x, b.x), fmax_sel(a.y, b.y), fmax_sel(a.z, b.z)); }
inline float luminance(const V3& c) { return 0.212671f * c.x + 0.715160f * c.y + 0.072169f * c.z; } // Utils.h:189-192

// column-major 4x4 (Math.h:338-423)
struct M4 {
    float m[16];
};
inline M4 m4_mul(const M4& a, const M4& b) { // Math.h:392-422: r[c*4+r] = dot(row r of a, column c of b)
    M4 r;
    for (int c = 0; c < 4; c++)
        for (int rr = 0; rr < 4; rr++)
            r.m[c * 4 + rr] = a.m[rr] * b.m[c * 4] + a.m[4 + rr] * b.m[c * 4 + 1] + a.m[8 + rr] * b.m[c * 4 + 2] +
                              a.m[12 + rr] * b.m[c * 4 + 3];
    return r;
}
inline V3 m4_point(const float* m, const V3& p) { // Math.h:368-390 with w = 1, .xyz()
    float w = 1.0;
    return V3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12] * w, m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13] * w,
              m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14] * w);
}

struct Box {
    V3 lo, hi;
    Box() : lo(INFINITY, INFINITY, INFINITY), hi(-INFINITY, -INFINITY, -INFINITY) {} // Math.h:745
};
inline Box box_union(const Box& a, const Box& b) { // Math.h:788-793
    Box r;
    r.lo = vmin(a.lo, b.lo);
    r.hi = vmax(a.hi, b.hi);
    return r;
}
inline float box_area(const Box& b) { // Math.h:749-752
    V3 s = b.hi - b.lo;
    return 2.0 * (s.x * s.y + s.x * s.z + s.y * s.z);
}

// ---------------------------------------------------------------- BVH build
struct Builder { // BVH.cpp:5-56
    const float* b6;
    int max_leaf, strat;
    std::vector<V3> cent;
    std::vector<orc_node> nodes;
    std::vector<int> idx;

    Box prim(int i) const {
        Box b;
        b.lo = V3(b6 + 6 * (size_t)i);
        b.hi = V3(b6 + 6 * (size_t)i + 3);
        return b;
    }
    Box range_bounds(int s, int e) const { // BVH.cpp:24-29
        Box b;
        for (int i = s; i < e; i++) b = box_union(b, prim(idx[i]));
        return b;
    }
    float sah(int s, int e, int axis, float pos) const { // BVH.cpp:30-46
        Box l, r;
        int lc = 0, rc = 0;
        for (int i = s; i < e; i++) {
            if (cent[idx[i]][axis] < pos) {
                lc++;
                l = box_union(l, prim(idx[i]));
            } else {
                rc++;
                r = box_union(r, prim(idx[i]));
            }
        }
        float cost = lc * box_area(l) + rc * box_area(r);
        return cost > 0 ? cost : 1e30f;
    }
    void build(int s, int e, int node) { // BVH.cpp:58-128
        int n = e - s;
        Box nb = range_bounds(s, e);
        nb.lo.store(nodes[node].bmin);
        nb.hi.store(nodes[node].bmax);
        if (n <= max_leaf) {
            nodes[node].left_node = s;
            nodes[node].count = n;
            return;
        }
        int split = 0;
        if (strat == 0) { // BVH_MIDDLE, BVH.cpp:75-89
            V3 c = (nb.lo + nb.hi) / 2.f;
            V3 ext = nb.hi - nb.lo;
            int ax = 0; // Math.h:775-783
            if (ext[1] > ext[ax]) ax = 1;
            if (ext[2] > ext[ax]) ax = 2;
            float mid = c[ax];
            auto it = std::partition(idx.begin() + s, idx.begin() + e, [&](int i) { return cent[i][ax] < mid; });
            split = int(it - idx.begin());
            if (split == s || split == e) split = (s + e) / 2;
        } else { // BVH_SAH, BVH.cpp:90-118
            float mid = 0;
            int axis = -1; // reference leaves these uninitialised (UB) when no plane improves on 1e30
            float best = 1e30f;
            for (int a = 0; a < 3; a++) {
                float lo = nb.lo[a], hi = nb.hi[a];
                if (lo == hi) continue;
                float scale = (hi - lo) / 100.f;
                for (int i = 0; i < 100; i++) {
                    float pos = lo + i * scale;
                    float cost = sah(s, e, a, pos);
                    if (cost < best) {
                        mid = pos;
                        axis = a;
                        best = cost;
                    }
                }
            }
            if (axis < 0) {
                split = (s + e) / 2; // defined behaviour chosen for the UB case: median split
            } else {
                auto it = std::partition(idx.begin() + s, idx.begin() + e, [&](int i) { return cent[i][axis] < mid; });
                split = int(it - idx.begin());
                if (split == s || split == e) split = (s + e) / 2;
            }
        }
        nodes[node].count = -1;
        int l = (int)nodes.size();
        nodes.resize(nodes.size() + 2); // BVH.cpp:122-123: left, then right, adjacent
        nodes[node].left_node = l;
        build(s, split, l);
        build(split, e, l + 1);
    }
};

// ------------------------------------------------------------- intersection
inline bool slab(const float* bmin, const float* bmax, const V3& o, const V3& d, float& t_out) { // Math.h:755-770
    V3 inv = rdiv(1.f, d);
    V3 t0 = (V3(bmin) - o) * inv;
    V3 t1 = (V3(bmax) - o) * inv;
    V3 a = vmin(t0, t1);
    V3 b = vmax(t0, t1);
    float tnear = fmax_sel(fmax_sel(a.x, a.y), a.z);
    float tfar = fmin_sel(fmin_sel(b.x, b.y), b.z);
    if (tnear > tfar || tfar < 0) return false;
    t_out = tnear;
    return true;
}

inline void tri_test(const V3& o, const V3& d, const V3& v0, const V3& v1, const V3& v2, float& t_out, V3& bary) { // Object.h:245-264
    const V3 e1 = v1 - v0;
    const V3 e2 = v2 - v0;
    const V3 h = cross(d, e2);
    const float a = dot(e1, h);
    if (a > -0.0000001 && a < 0.0000001) return; // compared in double, as written
    const float f = 1 / a;
    const V3 s = o - v0;
    const float u = f * dot(s, h);
    if (u < 0 || u > 1) return;
    const V3 q = cross(s, e1);
    const float v = f * dot(d, q);
    if (v < 0 || u + v > 1) return;
    const float t = f * dot(e2, q);
    if (t > 0.0001f) {
        t_out = t;
        bary = V3(u, v, 1 - u - v);
    }
}

struct MeshHit {
    int tri = -1; // original triangle index
    float t = INFINITY;
    float u = 0, v = 0;
};

bool g_count_product_occlusion_walk = false; // orc_count_product_occlusion_walk(): visit statistics only

// TriangleMesh::intersect traversal, Object.h:273-344.  any = stop at the first
// accepted hit (boolean use of trace_scene, main.cpp:250,281).
// cull_tmax is NOT the reference: it restates the one place where the product's walks leave the reference's procedure
// (boxes entered beyond the end of the segment handed in are skipped: occlusion rays, and a mesh instance entered with
// the closest hit of other instances as tmax) and exists only so that tools/measure_visits.py can count
// the nodes THOSE walks visit.  Every parity check runs with cull_tmax = false.
inline bool mesh_traverse(const orc_mesh* s, const V3& o, const V3& d, float tmin, float tmax, bool any, MeshHit& out,
                          orc_visits* vis, bool cull_tmax = false) {
    int stack[64];
    stack[0] = 0;
    int sp = 1;
    float t = INFINITY;
    const float segment_end = (tmax > 0.f && tmax < INFINITY) ? tmax * 1.0001f + 0.000001f : (tmax <= 0.f ? tmax : INFINITY);
    int inner = 0, leaf = 0;
    while (sp > 0) {
        const orc_node* nd = &s->nodes[stack[--sp]];
        if (nd->count != -1) {
            for (int i = 0; i < nd->count; i++) {
                leaf++; // visit statistics: one unit per triangle tested (= per leaf with the reference's one-triangle leaves)
                V3 bary_tmp;
                float t_tmp = -1;
                int tri = s->indices[nd->left_node + i];
                const orc_triangle& T = s->tris[tri];
                tri_test(o, d, V3(s->verts[T.idx[0]].pos), V3(s->verts[T.idx[1]].pos), V3(s->verts[T.idx[2]].pos), t_tmp,
                         bary_tmp);
                bool res = t_tmp > 0;
                if (!res || t_tmp > tmax || t_tmp < tmin) continue;
                tmax = t_tmp;
                t = t_tmp;
                out.tri = tri;
                out.t = t_tmp;
                out.u = bary_tmp.x;
                out.v = bary_tmp.y;
                if (any) {
                    if (vis) { vis->inner = inner; vis->leaf = leaf; }
                    return true;
                }
            }
            continue;
        }
        inner++;
        float ld, rd;
        const orc_node* L = &s->nodes[nd->left_node];
        const orc_node* R = &s->nodes[nd->left_node + 1];
        bool lh = slab(L->bmin, L->bmax, o, d, ld);
        bool rh = slab(R->bmin, R->bmax, o, d, rd);
        lh = lh && ld < t;
        rh = rh && rd < t;
        if (cull_tmax) {
            lh = lh && ld < segment_end;
            rh = rh && rd < segment_end;
        }
        if (lh && rh) {
            if (ld < rd) {
                stack[sp++] = nd->left_node + 1;
                stack[sp++] = nd->left_node;
            } else {
                stack[sp++] = nd->left_node;
                stack[sp++] = nd->left_node + 1;
            }
        } else if (lh) {
            stack[sp++] = nd->left_node;
        } else if (rh) {
            stack[sp++] = nd->left_node + 1;
        }
    }
    if (vis) { vis->inner = inner; vis->leaf = leaf; }
    return std::isfinite(t);
}
inline orc_mesh flat_mesh(const orc_scene* s) { // the merged mesh of a flat scene as a mesh view
    orc_mesh m;
    m.n_verts = s->n_verts;
    m.verts = s->verts;
    m.n_tris = s->n_tris;
    m.tris = s->tris;
    m.n_nodes = s->n_nodes;
    m.nodes = s->nodes;
    m.indices = s->indices;
    return m;
}
inline bool mesh_traverse(const orc_scene* s, const V3& o, const V3& d, float tmin, float tmax, bool any, MeshHit& out,
                          orc_visits* vis, bool cull_tmax = false) {
    const orc_mesh m = flat_mesh(s);
    return mesh_traverse(&m, o, d, tmin, tmax, any, out, vis, cull_tmax);
}

// ------------------------------------------------- legacy two-level scenes
// Intersection as the Geometry classes fill it (Def.h:11-40); u, v, tri: what the C-ABI reports besides
struct GeoHit {
    float t = INFINITY;
    V3 point = V3(0, 0, 0), normal = V3(0, 0, 0);
    bool front_face = true;
    float u = 0, v = 0;
    int tri = 0;
    void set_face_normal(const V3& dir, const V3& outward) { // Def.h:31-34
        front_face = dot(dir, outward) < 0;
        normal = front_face ? outward : -outward;
    }
    void set_forward_face(const V3& dir, const V3& outward) { // Def.h:35-39
        bool front = dot(dir, outward) < 0;
        front_face = true;
        normal = front ? outward : -outward;
    }
};
inline float c_fmin(float a, float b) { return (float)std::fmin((double)a, (double)b); } // ::fmin on floats, as written
inline float c_fmax(float a, float b) { return (float)std::fmax((double)a, (double)b); }

inline bool sphere_intersect(const float* p, const V3& o, const V3& d, float tmin, float tmax, GeoHit* si) { // Object.h:34-55
    const float radius = p[0];
    V3 center(0, 0, 0);
    V3 oc = o - center;
    float a = dot(d, d);
    float halfb = dot(oc, d);
    float c = dot(oc, oc) - radius * radius;
    float discriminant = halfb * halfb - a * c;
    if (discriminant < 0) return false;
    float sqrt_dist = std::sqrt(discriminant);
    float root = (-halfb - sqrt_dist) / a;
    if (root < tmin || root > tmax) {
        root = (-halfb + sqrt_dist) / a;
        if (root < tmin || root > tmax) return false;
    }
    si->t = root;
    si->point = o + d * root; // Ray::at, Math.h:627-629
    si->set_face_normal(d, (si->point - center) / radius);
    return true;
}
inline bool rectangle_intersect(const float* p, const V3& o, const V3& d, float tmin, float tmax, GeoHit* si) { // Object.h:93-120
    const V3 N(p), T(p + 3), B(p + 6);
    const float u = p[9], v = p[10];
    float denom = dot(N, d);
    if (std::fabs((double)denom) < 0.01) return false;
    float dist = dot(o, N);
    float t = -dist / denom;
    if (t < tmin || t > tmax) return false;
    V3 pi = o + d * t;
    float u_dist = dot(pi, T);
    if (u_dist < -u || u_dist > u) return false;
    float v_dist = dot(pi, B);
    if (v_dist < -v || v_dist > v) return false;
    si->point = pi;
    si->set_face_normal(d, N);
    si->u = u_dist / u;
    si->v = v_dist / v;
    si->t = t;
    return true;
}
inline bool disk_intersect(const float* p, const V3& o, const V3& d, float tmin, float tmax, GeoHit* si) { // Object.h:141-162
    const V3 normal(p);
    const float outer_radius = p[3], inner_radius = p[4];
    float denom = dot(normal, d);
    if (std::fabs((double)denom) < 0.01) return false;
    float dist = dot(o, normal);
    float t = -dist / denom;
    if (t < tmin || t > tmax) return false;
    V3 pi = o + d * t;
    float dist_radius = len(pi);
    if (dist_radius > outer_radius || dist_radius < inner_radius) return false;
    si->point = pi;
    si->set_face_normal(d, normal);
    si->t = t;
    return true;
}
inline bool box_intersect(const float* p, const V3& o, const V3& d, float tmin, float tmax, GeoHit* si) { // Object.h:175-204
    float greatest_minimum = -INFINITY;
    int gm_idx = 0; // (uninitialised in the reference; always assigned before use when the test passes with tmin >= 0)
    for (int a = 0; a < 3; a++) {
        float t0 = c_fmin((p[a] - o[a]) / d[a], (p[3 + a] - o[a]) / d[a]);
        float t1 = c_fmax((p[a] - o[a]) / d[a], (p[3 + a] - o[a]) / d[a]);
        tmin = c_fmax(t0, tmin);
        tmax = c_fmin(t1, tmax);
        if (tmax <= tmin) return false;
        if (tmin > greatest_minimum) {
            greatest_minimum = tmin;
            gm_idx = a;
        }
    }
    int idx = gm_idx;
    float t = greatest_minimum;
    float nrm[3] = {0, 0, 0};
    nrm[idx] = (d[idx] > 0) ? -1.f : 1.f;
    si->point = o + d * t;
    si->set_face_normal(d, V3(nrm));
    si->t = t;
    return true;
}
inline bool cylinder_intersect(const float* p, const V3& o, const V3& d, float tmin, float tmax, GeoHit* si) { // Object.h:216-239
    const float radius = p[0], ymin = p[1], ymax = p[2];
    float a = d.x * d.x + d.z * d.z;
    float b = 2 * (o.x * d.x + o.z * d.z);
    float c = o.x * o.x + o.z * o.z - radius * radius;
    float discriminant = b * b - 4 * a * c;
    if (discriminant < 0) return false;
    float sqrt_dist = std::sqrt(discriminant);
    float root = (-b - sqrt_dist) / (2 * a);
    V3 point = o + d * root;
    if (root < tmin || root > tmax || point.y < ymin || point.y > ymax) {
        root = (-b + sqrt_dist) / (2 * a);
        point = o + d * root;
        if (root < tmin || root > tmax || point.y < ymin || point.y > ymax) return false;
    }
    V3 normal = V3(point.x, 0, point.z) / radius;
    si->point = point;
    si->set_forward_face(d, normal);
    si->t = root;
    return true;
}
// TriangleMesh::intersect (Object.h:273-357): the walk, then the interpolated normal
inline bool trimesh_intersect(const orc_mesh* m, const V3& o, const V3& d, float tmin, float tmax, GeoHit* si, orc_visits* vis) {
    MeshHit h;
    if (!mesh_traverse(m, o, d, tmin, tmax, false, h, vis, g_count_product_occlusion_walk)) return false;
    const orc_triangle& tri = m->tris[h.tri];
    si->point = o + d * h.t;
    float bw = 1 - h.u - h.v; // bary = (bary_temp.z, .x, .y): weights of corners 0, 1, 2 (Object.h:317,351-353)
    V3 N = bw * V3(m->verts[tri.idx[0]].normal) + h.u * V3(m->verts[tri.idx[1]].normal) + h.v * V3(m->verts[tri.idx[2]].normal);
    si->set_face_normal(d, normalize(N));
    si->t = h.t;
    si->u = h.u;
    si->v = h.v;
    si->tri = h.tri;
    return true;
}
inline V3 to_model_vector(const float* im, const V3& v) { // Math.h:675-683
    return V3(im[0] * v.x + im[4] * v.y + im[8] * v.z, im[1] * v.x + im[5] * v.y + im[9] * v.z, im[2] * v.x + im[6] * v.y + im[10] * v.z);
}
inline V3 to_world_normal(const float* im, const V3& n) { // Math.h:667-674
    return V3(im[0] * n.x + im[1] * n.y + im[2] * n.z, im[4] * n.x + im[5] * n.y + im[6] * n.z, im[8] * n.x + im[9] * n.y + im[10] * n.z);
}
// Instance::intersect, Object.h:380-394
inline bool instance_intersect(const orc_two_level* tl, const orc_instance& I, const V3& o, const V3& d, float tmin, float tmax,
                               GeoHit* si, orc_visits* vis) {
    const V3 mo = m4_point(I.inv_m, o), md = to_model_vector(I.inv_m, d); // Transform::to_model_ray, Math.h:692-694
    bool hit = false;
    if (vis) vis->inner = vis->leaf = 0;
    switch (I.geometry) {
    case 0: hit = trimesh_intersect(&tl->meshes[I.mesh], mo, md, tmin, tmax, si, vis); break;
    case 1: hit = sphere_intersect(I.p, mo, md, tmin, tmax, si); break;
    case 2: hit = rectangle_intersect(I.p, mo, md, tmin, tmax, si); break;
    case 3: hit = disk_intersect(I.p, mo, md, tmin, tmax, si); break;
    case 4: hit = box_intersect(I.p, mo, md, tmin, tmax, si); break;
    case 5: hit = cylinder_intersect(I.p, mo, md, tmin, tmax, si); break;
    }
    if (!hit) return false;
    si->point = m4_point(I.m, si->point);
    si->normal = normalize(to_world_normal(I.inv_m, si->normal));
    return true;
}
// Scene::closest_hit, Scene.cpp:17-77.  *index = Intersection::index.  `res` keeps what earlier hits wrote, as the
// reference's reused Intersection does; u, v and tri are reset per accepted hit (they are this build's report fields).
inline bool scene_closest_hit(const orc_two_level* tl, const V3& o, const V3& d, GeoHit* res, int* index, orc_visits* vis) {
    int stack[64];
    stack[0] = 0;
    int sp = 1;
    float t = INFINITY;
    int inner = 0, leaf = 0;
    if (tl->n_instances <= 0) return false;
    while (sp > 0) {
        const orc_node* nd = &tl->tlas_nodes[stack[--sp]];
        if (nd->count != -1) {
            for (int i = 0; i < nd->count; i++) {
                const int k = tl->tlas_indices[nd->left_node + i];
                GeoHit tmp = *res;
                tmp.u = tmp.v = 0;
                tmp.tri = 0;
                orc_visits iv;
                bool hit = instance_intersect(tl, tl->instances[k], o, d, 0, t, &tmp, &iv);
                inner += 3; // visit statistics: the instance's 192-byte record = three 64-byte units
                inner += iv.inner;
                leaf += iv.leaf;
                if (!hit) continue;
                *res = tmp;
                t = res->t;
                *index = k;
            }
            continue;
        }
        inner++;
        float ld, rd;
        const orc_node* L = &tl->tlas_nodes[nd->left_node];
        const orc_node* R = &tl->tlas_nodes[nd->left_node + 1];
        bool lh = slab(L->bmin, L->bmax, o, d, ld);
        bool rh = slab(R->bmin, R->bmax, o, d, rd);
        lh = lh && ld < t;
        rh = rh && rd < t;
        if (lh && rh) {
            if (ld < rd) {
                stack[sp++] = nd->left_node + 1;
                stack[sp++] = nd->left_node;
            } else {
                stack[sp++] = nd->left_node;
                stack[sp++] = nd->left_node + 1;
            }
        } else if (lh) {
            stack[sp++] = nd->left_node;
        } else if (rh) {
            stack[sp++] = nd->left_node + 1;
        }
    }
    if (vis) { vis->inner = inner; vis->leaf = leaf; }
    return std::isfinite(t);
}
// NOT the reference: the product's occlusion walk over a two-level scene (TLAS instead of the loop over all instances,
// boxes beyond the segment's end skipped), restated only so that tools/measure_visits.py can count the records THAT walk
// reads (orc_count_product_occlusion_walk).  An instance test counts as 3 inner visits (its 192-byte record).
inline bool scene_trace_any_product_walk(const orc_two_level* tl, const V3& o, const V3& d, float tmax, orc_visits* vis) {
    const float segment_end = (tmax > 0.f && tmax < INFINITY) ? tmax * 1.0001f + 0.000001f : (tmax <= 0.f ? tmax : INFINITY);
    int stack[64];
    stack[0] = 0;
    int sp = 1, inner = 0, leaf = 0;
    bool hit = false;
    while (sp > 0 && !hit) {
        const orc_node* nd = &tl->tlas_nodes[stack[--sp]];
        if (nd->count != -1) {
            for (int i = 0; i < nd->count && !hit; i++) {
                const orc_instance& I = tl->instances[tl->tlas_indices[nd->left_node + i]];
                inner += 3;
                if (I.geometry == 0) {
                    MeshHit h;
                    orc_visits iv;
                    hit = mesh_traverse(&tl->meshes[I.mesh], m4_point(I.inv_m, o), to_model_vector(I.inv_m, d), 0, tmax, true, h, &iv, true);
                    inner += iv.inner;
                    leaf += iv.leaf;
                } else {
                    GeoHit tmp;
                    hit = instance_intersect(tl, I, o, d, 0, tmax, &tmp, nullptr);
                }
            }
            continue;
        }
        inner++;
        float ld, rd;
        const orc_node* L = &tl->tlas_nodes[nd->left_node];
        const orc_node* R = &tl->tlas_nodes[nd->left_node + 1];
        bool lh = slab(L->bmin, L->bmax, o, d, ld) && ld < segment_end;
        bool rh = slab(R->bmin, R->bmax, o, d, rd) && rd < segment_end;
        if (lh) stack[sp++] = nd->left_node;
        if (rh) stack[sp++] = nd->left_node + 1;
    }
    vis->inner = inner;
    vis->leaf = leaf;
    return hit;
}

// Scene::trace_scene (Scene.h:113-129) as a boolean: the linear loop
inline bool scene_trace_any(const orc_two_level* tl, const V3& o, const V3& d, float tmin, float tmax, orc_visits* vis) {
    bool hit = false;
    float closest_so_far = tmax;
    int inner = 0, leaf = 0;
    for (int i = 0; i < tl->n_instances; i++) {
        GeoHit tmp;
        orc_visits iv;
        if (instance_intersect(tl, tl->instances[i], o, d, tmin, closest_so_far, &tmp, &iv)) {
            hit = true;
            closest_so_far = tmp.t;
        }
        inner += iv.inner;
        leaf += iv.leaf;
    }
    if (vis) { vis->inner = inner; vis->leaf = leaf; }
    return hit;
}

// ------------------------------------------------------------------- random
inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
inline uint32_t rng_key(uint32_t pixel, uint32_t sample, uint32_t seed) {
    uint32_t k = mix32(pixel * 0x9E3779B1u + seed);
    return mix32(k ^ (sample * 0x85EBCA77u + 0x165667B1u));
}
inline float rng_float(uint32_t key, uint32_t ctr) {
    return (float)(mix32(key + ctr * 0xC2B2AE3Du) >> 8) * 5.9604644775390625e-08f; // [0,1)
}

std::mutex g_shared_rng_mutex;
uint32_t g_shared_rng_state = 0x12345678u;

struct Rng {
    uint32_t key = 0, ctr = 0;
    bool shared = false;
    float next() {
        if (shared) { // one locked global stream: the contention profile of glibc rand() (Utils.h:7-10)
            std::lock_guard<std::mutex> g(g_shared_rng_mutex);
            uint32_t x = g_shared_rng_state; // xorshift32, Math.h:723-730
            x ^= x << 13;
            x ^= x >> 17;
            x ^= x << 5;
            g_shared_rng_state = x;
            return (float)(x >> 8) * 5.9604644775390625e-08f;
        }
        return rng_float(key, ctr++);
    }
};

// array-fed stream for the material unit hooks
struct RndSrc {
    Rng* rng = nullptr;
    const float* arr = nullptr;
    int used = 0;
    float next() {
        if (arr) return arr[used++];
        used++;
        return rng->next();
    }
};

// --------------------------------------------------------- shading helpers
inline void onb(const V3& N, V3& T, V3& B) { // Utils.h:53-59
    V3 up(0, 0, 1);
    if (std::fabs(N.z) > 0.999) up = V3(1, 0, 0);
    T = normalize(cross(up, N));
    B = cross(N, T);
}
inline V3 reflect(const V3& v, const V3& n) { return v - (2 * dot(v, n)) * n; } // Utils.h:85-88
inline V3 refract(const V3& uv, const V3& n, float eta) {                        // Utils.h:89-95
    float theta = fmin_sel(dot(-uv, n), 1.0);
    V3 perp = eta * (uv + theta * n);
    float k = -std::sqrt(std::fabs(1.0 - (double)len2(perp)));
    V3 par = k * n;
    return perp + par;
}
inline float schlick_reflectance(float cosine, float ref_idx) { // Utils.h:96-101
    float r0 = (1 - ref_idx) / (1 + ref_idx);
    r0 = r0 * r0;
    return r0 + (1 - r0) * std::pow((double)(1 - cosine), 5.0);
}
inline float power_heuristic(float a, float b) { // Utils.h:130-136
    float denom = a * a + b * b;
    if (std::fabs(denom) < 0.0001) return 0.f;
    return (a * a) / denom;
}
inline V3 cosine_sample(RndSrc& r) { // Utils.h:70-83
    float r1 = r.next();
    float r2 = r.next();
    float phi = 2 * kPi * r1;
    float ct = std::sqrt(1 - r2);
    float st = std::sqrt(r2);
    float x = std::cos((double)phi) * st;
    float y = std::sin((double)phi) * st;
    return V3(x, y, ct);
}

struct Surf { // Intersection, Def.h:11-40
    V3 point, normal;
    bool front_face = true;
    float u = 0, v = 0; // a Rectangle's si->u, si->v; 0 where the Geometry writes none (defined here: the reference leaves them stale)
};

struct Scatter {
    V3 origin = V3(0, 0, 0), dir = V3(1, 0, 0); // Ray(), Math.h:624
    V3 f;
    float pdf = 0;
};

// ---- GGX Microfacet, Material.cpp:4-109
float ggx_D(const V3& N, const V3& H, float alpha) { // Material.cpp:4-13
    float ndh = fmax_sel(dot(H, N), 0);
    if (ndh <= 0) return 0;
    float half_angle = std::acos((double)ndh);
    float a2 = alpha * alpha;
    float denom = (double)kPi * std::pow((double)ndh, 4.0) *
                  std::pow((double)a2 + std::pow(std::tan((double)half_angle), 2.0), 2.0);
    if (denom <= 0) return 0;
    return a2 / denom;
}
// Beckmann distribution (NEW: no reference arithmetic; Walter et al. 2007 eq. 25)
float beckmann_D(const V3& N, const V3& H, float alpha) {
    float ndh = fmax_sel(dot(H, N), 0);
    if (ndh <= 0) return 0;
    float c2 = ndh * ndh;
    float a2 = alpha * alpha;
    float t2 = (1 - c2) / c2;
    float denom = kPi * a2 * c2 * c2;
    if (denom <= 0) return 0;
    return (float)std::exp((double)(-t2 / a2)) / denom;
}
V3 microfacet_sample_h(const V3& N, float alpha, bool beckmann, RndSrc& r) { // Material.cpp:14-30
    float r1 = r.next();
    float r2 = r.next();
    float theta_m;
    if (beckmann)
        theta_m = std::atan(std::sqrt((double)(-(alpha * alpha) * (float)std::log((double)(1 - r1)))));
    else
        theta_m = std::atan(((double)alpha * std::sqrt((double)r1)) / std::sqrt((double)(1 - r1)));
    float phi = 2 * kPi * r2;
    float st = std::sin((double)theta_m);
    V3 m((float)(std::cos((double)phi) * st), (float)(std::sin((double)phi) * st), (float)std::cos((double)theta_m));
    V3 T, B;
    onb(N, T, B);
    return T * m.x + B * m.y + N * m.z;
}
float microfacet_pdf(const V3& V, const V3& H, const V3& N, float alpha, bool beckmann) { // Material.cpp:31-34
    float D = beckmann ? beckmann_D(N, H, alpha) : ggx_D(N, H, alpha);
    return D * fmax_sel(dot(H, N), 0) / (4.f * fmax_sel(dot(V, H), 0) + 0.0000001f);
}
V3 fresnel_schlick(float vdh, const V3& F0) { // Material.cpp:37-40
    float p = std::pow((double)clampf(1.f - vdh, 0.0, 1.0), 5.0);
    return F0 + (V3(1.f) - F0) * p;
}
float schlick_ggx_G1(float ndv, float rough) { // Material.cpp:42-48
    float k = rough * rough * 0.5;
    float denom = ndv * (1 - k) + k;
    if (denom == 0) return 0;
    return ndv / denom;
}
V3 microfacet_eval(const V3& n, const V3& V, const V3& H, const V3& L, float rough, bool beckmann) { // Material.cpp:57-66
    float alpha = rough * rough;
    float D = beckmann ? beckmann_D(n, H, alpha) : ggx_D(n, H, alpha);
    V3 F = fresnel_schlick(fmax_sel(dot(V, H), 0), V3(0.8f));
    float G = schlick_ggx_G1(fmax_sel(dot(n, V), 0), rough) * schlick_ggx_G1(fmax_sel(dot(n, L), 0), rough);
    float denom = 4.0 * fmax_sel(dot(n, V), 0.0) * fmax_sel(dot(n, L), 0.0) + 0.0000001f;
    return (fmax_sel(dot(n, L), 0) * ((D * F) * G)) / denom;
}

// ---- Disney pieces
float diffuse_fresnel(float ndw, float f90) { return (1 + (f90 - 1) * std::pow((double)(1 - ndw), 5.0)); } // Material.cpp:114-117

V3 disney_diffuse_eval(const V3& albedo, float rough, float subsurface, const V3& n, const V3& in_dir, const V3& out_dir,
                       float* pdf) { // Material.cpp:128-153
    *pdf = fmax_sel(dot(n, out_dir), 0.f) / kPi;
    V3 Wh = normalize(-in_dir + out_dir);
    float hdo = fmax_sel(dot(Wh, out_dir), 0);
    float ndo = fmax_sel(dot(n, out_dir), 0);
    float ndi = fmax_sel(dot(n, -in_dir), 0);
    float fd90 = 0.5f + 2.f * rough * hdo * hdo;
    V3 base = ((albedo / kPi) * diffuse_fresnel(ndi, fd90)) * diffuse_fresnel(ndo, fd90) * ndo;
    float fss90 = rough * hdo * hdo;
    float ss = (float)((double)(diffuse_fresnel(ndi, fss90) * diffuse_fresnel(ndo, fss90) * (1.f / (ndi + ndo) - 0.5f)) + 0.5f);
    V3 sub = (((float)1.25 * albedo) / kPi) * ss * ndo;
    return (1 - subsurface) * base + subsurface * sub;
}

float tr_D(const V3& H, float ax, float ay) { // Material.cpp:160-164
    double inner = std::pow((double)(H.x / ax), 2.0) + std::pow((double)(H.y / ay), 2.0) + (double)(H.z * H.z);
    float denom = (double)(kPi * ax * ay) * std::pow(inner, 2.0);
    return 1 / denom;
}
float smith_G1(const V3& w, float ax, float ay) { // Material.cpp:166-171
    float inner = 1 + (w.x * ax * w.x * ax + w.y * ay * w.y * ay) / (w.z * w.z);
    float denom = 1.f + (std::sqrt(1 + inner) - 1) / 2.f;
    return 1 / denom;
}
V3 sample_vndf(const V3& Wi, float ax, float ay, RndSrc& r) { // Material.cpp:178-199
    float r1 = r.next();
    float r2 = r.next();
    V3 Vh = normalize(V3(ax * Wi.x, ay * Wi.y, Wi.z));
    float l2 = Wi.x * Wi.x + Wi.y * Wi.y;
    V3 T1 = l2 > 0 ? V3(-Vh.y, Vh.x, 0) / std::sqrt(l2) : V3(1, 0, 0);
    V3 T2 = cross(T1, Vh);
    float rr = std::sqrt(r1);
    float phi = 2.0 * kPi * r2;
    float t1 = rr * std::cos((double)phi);
    float t2 = rr * std::sin((double)phi);
    float s = 0.5 * (1 + Vh.z);
    t2 = (1 - s) * std::sqrt(1 - t1 * t1) + s * t2;
    V3 Nh = t1 * T1 + t2 * T2 + std::sqrt(fmax_sel(1 - t1 * t1 - t2 * t2, 0)) * Vh;
    return normalize(V3(ax * Nh.x, ay * Nh.y, fmax_sel(0, Nh.z)));
}
inline void aniso_alphas(float rough, float aniso, float& ax, float& ay) { // Material.cpp:216-218
    float aspect = std::sqrt(1.0 - aniso * 0.9);
    ax = fmax_sel(0.001, rough * rough / aspect);
    ay = fmax_sel(0.001, rough * rough * aspect);
}
V3 disney_metal_eval(const V3& V, const V3& L, const V3& H, float ax, float ay, const V3& base, float* pdf) { // Material.cpp:201-212
    V3 F = fresnel_schlick(fmax_sel(dot(V, H), 0), base);
    float D = tr_D(H, ax, ay);
    float g1 = smith_G1(V, ax, ay);
    float g2 = smith_G1(L, ax, ay);
    float G = g1 * g2;
    *pdf = D * g1 / (4 * V.z);
    return ((F * D) * G) / (4 * V.z);
}

float r0_from_ior(float ior) { return std::pow((double)(ior - 1), 2.0) / std::pow((double)(ior + 1), 2.0); } // Material.cpp:256-259
float clearcoat_D(float ag, float hdn) {                                                                    // Material.cpp:260-265
    float ag2 = ag * ag;
    float denom = kPi * (float)std::log((double)ag2) * (1 + (ag2 - 1) * hdn * hdn);
    return (ag2 - 1) / denom;
}
V3 clearcoat_sample(float ag, RndSrc& r) { // Material.cpp:267-280
    float ag2 = ag * ag;
    float r1 = r.next();
    float r2 = r.next();
    float phi = 2 * kPi * r2;
    float ct = std::sqrt((1 - std::pow((double)ag2, (double)(1 - r1))) / (1 - ag2));
    float st = clampf(std::sqrt(1 - (ct * ct)), 0, 1);
    float sp = std::sin((double)phi);
    float cp = std::cos((double)phi);
    return V3(st * cp, st * sp, ct);
}
V3 clearcoat_eval(const V3& V, const V3& L, const V3& H, float gloss, float* pdf) { // Material.cpp:282-294
    float r0 = r0_from_ior(1.5);
    float ag = (1 - gloss) * 0.1 + gloss * 0.001;
    float Fs = r0 + (1 - r0) * std::pow((double)(1 - fmax_sel(dot(H, L), 0)), 5.0);
    V3 F(Fs);
    float D = clearcoat_D(ag, H.z);
    float G = smith_G1(V, 0.25, 0.25) * smith_G1(L, 0.25, 0.25);
    *pdf = clearcoat_D(ag, H.z) / (4 * fmax_sel(dot(H, L), 0) + 0.00000001);
    return ((F * D) * G) / (4 * V.z);
}

float dielectric_fresnel3(float ndi, float ndt, float eta) { // Material.cpp:343-349
    float rs = (ndi - eta * ndt) / (ndi + eta * ndt);
    float rp = (ndi * eta - ndt) / (ndi * eta + ndt);
    return (rs * rs + rp * rp) / 2;
}
float dielectric_fresnel(float ndi, float eta) { // Material.cpp:350-357
    float ndt2 = 1 - (1 - ndi * ndi) / (eta * eta);
    if (ndt2 < 0) return 1;
    float ndt = std::sqrt(ndt2);
    return dielectric_fresnel3(std::fabs(ndi), ndt, eta);
}
V3 disney_refraction_eval(const V3& V, const V3& L, const V3& H, float eta, float ax, float ay, const V3& base, float* pdf) { // Material.cpp:360-377
    float D = tr_D(H, ax, ay);
    float hdi = dot(H, V);
    float F = dielectric_fresnel(hdi, 1 / eta);
    float g1 = smith_G1(V, ax, ay);
    float g2 = smith_G1(L, ax, ay);
    float G = g1 * g2;
    float hdo = std::fabs(dot(H, L));
    float sd = hdi + eta * hdo;
    V3 out = (((((base * (1 - F)) * D) * G) * eta) * eta) * std::fabs(hdo * hdi) / (std::fabs(V.z) * sd * sd);
    float dh = eta * eta * hdo / (sd * sd);
    *pdf = (1 - F) * D * g1 * std::fabs(dh * hdi / V.z);
    return out;
}
V3 disney_reflection_eval(const V3& V, const V3& L, const V3& H, float eta, float ax, float ay, const V3& base, float* pdf) { // Material.cpp:379-392
    float hdi = dot(H, V);
    float F = dielectric_fresnel(hdi, eta);
    float D = tr_D(H, ax, ay);
    float g1 = smith_G1(V, ax, ay);
    float g2 = smith_G1(L, ax, ay);
    float G = g1 * g2;
    *pdf = F * D * g1 / (4 * V.z);
    return (((base * F) * D) * G) / (4 * V.z);
}

V3 sheen_eval(const V3& base, float tint_amt, const V3& n, const V3& in_dir, const V3& out_dir, float* pdf) { // Material.cpp:554-567
    *pdf = fmax_sel(dot(n, out_dir), 0.f) / kPi;
    V3 half = normalize(-in_dir + out_dir);
    float denom = luminance(base);
    if (denom < 0) denom = 1;
    V3 tint = base / denom;
    V3 c = V3(1 - tint_amt) + tint_amt * tint;
    float p = std::pow((double)(1 - std::fabs(dot(half, out_dir))), 5.0);
    return (c * p) * dot(n, out_dir);
}

// DisneyUber, Material.cpp:571-722.  s[] order = Material.h:184-196.
struct Uber {
    V3 base;
    float spec_trans, metallic, subsurface, specular, rough, spec_tint, aniso, sheen, sheen_tint, clearcoat, cc_gloss, ior;
    explicit Uber(const orc_material* m) : base(m->c0) {
        const float* s = m->s;
        spec_trans = s[0]; metallic = s[1]; subsurface = s[2]; specular = s[3]; rough = s[4]; spec_tint = s[5];
        aniso = s[6]; sheen = s[7]; sheen_tint = s[8]; clearcoat = s[9]; cc_gloss = s[10]; ior = s[11];
    }
    V3 eval_diffuse(const V3& V, const V3& L, const V3& H, float* pdf) const { // Material.cpp:571-589
        *pdf = fmax_sel(L.z, 0.f) / kPi;
        float hdo = fmax_sel(dot(H, L), 0);
        float ndo = fmax_sel(L.z, 0);
        float ndi = fmax_sel(V.z, 0);
        float fd90 = 0.5f + 2.f * rough * hdo * hdo;
        V3 b = ((base / kPi) * diffuse_fresnel(ndi, fd90)) * diffuse_fresnel(ndo, fd90) * ndo;
        float fss90 = rough * hdo * hdo;
        float ss = (float)((double)(diffuse_fresnel(ndi, fss90) * diffuse_fresnel(ndo, fss90) * (1.f / (ndi + ndo) - 0.5f)) + 0.5f);
        V3 sub = (((float)1.25 * base) / kPi) * ss * ndo;
        V3 d = (1 - subsurface) * b + subsurface * sub;
        return (d * (1 - spec_trans)) * (1 - metallic);
    }
    V3 eval_metal(const V3& V, const V3& L, const V3& H, float eta, float ax, float ay, float* pdf) const { // Material.cpp:591-609
        V3 c_tint = base / luminance(base);
        V3 ks = V3(1 - spec_tint) + spec_tint * c_tint;
        V3 C0 = ((specular * r0_from_ior(eta)) * (1 - metallic)) * ks + metallic * base;
        float p = std::pow((double)(1 - fmax_sel(dot(L, H), 0)), 5.0);
        V3 F = C0 + (V3(1) - C0) * p;
        float D = tr_D(H, ax, ay);
        float g1 = smith_G1(V, ax, ay);
        float g2 = smith_G1(L, ax, ay);
        float G = g1 * g2;
        *pdf = D * g1 / (4 * V.z);
        V3 f = ((F * D) * G) / (4 * V.z);
        return f * (1 - spec_trans * (1 - metallic));
    }
};

V3 mat_emitted(const orc_material* m) { return m->type == 3 ? V3(m->c0) : V3(0, 0, 0); } // Material.h:47-49,278-280

// Material::Eval dispatch
V3 mat_eval(const orc_material* m, const Surf& si, const V3& in_dir, const V3& out_dir, float* pdf) {
    const V3& n = si.normal;
    switch (m->type) {
    case 0: { // MatteMaterial::Eval, Material.h:81-88
        *pdf = fmax_sel(dot(n, out_dir), 0.f) / kPi;
        V3 albedo(m->c0); // ConstantTexture::sample, Material.h:19-21
        if (m->s[0] != 0) { // CheckeredTexture::sample, Material.h:29-36 (s0 = grid = PI / the constructor's argument, c0 even, c1 odd)
            const float grid = m->s[0];
            auto modulo = [](float a, float b) { // Math.h:39-42
                float r = std::fmod(a, b);
                return (r < 0) ? r + b : r;
            };
            bool odd_ = (modulo(si.u, grid) < grid * 0.5) ^ (modulo(si.v, grid) < grid * 0.5);
            albedo = odd_ ? V3(m->c1) : V3(m->c0);
        }
        return (albedo * fmax_sel(dot(n, out_dir), 0.f)) / kPi;
    }
    case 1:
    case 2: { // Microfacet::Eval, Material.cpp:87-98
        bool beck = m->type == 2;
        V3 Wo = -in_dir;
        V3 Wh = normalize(Wo + out_dir);
        float alpha = m->s[0] * m->s[0];
        *pdf = microfacet_pdf(Wo, Wh, n, alpha, beck);
        return microfacet_eval(n, Wo, Wh, out_dir, m->s[0], beck);
    }
    case 4: return disney_diffuse_eval(V3(m->c0), m->s[0], m->s[1], n, in_dir, out_dir, pdf);
    case 5: { // DisneyMetal::Eval, Material.cpp:234-248
        float ax, ay;
        aniso_alphas(m->s[0], m->s[1], ax, ay);
        V3 T, B;
        onb(n, T, B);
        V3 V(dot(-in_dir, T), dot(-in_dir, B), dot(-in_dir, n));
        V3 L(dot(out_dir, T), dot(out_dir, B), dot(out_dir, n));
        V3 H = normalize(V + L);
        return disney_metal_eval(V, L, H, ax, ay, V3(m->c0), pdf);
    }
    case 6: { // DisneyClearcoat::Eval, Material.cpp:314-325
        V3 T, B;
        onb(n, T, B);
        V3 V(dot(-in_dir, T), dot(-in_dir, B), dot(-in_dir, n));
        V3 L(dot(out_dir, T), dot(out_dir, B), dot(out_dir, n));
        V3 H = normalize(V + L);
        return clearcoat_eval(V, L, H, m->s[0], pdf);
    }
    case 7: { // DisneyGlass::Eval, Material.cpp:432-456
        bool refl = dot(n, out_dir) >= 0;
        float eta = si.front_face ? 1 / m->s[2] : m->s[2];
        float ax, ay;
        aniso_alphas(m->s[0], m->s[1], ax, ay);
        V3 T, B;
        onb(n, T, B);
        V3 V(dot(-in_dir, T), dot(-in_dir, B), dot(-in_dir, n));
        V3 L(dot(out_dir, T), dot(out_dir, B), dot(out_dir, n));
        if (refl) {
            V3 H = normalize(V + L);
            return disney_reflection_eval(V, L, H, eta, ax, ay, V3(m->c0), pdf);
        }
        V3 H = normalize(V + L * eta);
        return disney_refraction_eval(V, L, H, eta, ax, ay, V3(m->c0), pdf);
    }
    case 8: { // RoughDielectric::Eval, Material.cpp:494-541
        bool refl = dot(n, out_dir) <= 0; // as written in the reference
        float eta = si.front_face ? 1 / m->s[0] : m->s[0];
        V3 hv = refl ? normalize(-in_dir + out_dir) : normalize(-in_dir + out_dir * eta);
        V3 T, B;
        onb(n, T, B);
        V3 lin(dot(-in_dir, T), dot(-in_dir, B), dot(-in_dir, n));
        V3 lh(dot(hv, T), dot(hv, B), dot(hv, n));
        V3 lo(dot(out_dir, T), dot(out_dir, B), dot(out_dir, n));
        if (dot(hv, V3(0, 0, 1)) < 0) hv = -hv;
        float hdi = dot(hv, -in_dir);
        float F = dielectric_fresnel(hdi, eta);
        float alpha = m->s[1] * m->s[1];
        float D = tr_D(lh, alpha, alpha);
        float g1 = smith_G1(lin, alpha, alpha);
        float g2 = smith_G1(lo, alpha, alpha);
        float G = g1 * g2;
        if (refl) {
            V3 out = (V3(m->c0) * (F * D * G)) / (4 * std::fabs(lin.z));
            *pdf = (F * D * g1) / (4 * std::fabs(lin.z));
            return out;
        }
        float hdo = std::fabs(dot(hv, out_dir));
        float sd = hdi + eta * hdo;
        V3 out = (((((V3(m->c1) * (1 - F)) * D) * G) * eta) * eta) * std::fabs(hdo * hdi) / (std::fabs(lin.z) * sd * sd);
        float dh = eta * eta * hdo / (sd * sd);
        *pdf = (1 - F) * D * g1 * std::fabs(dh * hdi / lin.z);
        return out;
    }
    case 9: return sheen_eval(V3(m->c0), m->s[0], n, in_dir, out_dir, pdf);
    case 10: { // DisneyUber::Eval, Material.cpp:678-722
        Uber u(m);
        V3 T, B;
        onb(n, T, B);
        V3 V(dot(-in_dir, T), dot(-in_dir, B), dot(-in_dir, n));
        V3 L(dot(out_dir, T), dot(out_dir, B), dot(out_dir, n));
        V3 H = normalize(V + L);
        float dw = (1 - u.metallic) * (1 - u.spec_trans);
        float mw = (1 - u.spec_trans * (1 - u.metallic));
        float gw = (1 - u.metallic) * u.spec_trans;
        float cw = 0.25 * u.clearcoat;
        if (!si.front_face) dw = mw = cw = 0;
        float sum = dw + mw + gw + cw;
        dw /= sum;
        mw /= sum;
        float ax, ay;
        aniso_alphas(u.rough, u.aniso, ax, ay);
        float eta = si.front_face ? 1 / u.ior : u.ior;
        float p, total = 0;
        V3 f(0, 0, 0);
        if (dw > 0) {
            f = f + u.eval_diffuse(V, L, H, &p);
            total += p * dw;
        }
        if (mw > 0) {
            f = f + u.eval_metal(V, L, H, eta, ax, ay, &p);
            total += p * mw;
        }
        *pdf = total;
        return f;
    }
    default: // Material::Eval base (Material.h:52-55): emissive, smooth glass
        *pdf = 0;
        return V3(0, 0, 0);
    }
}

// Material::Sample_Eval dispatch
Scatter mat_sample(const orc_material* m, const Surf& si, const V3& in_dir, RndSrc& r) {
    Scatter sc;
    const V3& n = si.normal;
    switch (m->type) {
    case 0: { // MatteMaterial::Sample_Eval, Material.h:71-79
        V3 d = cosine_sample(r);
        V3 T, B;
        onb(n, T, B);
        d = d.x * T + d.y * B + d.z * n;
        sc.origin = si.point + n * 0.001f;
        sc.dir = d;
        sc.f = mat_eval(m, si, in_dir, d, &sc.pdf);
        return sc;
    }
    case 1:
    case 2: { // Microfacet::Sample_Eval, Material.cpp:68-86
        bool beck = m->type == 2;
        float alpha = m->s[0] * m->s[0];
        V3 Wo = -in_dir;
        V3 Wh = microfacet_sample_h(n, alpha, beck, r);
        V3 Wi = -reflect(-in_dir, Wh);
        sc.origin = si.point + n * (float)0.0001;
        sc.dir = Wi;
        sc.pdf = microfacet_pdf(Wo, Wh, n, alpha, beck);
        if (sc.pdf != sc.pdf) sc.pdf = 0;
        sc.f = microfacet_eval(n, Wo, Wh, Wi, m->s[0], beck);
        return sc;
    }
    case 4: { // DisneyDiffuse::Sample_Eval, Material.cpp:119-127
        V3 d = cosine_sample(r);
        V3 T, B;
        onb(n, T, B);
        d = d.x * T + d.y * B + d.z * n;
        sc.origin = si.point + n * 0.001f;
        sc.dir = d;
        sc.f = mat_eval(m, si, in_dir, d, &sc.pdf);
        return sc;
    }
    case 5: { // DisneyMetal::Sample_Eval, Material.cpp:214-233
        float ax, ay;
        aniso_alphas(m->s[0], m->s[1], ax, ay);
        V3 T, B;
        onb(n, T, B);
        V3 V(dot(-in_dir, T), dot(-in_dir, B), dot(-in_dir, n));
        V3 H = sample_vndf(V, ax, ay, r);
        V3 L = reflect(-V, H);
        sc.f = disney_metal_eval(V, L, H, ax, ay, V3(m->c0), &sc.pdf);
        sc.dir = L.x * T + L.y * B + L.z * n;
        sc.origin = si.point + n * (float)0.00001;
        return sc;
    }
    case 6: { // DisneyClearcoat::Sample_Eval, Material.cpp:297-313
        V3 T, B;
        onb(n, T, B);
        V3 V(dot(-in_dir, T), dot(-in_dir, B), dot(-in_dir, n));
        float ag = (1 - m->s[0]) * 0.1 + m->s[0] * 0.001;
        V3 H = clearcoat_sample(ag, r);
        V3 L = reflect(-V, H);
        sc.f = clearcoat_eval(V, L, H, m->s[0], &sc.pdf);
        sc.dir = L.x * T + L.y * B + L.z * n;
        sc.origin = si.point + n * (float)0.00001;
        return sc;
    }
    case 7: { // DisneyGlass::Sample_Eval, Material.cpp:396-431
        float eta = si.front_face ? (1 / m->s[2]) : m->s[2];
        float ax, ay;
        aniso_alphas(m->s[0], m->s[1], ax, ay);
        V3 T, B;
        onb(n, T, B);
        V3 V(dot(-in_dir, T), dot(-in_dir, B), dot(-in_dir, n));
        V3 H = sample_vndf(V, ax, ay, r);
        float hdi = dot(H, V);
        float F = dielectric_fresnel(hdi, 1 / eta);
        bool do_reflect = F > r.next();
        if (do_reflect) {
            V3 L = normalize(reflect(-V, H));
            sc.dir = L.x * T + L.y * B + L.z * n; // to_world, Utils.h:64-67
            sc.origin = si.point + n * (float)0.00001;
            sc.f = disney_reflection_eval(V, L, H, eta, ax, ay, V3(m->c0), &sc.pdf);
        } else {
            V3 L = normalize(refract(-V, H, eta));
            sc.dir = L.x * T + L.y * B + L.z * n;
            sc.origin = si.point - n * (float)0.00001;
            sc.f = disney_refraction_eval(V, L, H, eta, ax, ay, V3(m->c0), &sc.pdf);
        }
        return sc;
    }
    case 8: { // RoughDielectric::Sample_Eval, Material.cpp:463-493
        float eta = si.front_face ? 1 / m->s[0] : m->s[0];
        float alpha = m->s[1] * m->s[1];
        V3 T, B;
        onb(n, T, B);
        V3 lin(dot(-in_dir, T), dot(-in_dir, B), dot(-in_dir, n));
        V3 lm = sample_vndf(lin, alpha, alpha, r);
        V3 wh = lm.x * T + lm.y * B + lm.z * n;
        float hdi = dot(wh, -in_dir);
        float F = dielectric_fresnel(hdi, eta);
        float r1 = r.next();
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
    case 9: { // DisneySheen::Sample_Eval, Material.cpp:545-553
        V3 d = cosine_sample(r);
        V3 T, B;
        onb(n, T, B);
        d = d.x * T + d.y * B + d.z * n;
        sc.origin = si.point + n * 0.001f;
        sc.dir = d;
        sc.f = mat_eval(m, si, in_dir, d, &sc.pdf);
        return sc;
    }
    case 10: { // DisneyUber::Sample_Eval, Material.cpp:611-677
        Uber u(m);
        V3 T, B;
        onb(n, T, B);
        V3 V(dot(-in_dir, T), dot(-in_dir, B), dot(-in_dir, n));
        float dw = (1 - u.metallic) * (1 - u.spec_trans);
        float mw = (1 - u.spec_trans * (1 - u.metallic));
        float gw = (1 - u.metallic) * u.spec_trans;
        float cw = 0.25 * u.clearcoat;
        if (!si.front_face) dw = mw = cw = 0;
        float c0 = dw, c1 = c0 + mw, c2 = c1 + gw, c3 = c2 + cw;
        float sum = c3;
        float lobe = r.next() * sum;
        float ax, ay;
        aniso_alphas(u.rough, u.aniso, ax, ay);
        float eta = si.front_face ? 1 / u.ior : u.ior;
        sc.pdf = 0;
        sc.f = V3(0, 0, 0);
        if (lobe < c0) {
            V3 L = cosine_sample(r);
            sc.dir = L.x * T + L.y * B + L.z * n;
            sc.origin = si.point + n * 0.001f;
            sc.f = u.eval_diffuse(V, L, normalize(V + L), &sc.pdf);
            sc.pdf = sc.pdf * (dw / sum);
        } else if (lobe < c1) {
            V3 H = sample_vndf(V, ax, ay, r);
            V3 L = reflect(-V, H);
            sc.dir = L.x * T + L.y * B + L.z * n;
            sc.origin = si.point + n * 0.001f;
            sc.f = u.eval_metal(V, L, H, eta, ax, ay, &sc.pdf);
            sc.pdf *= (mw / sum);
        } // glass / clearcoat lobes are unwritten in the reference: f = 0, pdf = 0 (Material.cpp:665-673)
        return sc;
    }
    case 11: { // GlassMaterial::Sample_Eval, Material.h:248-266
        float ratio = si.front_face ? (float)(1.0 / m->s[0]) : m->s[0];
        float ct = fmin_sel(dot(-in_dir, n), 1.0);
        float st = std::sqrt(1.f - ct * ct);
        bool refl = ratio * st > 1.f || schlick_reflectance(ct, ratio) > r.next();
        V3 d = refl ? reflect(in_dir, n) : refract(in_dir, n, ratio);
        sc.origin = si.point + n * (refl ? 0.001f : -0.001f);
        sc.dir = normalize(d);
        sc.pdf = 1;
        sc.f = V3(1, 1, 1);
        return sc;
    }
    default: // Material::Sample_Eval base (Material.h:56-59): emissive
        sc.pdf = 0;
        sc.f = V3(0, 0, 0);
        return sc;
    }
}

// ------------------------------------------------------ environment light
inline int imod(int a, int b) { // Math.h:44-47 (through float, as written)
    float r = a % b;
    return (r < 0) ? r + b : r;
}
inline V3 env_texel(const orc_scene* s, int x, int y) { // Image.h:27-31 (RGB_32f)
    const float* p = s->env_rgb + ((size_t)s->env_w * y + x) * 3;
    return V3(p[0], p[1], p[2]);
}
V3 env_lookup(const orc_scene* s, float u, float v) { // Image.h:46-64
    int w = s->env_w, h = s->env_h;
    u = u * w - 0.5;
    v = v * h - 0.5;
    int ufi = imod((int)u, w);
    int vfi = imod((int)v, h);
    int uci = imod(ufi + 1, w);
    int vci = imod(vfi + 1, h);
    float ip;
    float uo = std::modf(u - ufi, &ip);
    float vo = std::modf(v - vfi, &ip);
    V3 ff = env_texel(s, ufi, vfi), fc = env_texel(s, ufi, vci), cf = env_texel(s, uci, vfi), cc = env_texel(s, uci, vci);
    return (ff * (1 - uo)) * (1 - vo) + (fc * (1 - uo)) * (vo) + (cf * (uo)) * (1 - vo) + (cc * (uo)) * (vo);
}
V3 env_eval(const orc_scene* s, const V3& dir, float* pdf) { // main.cpp:144-164
    float theta = std::acos((double)dir.y);
    float phi = std::atan2((double)dir.z, (double)dir.x);
    float u = (kPi + phi) / kPi * 0.5 + s->env_offset;
    float v = theta / kPi;
    V3 color = env_lookup(s, u, v);
    *pdf = luminance(color) / s->env_total;
    float st = std::sin((double)theta);
    if (st == 0)
        *pdf = 0;
    else
        *pdf *= s->env_w * s->env_h / (2 * kPi * kPi * st);
    return color;
}
V3 env_sample(const orc_scene* s, float rnd, V3* dir, float* pdf) { // main.cpp:121-143,165-184
    int w = s->env_w, h = s->env_h;
    int lo = 0, hi = h - 1;
    while (lo < hi) {
        int mid = (lo + hi) / 2;
        if (rnd < s->env_cdf[(mid + 1) * w - 1])
            hi = mid;
        else
            lo = mid + 1;
    }
    int y = lo;
    lo = 0;
    hi = w - 1;
    while (lo < hi) {
        int mid = (lo + hi) / 2;
        if (rnd < s->env_cdf[y * w + mid])
            hi = mid;
        else
            lo = mid + 1;
    }
    int x = lo;
    float ux = (float)x / w, uy = (float)y / h;
    V3 color = env_lookup(s, ux, uy);
    ux -= s->env_offset;
    float phi = ux * 2 * kPi;
    float theta = uy * kPi;
    double st_d = std::sin((double)theta);
    *dir = V3((float)(-st_d * std::cos((double)phi)), (float)std::cos((double)theta), (float)(-st_d * std::sin((double)phi)));
    *pdf = luminance(color) / s->env_total;
    float st = st_d;
    if (st == 0)
        *pdf = 0;
    else
        *pdf *= w * h / (2 * kPi * kPi * st);
    return color;
}

// ------------------------------------------------------------- integrator
struct Ctx {
    const orc_scene* s;
    uint64_t rays_closest = 0, rays_shadow = 0, inner = 0, leaf = 0, s_inner = 0, s_leaf = 0;
};


bool occluded(Ctx& c, const V3& o, const V3& d, float tmin, float tmax) {
    c.rays_shadow++;
    MeshHit h;
    orc_visits vis;
    bool hit = c.s->two_level ? (g_count_product_occlusion_walk ? scene_trace_any_product_walk(c.s->two_level, o, d, tmax, &vis)
                                                                : scene_trace_any(c.s->two_level, o, d, tmin, tmax, &vis))
                              : mesh_traverse(c.s, o, d, tmin, tmax, true, h, &vis, g_count_product_occlusion_walk);
    c.s_inner += vis.inner;
    c.s_leaf += vis.leaf;
    return hit;
}

// Instance::sample_geometry for a light (Object.h:396-403,124-131) + TRIMESH extension
void light_sample(const orc_scene* s, const orc_light& L, RndSrc& r, V3* point, V3* normal) {
    if (L.kind == 0) {
        float r1 = -1 + (1 - -1) * r.next(); // random_float(-1,1), Utils.h:12-15
        float r2 = -1 + (1 - -1) * r.next();
        V3 p = (V3(L.t) * L.hu) * r1 + (V3(L.b) * L.hv) * r2;
        *point = m4_point(L.m, p);
        const float* im = L.inv_m; // to_world_normal, Math.h:667-674
        V3 N(L.n);
        *normal = V3(im[0] * N.x + im[1] * N.y + im[2] * N.z, im[4] * N.x + im[5] * N.y + im[6] * N.z,
                     im[8] * N.x + im[9] * N.y + im[10] * N.z);
        return;
    }
    float r0 = r.next();
    float b1 = r.next();
    float b2 = r.next();
    int lo = 0, hi = L.count - 1;
    while (lo < hi) {
        int mid = (lo + hi) / 2;
        if (r0 < s->emissive_tris[L.first + mid].cdf)
            hi = mid;
        else
            lo = mid + 1;
    }
    const orc_triangle& T = s->tris[s->emissive_tris[L.first + lo].tri];
    V3 v0(s->verts[T.idx[0]].pos), v1(s->verts[T.idx[1]].pos), v2(s->verts[T.idx[2]].pos);
    float su = std::sqrt(b1);
    float u = 1 - su;
    float v = b2 * su;
    *point = v0 + u * (v1 - v0) + v * (v2 - v0);
    V3 N = (1 - u - v) * V3(s->verts[T.idx[0]].normal) + u * V3(s->verts[T.idx[1]].normal) + v * V3(s->verts[T.idx[2]].normal);
    *normal = normalize(N);
}

V3 nee(Ctx& c, const Surf& si, const orc_material* mat, const V3& ray_dir, RndSrc& r) { // main.cpp:240-317
    const orc_scene* s = c.s;
    V3 direct(0, 0, 0);
    if (s->env_rgb) {
        float env_pdf;
        V3 dir;
        V3 col = env_sample(s, r.next(), &dir, &env_pdf);
        if (dot(dir, si.normal) > 0 && !occluded(c, si.point + si.normal * (float)0.0001, dir, 0, INFINITY)) {
            float bp;
            V3 f = mat_eval(mat, si, ray_dir, dir, &bp);
            float mis = power_heuristic(env_pdf, bp);
            direct = direct + ((mis * f) * col) / env_pdf;
        }
    }
    for (int i = 0; i < s->n_lights; i++) {
        const orc_light& L = s->lights[i];
        V3 sample(0, 0, 0);
        float area = L.area;
        V3 point, normal;
        light_sample(s, L, r, &point, &normal);
        V3 ld = point - si.point;
        float length = len(ld);
        ld = ld / length;
        if (!occluded(c, si.point + si.normal * (float)0.0001, ld, 0, (float)(length - 0.001))) {
            float cos_light = fmax_sel(dot(-normal, ld), 0.f);
            float d2 = length * length;
            float denom = cos_light * area;
            if (!(std::fabs(denom) < 0.00001)) {
                float lpdf = d2 / denom;
                float bp;
                V3 f = mat_eval(mat, si, ray_dir, ld, &bp);
                float mis = power_heuristic(lpdf, bp);
                sample = sample + (mis * f) / lpdf;
            }
        }
        direct = direct + (sample * (1.f / 1)) * mat_emitted(&s->materials[L.material]);
    }
    return direct;
}

void camera_ray(const orc_camera* cam, int x, int y, uint32_t i, uint32_t n, V3* o, V3* d) {
    // hammersley_2d, Utils.h:138-151
    uint32_t bits = i;
    bits = (bits << 16u) | (bits >> 16u);
    bits = ((bits & 0x55555555u) << 1u) | ((bits & 0xAAAAAAAAu) >> 1u);
    bits = ((bits & 0x33333333u) << 2u) | ((bits & 0xCCCCCCCCu) >> 2u);
    bits = ((bits & 0x0F0F0F0Fu) << 4u) | ((bits & 0xF0F0F0F0u) >> 4u);
    bits = ((bits & 0x00FF00FFu) << 8u) | ((bits & 0xFF00FF00u) >> 8u);
    float hx = (float)i / n;
    float hy = float(bits) * 2.3283064365386963e-10;
    float rx = hx + (float)x; // coords += vec2(x, y), main.cpp:978-979
    float ry = hy + (float)y;
    // Camera::get_ray, Scene.h:61-75
    float dx = rx - std::floor(rx);
    float dy = ry - std::floor(ry);
    float ox, oy;
    if (cam->filter == 0) { // GaussianFilter::sample, Filter.h:34-38
        float rad = cam->filter_param * std::sqrt(-2 * std::log((double)fmax_sel(dx, 1e-8)));
        ox = rad * std::cos((double)(2 * kPi * dy));
        oy = rad * std::sin((double)(2 * kPi * dy));
    } else { // BoxFilter::sample, Filter.h:18-23
        ox = 2 * dx - 1;
        oy = 2 * dy - 1;
        ox *= cam->filter_param * 0.5;
        oy *= cam->filter_param * 0.5;
    }
    float px = (std::floor(rx) + 0.5f) + ox;
    float py = (std::floor(ry) + 0.5f) + oy;
    const float* m = cam->raster_to_world;
    float z = 0, w = 1;
    V3 pos(m[0] * px + m[4] * py + m[8] * z + m[12] * w, m[1] * px + m[5] * py + m[9] * z + m[13] * w,
           m[2] * px + m[6] * py + m[10] * z + m[14] * w);
    *o = V3(cam->pos);
    *d = normalize(pos - *o);
}

V3 radiance(Ctx& c, const orc_camera* cam, const orc_options* opt, int x, int y, int i) { // main.cpp:325-441
    const orc_scene* s = c.s;
    V3 ro, rd;
    camera_ray(cam, x, y, (uint32_t)i, (uint32_t)opt->spp_total, &ro, &rd);
    Rng rng;
    rng.key = rng_key((uint32_t)(y * opt->width + x), (uint32_t)i, opt->seed);
    rng.shared = opt->shared_rng != 0;
    RndSrc r;
    r.rng = &rng;

    Surf si;
    V3 prev_point(0, 0, 0);
    float scatter_pdf = 1.f;
    V3 rad(0, 0, 0), T(1.f);
    for (int b = 0; b < opt->max_depth; b++) {
        MeshHit h;
        GeoHit gh;
        int hit_index = -1;
        orc_visits vis;
        c.rays_closest++;
        bool hit = s->two_level ? scene_closest_hit(s->two_level, ro, rd, &gh, &hit_index, &vis)
                                : mesh_traverse(s, ro, rd, 0, INFINITY, false, h, &vis); // Scene.cpp:37: (r, 0, t = INF)
        c.inner += vis.inner;
        c.leaf += vis.leaf;
        if (!hit) { // main.cpp:340-359
            if (s->env_rgb) {
                float env_pdf;
                V3 bg = env_eval(s, rd, &env_pdf);
                float w = 1.f;
                if (b > 0) w = power_heuristic(scatter_pdf, env_pdf);
                rad = rad + (T * bg) * w;
            }
            break;
        }
        const orc_material* mat;
        float hit_area; // world.instances[si.index].get_area(), main.cpp:376
        if (s->two_level) {
            const orc_instance& I = s->two_level->instances[hit_index];
            si.point = gh.point;
            si.normal = gh.normal;
            si.front_face = gh.front_face;
            si.u = I.geometry != 0 ? gh.u : 0.f; // (a mesh hit's gh.u, gh.v are the triangle's barycentrics, reported only)
            si.v = I.geometry != 0 ? gh.v : 0.f;
            mat = &s->materials[I.material]; // Object.h:387
            hit_area = I.area;
        } else {
            // TriangleMesh::intersect tail, Object.h:350-355
            const orc_triangle& tri = s->tris[h.tri];
            si.point = ro + rd * h.t;
            float bw = 1 - h.u - h.v;
            V3 N = bw * V3(s->verts[tri.idx[0]].normal) + h.u * V3(s->verts[tri.idx[1]].normal) + h.v * V3(s->verts[tri.idx[2]].normal);
            N = normalize(N);
            si.front_face = dot(rd, N) < 0; // Def.h:31-34
            si.normal = si.front_face ? N : -N;
            mat = &s->materials[tri.material];
            int light = s->tri_light ? s->tri_light[h.tri] : -1;
            hit_area = light >= 0 ? s->lights[light].area : 0.f; // Geometry::area() default, Object.h:18-20
        }

        V3 emitted = mat_emitted(mat);
        float weight = 1.f;
        if (b > 0 && dot(emitted, V3(1.0f)) > 0.1) { // main.cpp:373-396
            float brdf_pdf = scatter_pdf;
            float cos_light = fmax_sel(dot(rd, -si.normal), 0);
            float area = hit_area;
            float denom = area * cos_light;
            if (denom <= 0 || !si.front_face) {
                weight = 0;
            } else {
                float lpdf = len2(prev_point - si.point) / (denom);
                weight = power_heuristic(brdf_pdf, lpdf);
            }
        }
        rad = rad + (weight * emitted) * T;

        Scatter sc = mat_sample(mat, si, rd, r);
        if (std::fabs(sc.pdf) < 0.000001) break; // main.cpp:410-412 (float abs)

        V3 direct = nee(c, si, mat, rd, r);
        rad = rad + T * direct;
        T = (T * sc.f) / sc.pdf;
        ro = sc.origin;
        rd = sc.dir;
        prev_point = si.point;
        scatter_pdf = sc.pdf;
    }
    return rad;
}

} // namespace

// =================================================================== C-ABI
extern "C" {

int32_t orc_bvh_build(const float* bounds6, int32_t n, int32_t max_per_leaf, int32_t strategy, orc_node* nodes,
                      int32_t* indices) {
    Builder b;
    b.b6 = bounds6;
    b.max_leaf = max_per_leaf;
    b.strat = strategy;
    b.idx.resize(n);
    b.cent.resize(n);
    for (int i = 0; i < n; i++) b.idx[i] = i;
    for (int i = 0; i < n; i++) b.cent[i] = (b.prim(i).lo + b.prim(i).hi) / 2.f; // Math.h:772-774
    b.nodes.reserve(2 * (size_t)n + 1);
    b.nodes.resize(1);
    b.nodes[0].count = -1;
    b.build(0, n, 0);
    std::memcpy(nodes, b.nodes.data(), b.nodes.size() * sizeof(orc_node));
    std::memcpy(indices, b.idx.data(), (size_t)n * sizeof(int32_t));
    return (int32_t)b.nodes.size();
}

void orc_tri_bounds(const orc_vertex* verts, const orc_triangle* tris, int64_t n_tris, float* out6) { // Object.cpp:3-9
    for (int64_t i = 0; i < n_tris; i++) {
        V3 a(verts[tris[i].idx[0]].pos), b(verts[tris[i].idx[1]].pos), c(verts[tris[i].idx[2]].pos);
        V3 lo = vmin(vmin(a, b), c), hi = vmax(vmax(a, b), c);
        lo.store(out6 + 6 * i);
        hi.store(out6 + 6 * i + 3);
    }
}

void orc_closest_hit(const orc_scene* s, const orc_ray* rays, int64_t n, orc_hit* hits, orc_visits* visits) {
    for (int64_t i = 0; i < n; i++) {
        MeshHit h;
        bool hit = mesh_traverse(s, V3(rays[i].o), V3(rays[i].d), 0, INFINITY, false, h, visits ? &visits[i] : nullptr);
        hits[i].tri = hit ? h.tri : -1;
        hits[i].t = hit ? h.t : INFINITY;
        hits[i].u = hit ? h.u : 0;
        hits[i].v = hit ? h.v : 0;
    }
}

void orc_any_hit(const orc_scene* s, const orc_ray* rays, const float* tmax, int64_t n, uint8_t* occ) {
    for (int64_t i = 0; i < n; i++) {
        MeshHit h;
        occ[i] = mesh_traverse(s, V3(rays[i].o), V3(rays[i].d), 0, tmax[i], true, h, nullptr) ? 1 : 0;
    }
}

void orc_closest_hit2(const orc_scene* s, const orc_ray* rays, int64_t n, orc_hit* hits, int32_t* index, float* normal3, uint8_t* front) {
    for (int64_t i = 0; i < n; i++) {
        GeoHit g;
        int k = -1;
        bool hit = scene_closest_hit(s->two_level, V3(rays[i].o), V3(rays[i].d), &g, &k, nullptr);
        hits[i].tri = hit ? g.tri : -1;
        hits[i].t = hit ? g.t : INFINITY;
        hits[i].u = hit ? g.u : 0;
        hits[i].v = hit ? g.v : 0;
        if (index) index[i] = hit ? k : -1;
        if (normal3) (hit ? g.normal : V3(0, 0, 0)).store(normal3 + 3 * i);
        if (front) front[i] = hit && g.front_face ? 1 : 0;
    }
}

void orc_any_hit2(const orc_scene* s, const orc_ray* rays, const float* tmax, int64_t n, uint8_t* occ) {
    for (int64_t i = 0; i < n; i++) occ[i] = scene_trace_any(s->two_level, V3(rays[i].o), V3(rays[i].d), 0, tmax[i], nullptr) ? 1 : 0;
}

void orc_world_bounds(const float* obj6, const float* m, float* out6) { // Math.h:800-812
    const V3 lo(obj6), hi(obj6 + 3);
    Box b;
    b.lo = b.hi = m4_point(m, lo); // Bounds(vec3), Math.h:746
    auto add = [&](const V3& p) { // bounds_union(Bounds, vec3), Math.h:794-799
        V3 w = m4_point(m, p);
        b.lo = vmin(b.lo, w);
        b.hi = vmax(b.hi, w);
    };
    add(V3(hi.x, lo.y, lo.z));
    add(V3(hi.x, lo.y, hi.z));
    add(V3(lo.x, lo.y, hi.z));
    add(V3(lo.x, hi.y, lo.z));
    add(V3(lo.x, hi.y, hi.z));
    add(V3(hi.x, hi.y, lo.z));
    add(V3(hi.x, hi.y, hi.z));
    b.lo.store(out6);
    b.hi.store(out6 + 3);
}

void orc_object_bounds(const orc_instance* I, float* out6) {
    const float* p = I->p;
    V3 lo(0, 0, 0), hi(0, 0, 0);
    switch (I->geometry) {
    case 1: // Sphere, Object.h:31-33
        lo = V3(-p[0], -p[0], -p[0]);
        hi = V3(p[0], p[0], p[0]);
        break;
    case 2: { // Rectangle, Object.h:81-92: Bounds((-u,0,-v),(u,0,v)) under the rotation with columns T, N, B
        float rot[16] = {p[3], p[4], p[5], 0, p[0], p[1], p[2], 0, p[6], p[7], p[8], 0, 0, 0, 0, 1};
        float obj[6] = {-p[9], 0, -p[10], p[9], 0, p[10]};
        orc_world_bounds(obj, rot, out6);
        return;
    }
    case 3: // Disk: the cube around the sphere of its outer radius (defined here; the reference has no bounds for it)
        lo = V3(-p[3], -p[3], -p[3]);
        hi = V3(p[3], p[3], p[3]);
        break;
    case 4: // Box: its own corners (defined here)
        lo = V3(p);
        hi = V3(p + 3);
        break;
    case 5: // Cylinder (defined here)
        lo = V3(-p[0], p[1], -p[0]);
        hi = V3(p[0], p[2], p[0]);
        break;
    }
    lo.store(out6);
    hi.store(out6 + 3);
}

void orc_look_at(const float* eye3, const float* at3, const float* up3, float* view16) { // Math.h:595-610
    V3 p(eye3);
    V3 front = normalize(V3(at3) - p);
    V3 z = -front;
    V3 x = normalize(cross(V3(up3), z));
    V3 y = cross(z, x);
    float m[16] = {x.x, x.y, x.z, 0, y.x, y.y, y.z, 0, z.x, z.y, z.z, 0, p.x, p.y, p.z, 1};
    std::memcpy(view16, m, sizeof(m));
}

void orc_camera_init(const float* view16, float fov, int32_t width, int32_t height, orc_camera* out) { // Scene.h:14-44
    M4 r2s = {{2.f / width, 0, 0, 0, 0, 2.f / height, 0, 0, 0, 0, 1, 0, -1, -1, 1, 1}};
    float h = std::tan((double)((fov / 2.f) * (kPi / 180.f)));
    float aspect = (float)width / height;
    M4 s2c = {{h * aspect, 0, 0, 0, 0, h, 0, 0, 0, 0, -1, 0, 0, 0, 0, 1}};
    M4 view;
    std::memcpy(view.m, view16, sizeof(view.m));
    M4 r2w = m4_mul(m4_mul(view, s2c), r2s);
    std::memcpy(out->raster_to_world, r2w.m, sizeof(r2w.m));
    out->pos[0] = view16[12];
    out->pos[1] = view16[13];
    out->pos[2] = view16[14];
    out->filter = 0;
    out->filter_param = 0.5f; // GaussianFilter(0.5), Scene.h:41
}

void orc_camera_ray(const orc_camera* cam, int32_t x, int32_t y, uint32_t i, uint32_t n, orc_ray* out) {
    V3 o, d;
    camera_ray(cam, x, y, i, n, &o, &d);
    o.store(out->o);
    d.store(out->d);
}

float orc_env_build_cdf(const float* rgb, int32_t w, int32_t h, float* cdf) { // main.cpp:97-120
    size_t n = (size_t)w * h;
    cdf[0] = luminance(V3(rgb));
    for (size_t i = 1; i < n; i++) cdf[i] = cdf[i - 1] + luminance(V3(rgb + 3 * i));
    float total = cdf[n - 1];
    for (size_t i = 1; i < n; i++) cdf[i] = cdf[i] / total;
    return total;
}

void orc_env_eval(const orc_scene* s, const float* dir3, float* rgb3, float* pdf) { env_eval(s, V3(dir3), pdf).store(rgb3); }

void orc_env_sample(const orc_scene* s, float rnd, float* dir3, float* rgb3, float* pdf) {
    V3 d;
    env_sample(s, rnd, &d, pdf).store(rgb3);
    d.store(dir3);
}

void orc_material_eval(const orc_material* m, const float* normal3, int32_t front_face, const float* in_dir3,
                       const float* out_dir3, float* f3, float* pdf) {
    Surf si;
    si.normal = V3(normal3);
    si.front_face = front_face != 0;
    mat_eval(m, si, V3(in_dir3), V3(out_dir3), pdf).store(f3);
}

int32_t orc_material_sample(const orc_material* m, const float* point3, const float* normal3, int32_t front_face,
                            const float* in_dir3, const float* rnd, float* out_origin3, float* out_dir3, float* f3,
                            float* pdf) {
    Surf si;
    si.point = V3(point3);
    si.normal = V3(normal3);
    si.front_face = front_face != 0;
    RndSrc r;
    r.arr = rnd;
    Scatter sc = mat_sample(m, si, V3(in_dir3), r);
    sc.origin.store(out_origin3);
    sc.dir.store(out_dir3);
    sc.f.store(f3);
    *pdf = sc.pdf;
    return r.used;
}

// batch forms of the four calls above (tests integrate pdfs by quadrature and histogram samplers: 10^5-10^6 calls)
void orc_env_eval_batch(const orc_scene* s, const float* dirs3, int64_t n, float* rgb3, float* pdf) {
    for (int64_t i = 0; i < n; i++) orc_env_eval(s, dirs3 + 3 * i, rgb3 + 3 * i, pdf + i);
}
void orc_env_sample_batch(const orc_scene* s, const float* rnd, int64_t n, float* dirs3, float* rgb3, float* pdf) {
    for (int64_t i = 0; i < n; i++) orc_env_sample(s, rnd[i], dirs3 + 3 * i, rgb3 + 3 * i, pdf + i);
}
void orc_material_eval_batch(const orc_material* m, const float* normal3, int32_t front_face, const float* in_dir3,
                             const float* out_dirs3, int64_t n, float* f3, float* pdf) {
    for (int64_t i = 0; i < n; i++) orc_material_eval(m, normal3, front_face, in_dir3, out_dirs3 + 3 * i, f3 + 3 * i, pdf + i);
}
void orc_material_sample_batch(const orc_material* m, const float* point3, const float* normal3, int32_t front_face,
                               const float* in_dir3, const float* rnd4, int64_t n, float* out_origin3, float* out_dirs3, float* f3,
                               float* pdf) { // four random numbers per sample (no material draws more)
    for (int64_t i = 0; i < n; i++)
        orc_material_sample(m, point3, normal3, front_face, in_dir3, rnd4 + 4 * i, out_origin3 + 3 * i, out_dirs3 + 3 * i, f3 + 3 * i, pdf + i);
}

uint32_t orc_rng_key(uint32_t pixel, uint32_t sample, uint32_t seed) { return rng_key(pixel, sample, seed); }
float orc_rng_draw(uint32_t key, uint32_t counter) { return rng_float(key, counter); }

void orc_count_product_occlusion_walk(int32_t on) { g_count_product_occlusion_walk = on != 0; }

void orc_radiance(const orc_scene* s, const orc_camera* cam, const orc_options* o, int32_t x, int32_t y, int32_t i,
                  float* rgb3, orc_counters* c) {
    Ctx ctx;
    ctx.s = s;
    radiance(ctx, cam, o, x, y, i).store(rgb3);
    if (c) {
        c->rays_closest += ctx.rays_closest;
        c->rays_shadow += ctx.rays_shadow;
        c->inner_visits += ctx.inner;
        c->leaf_visits += ctx.leaf;
        c->shadow_inner_visits += ctx.s_inner;
        c->shadow_leaf_visits += ctx.s_leaf;
        c->paths += 1;
    }
}

void orc_render(const orc_scene* s, const orc_camera* cam, const orc_options* o, float* sum, orc_counters* c) {
    // Renderer::render_image + TileQueue, main.cpp:904-935,950-1005: 16x16 tiles, x outer / y inner, all spp of a
    // pixel consecutively, one std::thread per hardware thread.
    const int extent = 16;
    int tw = (o->x1 - o->x0 + extent - 1) / extent, th = (o->y1 - o->y0 + extent - 1) / extent;
    std::atomic<int> next(0);
    int nthreads = o->threads;
    if (nthreads <= 0) nthreads = (int)std::thread::hardware_concurrency();
    if (nthreads <= 0) nthreads = 1;
    std::mutex cm;
    auto t0 = std::chrono::steady_clock::now();
    auto work = [&]() {
        Ctx ctx;
        ctx.s = s;
        uint64_t paths = 0;
        for (;;) {
            int t = next.fetch_add(1);
            if (t >= tw * th) break;
            int tx0 = o->x0 + (t % tw) * extent, ty0 = o->y0 + (t / tw) * extent;
            int tx1 = std::min(tx0 + extent, o->x1), ty1 = std::min(ty0 + extent, o->y1);
            for (int x = tx0; x < tx1; x++)
                for (int y = ty0; y < ty1; y++) {
                    float* px = sum + ((size_t)y * o->width + x) * 3;
                    V3 total(px);
                    for (int k = 0; k < o->spp_count; k++) {
                        V3 sv = radiance(ctx, cam, o, x, y, o->spp_begin + k);
                        if (sv.x != sv.x) sv = V3(0, 0, 0); // main.cpp:984-985
                        total = total + sv;
                        paths++;
                    }
                    total.store(px);
                }
        }
        std::lock_guard<std::mutex> g(cm);
        if (c) {
            c->rays_closest += ctx.rays_closest;
            c->rays_shadow += ctx.rays_shadow;
            c->inner_visits += ctx.inner;
            c->leaf_visits += ctx.leaf;
            c->shadow_inner_visits += ctx.s_inner;
            c->shadow_leaf_visits += ctx.s_leaf;
            c->paths += paths;
        }
    };
    std::vector<std::thread> pool;
    for (int i = 0; i < nthreads; i++) pool.emplace_back(work);
    for (auto& t : pool) t.join();
    if (c) c->seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

void orc_resolve(const float* sum, int32_t w, int32_t h, int32_t spp, float gamma, uint8_t* rgb) { // main.cpp:865-877,990
    float scale = 1.0 / spp;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const float* p = sum + ((size_t)y * w + x) * 3;
            uint8_t* q = rgb + ((size_t)(h - y - 1) * w + x) * 3;
            for (int k = 0; k < 3; k++) {
                float cch = p[k] * scale;
                cch = std::pow((double)cch, (double)(1 / gamma));
                q[k] = (uint8_t)(clampf(cch, 0, 1) * 255);
            }
        }
}

} // extern "C"
