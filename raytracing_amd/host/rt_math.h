// rt_math.h — host-side POD math for the Scene/Mesh/BVH mirror.
// Same names and float recipes as the reference's Math.h (vec3 92-157,
// mat4 338-423, Ray 622-633, Transform 641-698, Bounds 743-786) where a result
// feeds the GPU path: select-based min/max, vec/float as reciprocal multiply.
#pragma once
#include <cmath>
#include <cstdint>

namespace rt {

const float PI = 3.1415926536;

inline float radians(float degrees) { return degrees * (PI / 180.f); }
inline float fmax_sel(float a, float b) { return (a > b) ? a : b; } // Math.h:17-20
inline float fmin_sel(float a, float b) { return (a < b) ? a : b; } // Math.h:21-24

struct vec2 {
    float x = 0, y = 0;
    vec2() {}
    vec2(float x_, float y_) : x(x_), y(y_) {}
};

struct vec3 {
    float x = 0, y = 0, z = 0;
    vec3() {}
    vec3(float e0, float e1, float e2) : x(e0), y(e1), z(e2) {}
    explicit vec3(float e) : x(e), y(e), z(e) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    vec3 operator-() const { return vec3(-x, -y, -z); }
    float length_squared() const { return x * x + y * y + z * z; }
    float length() const { return std::sqrt(length_squared()); }
};
inline vec3 operator+(const vec3& a, const vec3& b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(const vec3& a, const vec3& b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator*(const vec3& a, const vec3& b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline vec3 operator*(float t, const vec3& v) { return vec3(t * v.x, t * v.y, t * v.z); }
inline vec3 operator*(const vec3& v, float t) { return vec3(v.x * t, v.y * t, v.z * t); }
inline vec3 operator/(const vec3& v, float t) { return (1 / t) * v; } // Math.h:193-196
inline float dot(const vec3& a, const vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(const vec3& u, const vec3& v) {
    return vec3(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x);
}
inline vec3 normalize(const vec3& v) { return v / v.length(); }
inline vec3 vec_min(const vec3& a, const vec3& b) { return vec3(fmin_sel(a.x, b.x), fmin_sel(a.y, b.y), fmin_sel(a.z, b.z)); }
inline vec3 vec_max(const vec3& a, const vec3& b) { return vec3(fmax_sel(a.x, b.x), fmax_sel(a.y, b.y), fmax_sel(a.z, b.z)); }

struct vec4 {
    float x = 0, y = 0, z = 0, w = 0;
    vec4() {}
    vec4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
    vec4(const vec3& v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
    vec3 xyz() const { return vec3(x, y, z); }
};

// column-major, m[c*4 + r]
struct mat4 {
    float m[16];
    mat4() {
        for (float& v : m) v = 0;
    }
    explicit mat4(float d) {
        for (float& v : m) v = 0;
        m[0] = m[5] = m[10] = m[15] = d;
    }
    static mat4 from_columns(const float* c16) {
        mat4 r;
        for (int i = 0; i < 16; i++) r.m[i] = c16[i];
        return r;
    }
    vec4 column(int c) const { return vec4(m[4 * c], m[4 * c + 1], m[4 * c + 2], m[4 * c + 3]); }
    vec4 operator*(const vec4& v) const {
        return vec4(m[0] * v.x + m[4] * v.y + m[8] * v.z + m[12] * v.w, m[1] * v.x + m[5] * v.y + m[9] * v.z + m[13] * v.w,
                    m[2] * v.x + m[6] * v.y + m[10] * v.z + m[14] * v.w, m[3] * v.x + m[7] * v.y + m[11] * v.z + m[15] * v.w);
    }
    mat4 operator*(const mat4& b) const {
        mat4 r;
        for (int c = 0; c < 4; c++)
            for (int rr = 0; rr < 4; rr++)
                r.m[c * 4 + rr] = m[rr] * b.m[c * 4] + m[4 + rr] * b.m[c * 4 + 1] + m[8 + rr] * b.m[c * 4 + 2] +
                                  m[12 + rr] * b.m[c * 4 + 3];
        return r;
    }
};

inline mat4 transpose(const mat4& a) {
    mat4 r;
    for (int c = 0; c < 4; c++)
        for (int rr = 0; rr < 4; rr++) r.m[c * 4 + rr] = a.m[rr * 4 + c];
    return r;
}

// determinant / inverse as the reference evaluates them (Math.h:425-542): fp32 throughout, the adjugate over the
// determinant, every cofactor a left-to-right sum of six (the determinant: twenty-four) left-to-right products, in the
// reference's term order — so Transform::inv_m and the camera chain carry the reference's bits.  The terms are kept as
// tables of signed element indices (column-major, Math.h:341-346).
namespace detail {
struct Term3 { int8_t sign, a, b, c; };
struct Term4 { int8_t sign, a, b, c, d; };
constexpr Term4 kDet[24] = {
    {+1, 12, 9, 6, 3},  {-1, 8, 13, 6, 3},  {-1, 12, 5, 10, 3}, {+1, 4, 13, 10, 3}, {+1, 8, 5, 14, 3},  {-1, 4, 9, 14, 3},
    {-1, 12, 9, 2, 7},  {+1, 8, 13, 2, 7},  {+1, 12, 1, 10, 7}, {-1, 0, 13, 10, 7}, {-1, 8, 1, 14, 7},  {+1, 0, 9, 14, 7},
    {+1, 12, 5, 2, 11}, {-1, 4, 13, 2, 11}, {-1, 12, 1, 6, 11}, {+1, 0, 13, 6, 11}, {+1, 4, 1, 14, 11}, {-1, 0, 5, 14, 11},
    {-1, 8, 5, 2, 15},  {+1, 4, 9, 2, 15},  {+1, 8, 1, 6, 15},  {-1, 0, 9, 6, 15},  {-1, 4, 1, 10, 15}, {+1, 0, 5, 10, 15}};
constexpr Term3 kAdj[16][6] = {
    {{+1, 9, 14, 7}, {-1, 13, 10, 7}, {+1, 13, 6, 11}, {-1, 5, 14, 11}, {-1, 9, 6, 15}, {+1, 5, 10, 15}},
    {{+1, 13, 10, 3}, {-1, 9, 14, 3}, {-1, 13, 2, 11}, {+1, 1, 14, 11}, {+1, 9, 2, 15}, {-1, 1, 10, 15}},
    {{+1, 5, 14, 3}, {-1, 13, 6, 3}, {+1, 13, 2, 7}, {-1, 1, 14, 7}, {-1, 5, 2, 15}, {+1, 1, 6, 15}},
    {{+1, 9, 6, 3}, {-1, 5, 10, 3}, {-1, 9, 2, 7}, {+1, 1, 10, 7}, {+1, 5, 2, 11}, {-1, 1, 6, 11}},
    {{+1, 12, 10, 7}, {-1, 8, 14, 7}, {-1, 12, 6, 11}, {+1, 4, 14, 11}, {+1, 8, 6, 15}, {-1, 4, 10, 15}},
    {{+1, 8, 14, 3}, {-1, 12, 10, 3}, {+1, 12, 2, 11}, {-1, 0, 14, 11}, {-1, 8, 2, 15}, {+1, 0, 10, 15}},
    {{+1, 12, 6, 3}, {-1, 4, 14, 3}, {-1, 12, 2, 7}, {+1, 0, 14, 7}, {+1, 4, 2, 15}, {-1, 0, 6, 15}},
    {{+1, 4, 10, 3}, {-1, 8, 6, 3}, {+1, 8, 2, 7}, {-1, 0, 10, 7}, {-1, 4, 2, 11}, {+1, 0, 6, 11}},
    {{+1, 8, 13, 7}, {-1, 12, 9, 7}, {+1, 12, 5, 11}, {-1, 4, 13, 11}, {-1, 8, 5, 15}, {+1, 4, 9, 15}},
    {{+1, 12, 9, 3}, {-1, 8, 13, 3}, {-1, 12, 1, 11}, {+1, 0, 13, 11}, {+1, 8, 1, 15}, {-1, 0, 9, 15}},
    {{+1, 4, 13, 3}, {-1, 12, 5, 3}, {+1, 12, 1, 7}, {-1, 0, 13, 7}, {-1, 4, 1, 15}, {+1, 0, 5, 15}},
    {{+1, 8, 5, 3}, {-1, 4, 9, 3}, {-1, 8, 1, 7}, {+1, 0, 9, 7}, {+1, 4, 1, 11}, {-1, 0, 5, 11}},
    {{+1, 12, 9, 6}, {-1, 8, 13, 6}, {-1, 12, 5, 10}, {+1, 4, 13, 10}, {+1, 8, 5, 14}, {-1, 4, 9, 14}},
    {{+1, 8, 13, 2}, {-1, 12, 9, 2}, {+1, 12, 1, 10}, {-1, 0, 13, 10}, {-1, 8, 1, 14}, {+1, 0, 9, 14}},
    {{+1, 12, 5, 2}, {-1, 4, 13, 2}, {-1, 12, 1, 6}, {+1, 0, 13, 6}, {+1, 4, 1, 14}, {-1, 0, 5, 14}},
    {{+1, 4, 9, 2}, {-1, 8, 5, 2}, {+1, 8, 1, 6}, {-1, 0, 9, 6}, {-1, 4, 1, 10}, {+1, 0, 5, 10}}};
} // namespace detail

inline float determinant(const mat4& a) { // Math.h:425-452
    const float* m = a.m;
    float acc = 0.f;
    for (int k = 0; k < 24; k++) {
        const detail::Term4& t = detail::kDet[k];
        const float p = ((m[t.a] * m[t.b]) * m[t.c]) * m[t.d];
        if (k == 0) acc = p; // the first term is positive
        else acc = t.sign > 0 ? acc + p : acc - p;
    }
    return acc;
}
inline mat4 inverse(const mat4& a) { // Math.h:454-542; a singular matrix is returned unchanged (457-458)
    const float det = determinant(a);
    if (det == 0.f) return a;
    const float inv_det = 1.f / det;
    const float* m = a.m;
    mat4 r;
    for (int e = 0; e < 16; e++) {
        float acc = 0.f;
        for (int k = 0; k < 6; k++) {
            const detail::Term3& t = detail::kAdj[e][k];
            const float p = (m[t.a] * m[t.b]) * m[t.c];
            if (k == 0) acc = p; // every cofactor starts with a positive term
            else acc = t.sign > 0 ? acc + p : acc - p;
        }
        r.m[e] = inv_det * acc;
    }
    return r;
}

inline mat4 translate(const mat4& m, const vec3& v) { // Math.h:554-561
    mat4 r(1.f);
    r.m[12] = v.x;
    r.m[13] = v.y;
    r.m[14] = v.z;
    return r * m;
}
inline mat4 scale(const mat4& m, const vec3& s) { // Math.h:586-593
    mat4 r(1.f);
    r.m[0] = s.x;
    r.m[5] = s.y;
    r.m[10] = s.z;
    return r * m;
}
inline mat4 rotate_y(const mat4& m, float degrees) { // Math.h:562-573
    float rad = radians(degrees);
    float s = std::sin(rad), c = std::cos(rad);
    mat4 r(1.f);
    r.m[0] = c;
    r.m[8] = s;
    r.m[2] = -s;
    r.m[10] = c;
    return r * m;
}
inline mat4 rotate_x(const mat4& m, float degrees) { // Math.h:574-585
    float rad = radians(degrees);
    float s = std::sin(rad), c = std::cos(rad);
    mat4 r(1.f);
    r.m[5] = c;
    r.m[9] = -s;
    r.m[6] = s;
    r.m[10] = c;
    return r * m;
}
inline mat4 look_at(const vec3& eye, const vec3& at, const vec3& up) { // Math.h:595-610
    vec3 z = -normalize(at - eye);
    vec3 x = normalize(cross(up, z));
    vec3 y = cross(z, x);
    mat4 r(1.f);
    r.m[0] = x.x; r.m[1] = x.y; r.m[2] = x.z;
    r.m[4] = y.x; r.m[5] = y.y; r.m[6] = y.z;
    r.m[8] = z.x; r.m[9] = z.y; r.m[10] = z.z;
    r.m[12] = eye.x; r.m[13] = eye.y; r.m[14] = eye.z;
    return r;
}
// rotation matrix of a unit quaternion (x, y, z, w) — glTF node TRS
inline mat4 from_quat(float x, float y, float z, float w) {
    mat4 r(1.f);
    r.m[0] = 1 - 2 * (y * y + z * z); r.m[1] = 2 * (x * y + z * w);     r.m[2] = 2 * (x * z - y * w);
    r.m[4] = 2 * (x * y - z * w);     r.m[5] = 1 - 2 * (x * x + z * z); r.m[6] = 2 * (y * z + x * w);
    r.m[8] = 2 * (x * z + y * w);     r.m[9] = 2 * (y * z - x * w);     r.m[10] = 1 - 2 * (x * x + y * y);
    return r;
}

struct Ray { // Math.h:622-633
    vec3 pos = vec3(0, 0, 0), dir = vec3(1, 0, 0);
    Ray() {}
    Ray(const vec3& p, const vec3& d) : pos(p), dir(d) {}
    vec3 at(float t) const { return pos + dir * t; }
};

class Transform { // Math.h:641-698
public:
    Transform() : m(1.f), inv_m(1.f) {}
    explicit Transform(const mat4& model_to_world) : m(model_to_world), inv_m(inverse(model_to_world)) {}
    vec3 to_world_point(const vec3& p) const { return (m * vec4(p, 1.0f)).xyz(); }
    vec3 to_model_point(const vec3& p) const { return (inv_m * vec4(p, 1.0f)).xyz(); }
    vec3 to_world_normal(const vec3& n) const {
        return vec3(inv_m.m[0] * n.x + inv_m.m[1] * n.y + inv_m.m[2] * n.z, inv_m.m[4] * n.x + inv_m.m[5] * n.y + inv_m.m[6] * n.z,
                    inv_m.m[8] * n.x + inv_m.m[9] * n.y + inv_m.m[10] * n.z);
    }
    vec3 to_model_vector(const vec3& v) const { // Math.h:675-683
        return vec3(inv_m.m[0] * v.x + inv_m.m[4] * v.y + inv_m.m[8] * v.z, inv_m.m[1] * v.x + inv_m.m[5] * v.y + inv_m.m[9] * v.z,
                    inv_m.m[2] * v.x + inv_m.m[6] * v.y + inv_m.m[10] * v.z);
    }
    const mat4& matrix() const { return m; }
    const mat4& inverse_matrix() const { return inv_m; }

private:
    mat4 m, inv_m;
};

struct Bounds { // Math.h:743-786
    vec3 bmin = vec3(INFINITY), bmax = vec3(-INFINITY);
    Bounds() {}
    explicit Bounds(const vec3& p) : bmin(p), bmax(p) {}
    Bounds(const vec3& lo, const vec3& hi) : bmin(lo), bmax(hi) {}
    float surface_area() const {
        vec3 s = bmax - bmin;
        return 2.0 * (s.x * s.y + s.x * s.z + s.y * s.z);
    }
    vec3 get_center() const { return (bmin + bmax) / 2.f; }
    int longest_axis() const {
        vec3 l = bmax - bmin;
        int a = 0;
        if (l[1] > l[a]) a = 1;
        if (l[2] > l[a]) a = 2;
        return a;
    }
};
inline Bounds bounds_union(const Bounds& a, const Bounds& b) { return Bounds(vec_min(a.bmin, b.bmin), vec_max(a.bmax, b.bmax)); }
inline Bounds bounds_union(const Bounds& a, const vec3& v) { return Bounds(vec_min(a.bmin, v), vec_max(a.bmax, v)); }
inline Bounds transform_bounds(const Bounds& o, const Transform& t) { // Bounds::transform_bounds, Math.h:800-812: corner order as written
    Bounds b(t.to_world_point(o.bmin));
    b = bounds_union(b, t.to_world_point(vec3(o.bmax.x, o.bmin.y, o.bmin.z)));
    b = bounds_union(b, t.to_world_point(vec3(o.bmax.x, o.bmin.y, o.bmax.z)));
    b = bounds_union(b, t.to_world_point(vec3(o.bmin.x, o.bmin.y, o.bmax.z)));
    b = bounds_union(b, t.to_world_point(vec3(o.bmin.x, o.bmax.y, o.bmin.z)));
    b = bounds_union(b, t.to_world_point(vec3(o.bmin.x, o.bmax.y, o.bmax.z)));
    b = bounds_union(b, t.to_world_point(vec3(o.bmax.x, o.bmax.y, o.bmin.z)));
    b = bounds_union(b, t.to_world_point(vec3(o.bmax.x, o.bmax.y, o.bmax.z)));
    return b;
}

} // namespace rt
