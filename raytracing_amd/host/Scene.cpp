// Scene.cpp — host scene: mesh BVH build, light tables, flattening into the
// GPU layout of include/rt_hip.h, upload, and the query/render forwards.
#include "Scene.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <thread>

namespace rt {

// ------------------------------------------------------------------ Mesh
Bounds tri_bounds(const vec3& v1, const vec3& v2, const vec3& v3) { // Object.cpp:3-9
    Bounds b(v1);
    b = bounds_union(b, v2);
    b = bounds_union(b, v3);
    return b;
}

void Mesh::build() { build(1, BVH_SAH); } // Object.cpp:21

void Mesh::build(int max_per_leaf, PartitionStrategy strat) {
    std::vector<Bounds> b;
    b.reserve(triangles.size());
    for (const MeshTriangle& t : triangles)
        b.push_back(tri_bounds(verts[t.indicies[0]].position, verts[t.indicies[1]].position, verts[t.indicies[2]].position));
    accelStructure = BVH::build(b, max_per_leaf, strat);
}

// --------------------------------------------------------- Filter / Camera
vec2 BoxFilter::sample(float dx, float dy) const { // Filter.h:18-23
    vec2 s(2 * dx - 1, 2 * dy - 1);
    s.x *= width * 0.5;
    s.y *= width * 0.5;
    return s;
}
vec2 GaussianFilter::sample(float dx, float dy) const { // Filter.h:34-38
    float r = stddev * std::sqrt(-2 * std::log((double)fmax_sel(dx, 1e-8)));
    return vec2(r * std::cos((double)(2 * PI * dy)), r * std::sin((double)(2 * PI * dy)));
}

Camera::Camera(const mat4& view_matrix, float fov, int width, int height, Filter* f) { // Scene.h:14-44
    mat4 raster_to_screen;
    raster_to_screen.m[0] = 2.f / width;
    raster_to_screen.m[5] = 2.f / height;
    raster_to_screen.m[10] = 1;
    raster_to_screen.m[12] = -1;
    raster_to_screen.m[13] = -1;
    raster_to_screen.m[14] = 1;
    raster_to_screen.m[15] = 1;
    float h = std::tan((double)radians(fov / 2.f));
    float aspect = (float)width / height;
    mat4 screen_to_camera;
    screen_to_camera.m[0] = h * aspect;
    screen_to_camera.m[5] = h;
    screen_to_camera.m[10] = -1;
    screen_to_camera.m[15] = 1;
    raster_to_world = view_matrix * screen_to_camera * raster_to_screen;
    camera_pos = view_matrix.column(3).xyz();
    filter = f ? f : new GaussianFilter(0.5);
}

Camera& Camera::operator=(Camera&& other) { // Scene.h:49-58
    delete filter;
    filter = other.filter;
    camera_pos = other.camera_pos;
    raster_to_world = other.raster_to_world;
    other.filter = nullptr;
    return *this;
}

Ray Camera::get_ray(float raster_x, float raster_y) const { // Scene.h:61-75
    float dx = raster_x - std::floor(raster_x);
    float dy = raster_y - std::floor(raster_y);
    vec2 offset = filter->sample(dx, dy);
    float px = (std::floor(raster_x) + 0.5f) + offset.x;
    float py = (std::floor(raster_y) + 0.5f) + offset.y;
    vec3 pos = (raster_to_world * vec4(px, py, 0, 1)).xyz();
    return Ray(camera_pos, normalize(pos - camera_pos));
}

rt_camera Camera::flatten() const {
    rt_camera c{};
    std::memcpy(c.raster_to_world, raster_to_world.m, sizeof(c.raster_to_world));
    c.pos[0] = camera_pos.x;
    c.pos[1] = camera_pos.y;
    c.pos[2] = camera_pos.z;
    c.filter = RT_FILTER_GAUSSIAN;
    c.filter_param = 0.5f;
    if (filter) filter->describe(&c.filter, &c.filter_param);
    return c;
}

// ------------------------------------------------------------------ Image
vec3 Image::get(int x, int y) const { // Image.h:19-43
    switch (internalFormat) {
    case Format::RGBA_32f: {
        const float* d = (const float*)imageData.data() + ((size_t)width * y + x) * 4;
        return vec3(d[0], d[1], d[2]);
    }
    case Format::RGB_32f: {
        const float* d = (const float*)imageData.data() + ((size_t)width * y + x) * 3;
        return vec3(d[0], d[1], d[2]);
    }
    case Format::RGBA_u8: {
        const uint8_t* d = (const uint8_t*)imageData.data() + ((size_t)width * y + x) * 4;
        return vec3(d[0], d[1], d[2]) / 255.f;
    }
    case Format::RGB_u8: {
        const uint8_t* d = (const uint8_t*)imageData.data() + ((size_t)width * y + x) * 3;
        return vec3(d[0], d[1], d[2]) / 255.f;
    }
    default: return vec3(0);
    }
}

HDRImage* make_uniform_environment(int width, int height, vec3 radiance) {
    HDRImage* img = new HDRImage;
    img->width = width;
    img->height = height;
    img->internalFormat = Image::Format::RGB_32f;
    img->imageData.resize((size_t)width * height * 3 * sizeof(float));
    float* d = (float*)img->imageData.data();
    for (int i = 0; i < width * height; i++) {
        d[3 * i] = radiance.x;
        d[3 * i + 1] = radiance.y;
        d[3 * i + 2] = radiance.z;
    }
    return img;
}

// -------------------------------------------------------------- Rectangle
Rectangle::Rectangle(vec3 normal, vec3 tangent, float u_length, float v_length) { // Object.h:64-70
    N = normal;
    T = tangent;
    B = normalize(cross(N, T));
    u = u_length / 2;
    v = v_length / 2;
}
Rectangle::Rectangle(vec3 normal, float u_length, float v_length) { // Object.h:71-80
    N = normal;
    vec3 up(0, 1, 0);
    if (N.y < -0.999) up = vec3(1, 0, 0);
    else if (N.y > 0.999) up = vec3(0, 0, -1);
    T = normalize(cross(up, N));
    B = normalize(cross(N, T));
    u = u_length / 2;
    v = v_length / 2;
}

Bounds Rectangle::object_bounds() const { // Object.h:81-92
    Bounds b(vec3(-u, 0, -v), vec3(u, 0, v));
    const float cols[16] = {T.x, T.y, T.z, 0, N.x, N.y, N.z, 0, B.x, B.y, B.z, 0, 0, 0, 0, 1};
    mat4 rotation = mat4::from_columns(cols);
    return transform_bounds(b, Transform(rotation));
}

static void put3(float* d, const vec3& v) {
    d[0] = v.x;
    d[1] = v.y;
    d[2] = v.z;
}
void Sphere::describe(rt_instance* o) const {
    o->geometry = RT_GEO_SPHERE;
    o->p[0] = radius;
}
void Rectangle::describe(rt_instance* o) const {
    o->geometry = RT_GEO_RECTANGLE;
    put3(o->p, N);
    put3(o->p + 3, T);
    put3(o->p + 6, B);
    o->p[9] = u;
    o->p[10] = v;
}
void Disk::describe(rt_instance* o) const {
    o->geometry = RT_GEO_DISK;
    put3(o->p, normal);
    o->p[3] = outer_radius;
    o->p[4] = inner_radius;
}
void Box::describe(rt_instance* o) const {
    o->geometry = RT_GEO_BOX;
    put3(o->p, bmin);
    put3(o->p + 3, bmax);
}
void Cylinder::describe(rt_instance* o) const {
    o->geometry = RT_GEO_CYLINDER;
    o->p[0] = radius;
    o->p[1] = ymin;
    o->p[2] = ymax;
}
TriangleMesh::TriangleMesh(const Mesh* mesh_, int max_per_node) : mesh(mesh_) { // Object.cpp:11-29
    std::vector<Bounds> b;
    b.reserve(mesh->triangles.size());
    for (const MeshTriangle& t : mesh->triangles)
        b.push_back(tri_bounds(mesh->verts[t.indicies[0]].position, mesh->verts[t.indicies[1]].position, mesh->verts[t.indicies[2]].position));
    bvh = BVH::build(b, max_per_node < 1 ? 1 : max_per_node, BVH_SAH);
    mesh_bounds = bvh.nodes[0].aabb;
}
void TriangleMesh::describe(rt_instance* o) const { o->geometry = RT_GEO_MESH; } // blas_root: Scene::flatten_two_level

// ------------------------------------------------------------------ Scene
Scene::~Scene() {
    if (gpu_scene) rt_scene_destroy(gpu_scene);
    for (auto& r : replicas) rt_scene_destroy(r.second);
    // instances may share a geometry (one TriangleMesh placed many times) and their materials are also in `materials`
    std::set<const void*> freed;
    auto once = [&](auto* p) {
        if (p && freed.insert(p).second) delete p;
    };
    for (Material* m : materials) once(m);
    for (Image* i : images) delete i;
    for (Instance& l : lights) {
        once(l.geometry);
        once(l.material);
    }
    for (Instance& i : instances) {
        once(i.geometry);
        once(i.material);
    }
    delete environment;
}

bool Scene::replace_material(size_t index, Material* m) {
    if (index >= materials.size() || !m) {
        delete m;
        return false;
    }
    Material* old = materials[index];
    if (old == m) return true;
    for (Instance& i : instances)
        if (i.material == old) i.material = m;
    for (Instance& l : lights)
        if (l.material == old) l.material = m;
    for (Material*& other : materials) // (a material listed twice stays one object)
        if (other == old) other = m;
    delete old;
    return true;
}

void Scene::set_environment(HDRImage* img, float degrees_offset) {
    delete environment;
    environment = img;
    environment_offset = degrees_offset / 360.f; // main.cpp:94
}

void Scene::add_rectangle_light(vec3 normal, float u_length, float v_length, vec3 emit, const mat4& transform) {
    Rectangle* rect = new Rectangle(normal, u_length, v_length);
    lights.push_back(Instance(rect, new EmissiveMaterial(emit), transform));
    const Transform& tf = lights.back().get_transform();
    const uint32_t mat = (uint32_t)materials.size();
    materials.push_back(new EmissiveMaterial(emit));

    vec3 n = normalize(tf.to_world_normal(rect->N)); // Object.h:389-390
    vec3 tu = rect->T * rect->u, bv = rect->B * rect->v;
    vec3 corner[4] = {-tu - bv, tu - bv, tu + bv, -tu + bv};
    const uint32_t base = (uint32_t)sceneMergedMesh.verts.size();
    for (int i = 0; i < 4; i++) {
        Vertex vtx;
        vtx.position = tf.to_world_point(corner[i]);
        vtx.normal = n;
        vtx.uv = vec2((i == 1 || i == 2) ? 1.f : 0.f, (i >= 2) ? 1.f : 0.f);
        sceneMergedMesh.verts.push_back(vtx);
    }
    light_first_tri.push_back((int)sceneMergedMesh.triangles.size());
    MeshTriangle a, b;
    a.indicies[0] = base; a.indicies[1] = base + 1; a.indicies[2] = base + 2;
    b.indicies[0] = base; b.indicies[1] = base + 2; b.indicies[2] = base + 3;
    a.materialIndex = b.materialIndex = mat;
    sceneMergedMesh.triangles.push_back(a);
    sceneMergedMesh.triangles.push_back(b);
}

void Scene::add_instance(const Mesh& mesh, const mat4& model_to_world, int material_override) {
    const Transform tf(model_to_world);
    const uint32_t base = (uint32_t)sceneMergedMesh.verts.size();
    for (const Vertex& v : mesh.verts) {
        Vertex w = v;
        w.position = tf.to_world_point(v.position);
        w.normal = normalize(tf.to_world_normal(v.normal));
        sceneMergedMesh.verts.push_back(w);
    }
    for (MeshTriangle t : mesh.triangles) {
        for (int c = 0; c < 3; c++) t.indicies[c] += base;
        if (material_override >= 0) t.materialIndex = (uint32_t)material_override;
        sceneMergedMesh.triangles.push_back(t);
    }
}

void Scene::build_acceleration(int max_per_leaf, PartitionStrategy strat) {
    if (!instances.empty()) { // Scene::build_top_level_bvh, Scene.cpp:6-15
        std::vector<Bounds> bounds;
        for (size_t i = 0; i < instances.size(); i++) bounds.push_back(instances[i].get_world_bounds());
        tlas = BVH::build(bounds, 1, BVH_SAH);
        return;
    }
    sceneMergedMesh.build(max_per_leaf, strat);
}

// Child-pair records of one BVH.  Numbering: the first top_pairs branch nodes in breadth-first order, then the rest in
// pre-order (a subtree stays close in memory); layout only — visit order and arithmetic are those of the node graph.
// leaf_ref(node) encodes a leaf; records are appended to `out`, refs are absolute indices into it.  Returns the root's ref.
template <class LeafRef, class LeafVisit>
static int32_t append_pairs(const BVH& bvh, int top_pairs, std::vector<rt_bvh_pair>& out, LeafRef leaf_ref, LeafVisit leaf_visit) {
    const int first = (int)out.size();
    std::vector<int> pair_of(bvh.nodes.size(), -1);
    int next = first;
    std::vector<int> level{0};
    for (size_t head = 0; head < level.size() && next - first < top_pairs; head++) {
        const BVHNode& nd = bvh.nodes[level[head]];
        if (nd.count != BVH_BRANCH) continue;
        pair_of[level[head]] = next++;
        level.push_back(nd.left_node);
        level.push_back(nd.left_node + 1);
    }
    std::vector<int> stack{0};
    while (!stack.empty()) {
        int n = stack.back();
        stack.pop_back();
        const BVHNode& nd = bvh.nodes[n];
        if (nd.count != BVH_BRANCH) {
            leaf_visit(nd);
            continue;
        }
        if (pair_of[n] < 0) pair_of[n] = next++;
        stack.push_back(nd.left_node + 1);
        stack.push_back(nd.left_node);
    }
    out.resize(next);
    auto ref_of = [&](int n) -> int32_t {
        const BVHNode& nd = bvh.nodes[n];
        return nd.count == BVH_BRANCH ? pair_of[n] : leaf_ref(nd);
    };
    for (size_t n = 0; n < bvh.nodes.size(); n++) {
        if (pair_of[n] < 0) continue;
        const BVHNode& L = bvh.nodes[bvh.nodes[n].left_node];
        const BVHNode& R = bvh.nodes[bvh.nodes[n].left_node + 1];
        rt_bvh_pair& p = out[pair_of[n]];
        put3(p.lmin, L.aabb.bmin);
        put3(p.lmax, L.aabb.bmax);
        put3(p.rmin, R.aabb.bmin);
        put3(p.rmax, R.aabb.bmax);
        p.lref = ref_of(bvh.nodes[n].left_node);
        p.rref = ref_of(bvh.nodes[n].left_node + 1);
        p.reserved[0] = p.reserved[1] = 0;
    }
    return ref_of(0);
}

// Triangles of `mesh` in the leaf order of `bvh`, appended to f.tris / f.normals (rt_tri.tri_id = index within the mesh).
static void append_triangles(const Mesh& mesh, const BVH& bvh, const std::vector<int32_t>* tri_light, FlatScene& f) {
    const size_t base = f.tris.size(), n_tris = mesh.triangles.size();
    f.tris.resize(base + n_tris);
    f.normals.resize(base + n_tris);
    for (size_t k = 0; k < n_tris; k++) {
        const int t = bvh.indicies[k];
        const MeshTriangle& T = mesh.triangles[t];
        const Vertex &a = mesh.verts[T.indicies[0]], &b = mesh.verts[T.indicies[1]], &c = mesh.verts[T.indicies[2]];
        rt_tri& o = f.tris[base + k];
        put3(o.v0, a.position);
        put3(o.e1, b.position - a.position); // Object.h:247-248
        put3(o.e2, c.position - a.position);
        o.tri_id = (uint32_t)t;
        o.material = (int32_t)T.materialIndex;
        o.light = tri_light ? (*tri_light)[t] : -1;
        rt_tri_normals& nn = f.normals[base + k];
        std::memset(&nn, 0, sizeof(nn));
        put3(nn.n0, a.normal);
        put3(nn.n1, b.normal);
        put3(nn.n2, c.normal);
    }
}

static void flatten_environment(const HDRImage* environment, FlatScene& f) {
    // EnviormentLight::build_cdf, main.cpp:97-120 (sequential fp32 prefix sum; cdf[0] is left un-normalised as written)
    if (environment && environment->width > 0) {
        const int w = environment->width, h = environment->height;
        f.env_cdf.resize((size_t)w * h);
        auto lum = [](const vec3& c) { return 0.212671f * c.x + 0.715160f * c.y + 0.072169f * c.z; };
        f.env_cdf[0] = lum(environment->get(0, 0));
        for (int i = 1; i < w * h; i++) f.env_cdf[i] = f.env_cdf[i - 1] + lum(environment->get(i % w, i / w));
        f.env_total = f.env_cdf[(size_t)w * h - 1];
        for (int i = 1; i < w * h; i++) f.env_cdf[i] = f.env_cdf[i] / f.env_total;
    }
}

static rt_light rect_light_record(const Instance& inst, int32_t material) {
    const Rectangle* rect = static_cast<const Rectangle*>(inst.get_geometry());
    rt_light L{};
    L.kind = RT_LIGHT_RECT;
    L.material = material;
    L.area = inst.get_area(); // ignores transform scaling, Object.h:404-407
    put3(L.n, rect->N);
    put3(L.t, rect->T);
    put3(L.b, rect->B);
    L.hu = rect->u;
    L.hv = rect->v;
    std::memcpy(L.m, inst.get_transform().matrix().m, sizeof(L.m));
    std::memcpy(L.inv_m, inst.get_transform().inverse_matrix().m, sizeof(L.inv_m));
    return L;
}

// Two-level scene: every distinct TriangleMesh's BVH + triangles (model space), the TLAS, the instance table.
void Scene::flatten_two_level() {
    FlatScene& f = flat_scene;
    auto material_index = [&](Material* m) -> int32_t { // a material the scene does not hold yet joins Scene::materials
        for (size_t i = 0; i < materials.size(); i++)
            if (materials[i] == m) return (int32_t)i;
        materials.push_back(m);
        return (int32_t)materials.size() - 1;
    };
    for (Instance& inst : instances) material_index(inst.material);
    for (Instance& l : lights) material_index(l.material);
    std::map<const Geometry*, int32_t> blas_root;
    // (RT_TUNE_HOST_TOP: the count for experiments with a device library built for another RT_TOP_PAIRS)
    const int kTopPairs = std::getenv("RT_TUNE_HOST_TOP") ? std::max(1, std::atoi(std::getenv("RT_TUNE_HOST_TOP"))) : 256;
    for (const Instance& inst : instances) {
        const TriangleMesh* tm = dynamic_cast<const TriangleMesh*>(inst.get_geometry());
        if (!tm || blas_root.count(tm)) continue;
        const int32_t tri_base = (int32_t)f.tris.size();
        append_triangles(*tm->mesh, tm->bvh, nullptr, f);
        blas_root[tm] = append_pairs(
            tm->bvh, kTopPairs, f.pairs, [&](const BVHNode& nd) { return ~(tri_base + nd.left_node); },
            [&](const BVHNode& nd) { f.tris[tri_base + nd.left_node + nd.count - 1].tri_id |= RT_TRI_LEAF_END; });
    }
    for (const Material* m : materials) f.materials.push_back(m->flatten());
    for (size_t i = 0; i < instances.size(); i++) {
        const Instance& inst = instances[i];
        rt_instance r{};
        inst.get_geometry()->describe(&r);
        r.material = material_index(inst.material);
        r.blas_root = r.geometry == RT_GEO_MESH ? blas_root[inst.get_geometry()] : 0;
        r.area = inst.get_area();
        std::memcpy(r.m, inst.get_transform().matrix().m, sizeof(r.m));
        std::memcpy(r.inv_m, inst.get_transform().inverse_matrix().m, sizeof(r.inv_m));
        f.instances.push_back(r);
    }
    for (const Instance& l : lights) f.lights.push_back(rect_light_record(l, material_index(l.material)));
    // TLAS: BVH::build(bounds, 1, ...) leaves one instance per leaf (Scene.cpp:13, 33-43)
    f.tlas_root_ref = append_pairs(
        tlas, kTopPairs, f.tlas_pairs, [&](const BVHNode& nd) { return ~tlas.indicies[nd.left_node]; }, [](const BVHNode&) {});
    flatten_environment(environment, f);
}

void Scene::flatten() {
    FlatScene& f = flat_scene;
    f = FlatScene();
    if (!instances.empty()) {
        if (!tlas.nodes.empty()) flatten_two_level();
        return;
    }
    const Mesh& mesh = sceneMergedMesh;
    const BVH& bvh = mesh.accelStructure;
    const size_t n_tris = mesh.triangles.size();
    if (bvh.nodes.empty()) return;

    // ---- lights: rectangles first (Scene::lights order), then one TRIMESH
    // light per emissive material that still owns triangles.
    f.tri_light.assign(n_tris, -1);
    for (size_t i = 0; i < lights.size(); i++) {
        const rt_light L = rect_light_record(lights[i], (int32_t)mesh.triangles[light_first_tri[i]].materialIndex);
        f.lights.push_back(L);
        f.tri_light[light_first_tri[i]] = (int32_t)i;
        f.tri_light[light_first_tri[i] + 1] = (int32_t)i;
    }
    std::map<uint32_t, std::vector<int>> emissive_by_material;
    for (size_t t = 0; t < n_tris; t++) {
        if (f.tri_light[t] >= 0) continue;
        uint32_t m = mesh.triangles[t].materialIndex;
        if (m >= materials.size()) continue;
        vec3 e = materials[m]->emitted();
        if (e.x > 0 || e.y > 0 || e.z > 0) emissive_by_material[m].push_back((int)t);
    }
    // slot of every original triangle
    std::vector<int> slot_of(n_tris, -1);
    for (size_t k = 0; k < bvh.indicies.size(); k++) slot_of[bvh.indicies[k]] = (int)k;
    for (auto& kv : emissive_by_material) {
        rt_light L{};
        L.kind = RT_LIGHT_TRIMESH;
        L.material = (int32_t)kv.first;
        L.first = (int32_t)f.emissive.size();
        L.count = (int32_t)kv.second.size();
        std::vector<float> areas;
        double total = 0;
        for (int t : kv.second) {
            const MeshTriangle& T = mesh.triangles[t];
            vec3 e1 = mesh.verts[T.indicies[1]].position - mesh.verts[T.indicies[0]].position;
            vec3 e2 = mesh.verts[T.indicies[2]].position - mesh.verts[T.indicies[0]].position;
            float a = 0.5f * cross(e1, e2).length();
            areas.push_back(a);
            total += a;
        }
        L.area = (float)total;
        double run = 0;
        for (size_t i = 0; i < kv.second.size(); i++) {
            run += areas[i];
            rt_emissive_tri et;
            et.slot = slot_of[kv.second[i]];
            et.cdf = (i + 1 == kv.second.size()) ? 1.0f : (float)(run / total);
            f.emissive.push_back(et);
            f.emissive_orig.push_back(kv.second[i]);
            f.tri_light[kv.second[i]] = (int32_t)f.lights.size();
        }
        f.lights.push_back(L);
    }

    // ---- triangles in leaf order, child-pair records (the top levels first: the traversal kernels keep them in LDS)
    append_triangles(mesh, bvh, &f.tri_light, f);
    constexpr int kTopPairs = 256;
    f.root_ref = append_pairs(
        bvh, kTopPairs, f.pairs, [](const BVHNode& nd) { return ~nd.left_node; },
        [&](const BVHNode& nd) { f.tris[nd.left_node + nd.count - 1].tri_id |= RT_TRI_LEAF_END; });

    // ---- materials
    for (const Material* m : materials) f.materials.push_back(m->flatten());

    flatten_environment(environment, f);
}

bool Scene::upload(int dev) {
    if (gpu_scene) {
        rt_scene_destroy(gpu_scene);
        gpu_scene = nullptr;
    }
    device = dev;
    gpu_scene = create_on(dev);
    return gpu_scene != nullptr;
}

bool Scene::upload_replicas(const std::vector<int>& devices) {
    for (auto& r : replicas) rt_scene_destroy(r.second);
    replicas.clear();
    for (int dev : devices) {
        rt_scene* g = create_on(dev);
        if (!g) {
            for (auto& r : replicas) rt_scene_destroy(r.second);
            replicas.clear();
            return false;
        }
        replicas.push_back({dev, g});
    }
    return true;
}

rt_scene* Scene::create_on(int dev) const {
    rt_scene* made = nullptr;
    const FlatScene& f = flat_scene;
    rt_scene_desc d{};
    d.root_ref = f.root_ref;
    d.n_pairs = (int64_t)f.pairs.size();
    d.pairs = f.pairs.data();
    d.n_tris = (int64_t)f.tris.size();
    d.tris = f.tris.data();
    d.normals = f.normals.data();
    d.n_materials = (int32_t)f.materials.size();
    d.materials = f.materials.data();
    d.n_lights = (int32_t)f.lights.size();
    d.lights = f.lights.data();
    d.n_emissive_tris = (int32_t)f.emissive.size();
    d.emissive_tris = f.emissive.data();
    std::vector<float> env_rgb;
    if (!f.env_cdf.empty()) {
        const int w = environment->width, h = environment->height;
        env_rgb.resize((size_t)w * h * 3);
        for (int i = 0; i < w * h; i++) put3(&env_rgb[3 * (size_t)i], environment->get(i % w, i / w));
        d.env_w = w;
        d.env_h = h;
        d.env_rgb = env_rgb.data();
        d.env_cdf = f.env_cdf.data();
        d.env_total = f.env_total;
        d.env_offset = environment_offset;
    }
    rt_tlas_desc tl{};
    tl.root_ref = f.tlas_root_ref;
    tl.n_pairs = (int64_t)f.tlas_pairs.size();
    tl.pairs = f.tlas_pairs.data();
    tl.n_instances = (int32_t)f.instances.size();
    tl.instances = f.instances.data();
    if ((f.instances.empty() ? rt_scene_create(&d, dev, &made) : rt_scene_create_instanced(&d, &tl, dev, &made)) != RT_OK) {
        std::printf("Scene upload failed: %s\n", rt_last_error());
        return nullptr;
    }
    return made;
}

void Scene::build_top_level_bvh() {
    build_acceleration();
    flatten();
    upload(device);
}

bool Scene::closest_hit(Ray r, Intersection* res) const { // Scene.cpp:17-77 + Object.h:350-355
    if (!gpu_scene) {
        std::printf("Scene::closest_hit: scene is not on a GPU (call build_top_level_bvh first)\n");
        return false;
    }
    rt_ray ray;
    put3(ray.o, r.pos);
    put3(ray.d, r.dir);
    rt_hit hit;
    if (!instances.empty()) { // Scene.cpp:17-77 over Instance::intersect (Object.h:380-394)
        int32_t index = -1;
        if (rt_trace_closest_indexed(gpu_scene, &ray, 1, &hit, &index) != RT_OK || index < 0) return false;
        const Instance& inst = instances[index];
        const Transform& tf = inst.get_transform();
        const Ray mr(tf.to_model_point(r.pos), tf.to_model_vector(r.dir));
        res->t = hit.t;
        res->index = index;
        res->material = inst.get_material();
        res->u = hit.u;
        res->v = hit.v;
        res->w0 = -r.dir;
        vec3 n; // the geometry's model-space normal, re-derived from the hit the GPU found
        bool forward_face = false;
        const Geometry* g = inst.get_geometry();
        const vec3 pm = mr.at(hit.t);
        if (const TriangleMesh* tm = dynamic_cast<const TriangleMesh*>(g)) {
            const MeshTriangle& T = tm->mesh->triangles[hit.tri];
            n = normalize((1 - hit.u - hit.v) * tm->mesh->verts[T.indicies[0]].normal + hit.u * tm->mesh->verts[T.indicies[1]].normal +
                          hit.v * tm->mesh->verts[T.indicies[2]].normal);
        } else if (const Sphere* sp = dynamic_cast<const Sphere*>(g)) {
            n = pm / sp->radius;
        } else if (const Rectangle* rc = dynamic_cast<const Rectangle*>(g)) {
            n = rc->N;
        } else if (const Disk* dk = dynamic_cast<const Disk*>(g)) {
            n = dk->normal;
        } else if (const Cylinder* cy = dynamic_cast<const Cylinder*>(g)) {
            n = vec3(pm.x, 0, pm.z) / cy->radius;
            forward_face = true; // set_forward_face, Def.h:35-39
        } else if (const Box* bx = dynamic_cast<const Box*>(g)) {
            int axis = 0; // Box::intersect's own slab loop (Object.h:176-198): the axis that set the entry distance
            float tmin = 0, greatest_minimum = -INFINITY;
            for (int a = 0; a < 3; a++) {
                float t0 = std::fmin((bx->bmin[a] - mr.pos[a]) / mr.dir[a], (bx->bmax[a] - mr.pos[a]) / mr.dir[a]);
                tmin = std::fmax(t0, tmin);
                if (tmin > greatest_minimum) {
                    greatest_minimum = tmin;
                    axis = a;
                }
            }
            n = vec3(0);
            n[axis] = mr.dir[axis] > 0 ? -1.f : 1.f;
        }
        const bool front = dot(mr.dir, n) < 0;
        res->front_face = forward_face ? true : front;
        n = front ? n : -n;
        res->point = tf.to_world_point(pm);
        res->normal = normalize(tf.to_world_normal(n));
        return true;
    }
    if (rt_trace_closest(gpu_scene, &ray, 1, &hit) != RT_OK || hit.tri < 0) return false;
    const MeshTriangle& T = sceneMergedMesh.triangles[hit.tri];
    res->t = hit.t;
    res->point = r.at(hit.t);
    vec3 N = (1 - hit.u - hit.v) * sceneMergedMesh.verts[T.indicies[0]].normal + hit.u * sceneMergedMesh.verts[T.indicies[1]].normal +
             hit.v * sceneMergedMesh.verts[T.indicies[2]].normal;
    N = normalize(N);
    res->front_face = dot(r.dir, N) < 0;
    res->normal = res->front_face ? N : -N;
    res->index = 0;
    res->material = T.materialIndex < materials.size() ? materials[T.materialIndex] : nullptr;
    res->w0 = -r.dir;
    return true;
}

bool Scene::any_hit(Ray r) const { // declared at Scene.h:133, never defined in the reference
    if (!gpu_scene) {
        std::printf("Scene::any_hit: scene is not on a GPU (call build_top_level_bvh first)\n");
        return false;
    }
    rt_ray ray;
    put3(ray.o, r.pos);
    put3(ray.d, r.dir);
    float tmax = INFINITY;
    uint8_t occ = 0;
    if (rt_trace_any(gpu_scene, &ray, &tmax, 1, &occ) != RT_OK) return false;
    return occ != 0;
}

// --------------------------------------------------------------- Renderer
void Renderer::initalize(const Options& options) { // main.cpp:936-948
    width = options.width;
    height = options.height;
    samples = options.samples_per_pixel;
    final_buffer.assign((size_t)width * height * 3, 0);
    radiance_sum.assign((size_t)width * height * 3, 0.f);
}

void shard_sample_window(int rank, int world, int spp, int* begin, int* count) {
    if (world < 1 || rank < 0 || rank >= world || spp < 0) { // (a caller's mistake must not become a division by zero)
        *begin = *count = 0;
        return;
    }
    const int base = spp / world, extra = spp % world; // the first `extra` ranks take one sample more
    *begin = rank * base + (rank < extra ? rank : extra);
    *count = base + (rank < extra ? 1 : 0);
}

std::vector<int32_t> shard_tiles(int rank, int world, int width, int height, int tile) {
    std::vector<int32_t> out;
    if (world < 1 || rank < 0 || rank >= world || tile < 1 || width < 1 || height < 1) return out; // world 0 would loop for ever, tile 0 divide by zero
    const int tx = (width + tile - 1) / tile, ty = (height + tile - 1) / tile;
    for (int t = rank; t < tx * ty; t += world) { // row-major tile index, as TileQueue counts (main.cpp:914-928)
        const int x0 = (t % tx) * tile, y0 = (t / tx) * tile;
        out.insert(out.end(), {x0, y0, std::min(x0 + tile, width), std::min(y0 + tile, height)});
    }
    return out;
}

// Options::devices names G GPUs: one host thread per GPU renders its share into a frame in that GPU's memory (no data-path
// collective), rt_reduce sums the frames into rank 0's, rank 0 resolves.  Where the reference starts hardware_concurrency()
// threads that pull 16x16 tiles (main.cpp:954-1005), this starts G threads that each own a GPU.
bool Renderer::render_sharded(const Camera& cam, const Scene& scene, const Options& options) {
    const int G = (int)options.devices.size();
    if ((int)scene.replica_count() != G) {
        std::printf("Renderer: Scene::upload_replicas(options.devices) must have placed the scene on the %d devices first\n", G);
        return false;
    }
    for (int r = 0; r < G; r++)
        if (scene.replica_device((size_t)r) != options.devices[(size_t)r]) {
            std::printf("Renderer: replica %d is on device %d, Options::devices says %d\n", r, scene.replica_device((size_t)r), options.devices[(size_t)r]);
            return false;
        }
    if (options.sharding == Options::SHARD_SAMPLES && samples < G) {
        std::printf("Renderer: %d samples per pixel cannot be split over %d GPUs\n", samples, G);
        return false;
    }
    std::vector<rt_comm*> comms((size_t)G, nullptr);
    if (rt_comm_create_local(options.devices.data(), G, comms.data()) != RT_OK) {
        std::printf("Renderer: %s\n", rt_last_error());
        return false;
    }
    const size_t n = (size_t)width * height * 3;
    std::vector<void*> frames((size_t)G, nullptr);
    std::vector<rt_stats> stats((size_t)G);
    std::vector<std::string> errors((size_t)G);
    const rt_camera c = cam.flatten();
    // Every frame is allocated HERE, before any rank starts: a rank that left early (no frame) would never reach rt_reduce, and
    // the others would wait for it in the collective for ever.
    for (int r = 0; r < G; r++)
        if (rt_buffer_alloc(options.devices[(size_t)r], n * sizeof(float), &frames[(size_t)r]) != RT_OK) {
            std::printf("Renderer: rank %d (device %d): %s\n", r, options.devices[(size_t)r], rt_last_error());
            for (int k = 0; k < G; k++) {
                rt_buffer_free(options.devices[(size_t)k], frames[(size_t)k]);
                rt_comm_destroy(comms[(size_t)k]);
            }
            return false;
        }
    auto rank_main = [&](int r) {
        auto failed = [&]() { errors[(size_t)r] = rt_last_error(); };
        rt_options o{};
        o.width = width;
        o.height = height;
        o.spp_total = samples;
        o.max_depth = options.max_depth;
        o.x0 = 0;
        o.y0 = 0;
        o.x1 = width;
        o.y1 = height;
        o.seed = options.seed;
        o.spp_per_pass = options.spp_per_pass;
        o.flags = RT_RENDER_CLEAR | RT_RENDER_DEVICE_OUT;
        stats[(size_t)r] = rt_stats{};
        int rc;
        if (options.sharding == Options::SHARD_TILES) {
            o.spp_begin = 0;
            o.spp_count = samples;
            const std::vector<int32_t> tiles = shard_tiles(r, G, width, height, options.gpu_tile_size > 0 ? options.gpu_tile_size : 64);
            if (tiles.empty()) { // more ranks than tiles: this rank contributes a zero frame
                rc = rt_buffer_zero(options.devices[(size_t)r], frames[(size_t)r], n * sizeof(float));
            } else {
                rc = rt_render_tiles(scene.replica((size_t)r), &c, &o, tiles.data(), (int32_t)(tiles.size() / 4), (float*)frames[(size_t)r], &stats[(size_t)r]);
            }
        } else {
            int begin, count;
            shard_sample_window(r, G, samples, &begin, &count);
            o.spp_begin = begin;
            o.spp_count = count;
            rc = rt_render(scene.replica((size_t)r), &c, &o, (float*)frames[(size_t)r], &stats[(size_t)r]);
        }
        if (rc != RT_OK) failed();
        // every rank enters the collective, also one whose render failed (its frame then holds what it holds; the error is reported)
        if (rt_reduce(comms[(size_t)r], (float*)frames[(size_t)r], n, 0) != RT_OK && errors[(size_t)r].empty()) failed();
    };
    std::vector<std::thread> threads;
    for (int r = 1; r < G; r++) threads.emplace_back(rank_main, r);
    rank_main(0);
    for (std::thread& t : threads) t.join();
    bool ok = true;
    for (int r = 0; r < G; r++)
        if (!errors[(size_t)r].empty()) {
            std::printf("Renderer: rank %d (device %d): %s\n", r, options.devices[(size_t)r], errors[(size_t)r].c_str());
            ok = false;
        }
    if (ok && rt_buffer_download(options.devices[0], frames[0], radiance_sum.data(), n * sizeof(float)) != RT_OK) {
        std::printf("Renderer: %s\n", rt_last_error());
        ok = false;
    }
    for (int r = 0; r < G; r++) {
        rt_buffer_free(options.devices[(size_t)r], frames[(size_t)r]);
        rt_comm_destroy(comms[(size_t)r]);
    }
    last_stats = rt_stats{};
    for (const rt_stats& st : stats) { // work adds up over the GPUs, time is the slowest GPU's
        last_stats.rays_closest += st.rays_closest;
        last_stats.rays_shadow += st.rays_shadow;
        last_stats.paths += st.paths;
        last_stats.launches_closest += st.launches_closest;
        last_stats.launches_shadow += st.launches_shadow;
        last_stats.ms_total = std::max(last_stats.ms_total, st.ms_total);
    }
    return ok;
}

void Renderer::render_image(const Camera& cam, const Scene& scene, const Options& options) { // main.cpp:950-1020
    initalize(options);
    if (options.devices.size() > 1) {
        if (!render_sharded(cam, scene, options)) {
            last_stats = rt_stats{};
            return;
        }
        if (rt_resolve(scene.replica(0), radiance_sum.data(), width, height, samples, options.gamma, final_buffer.data()) != RT_OK) {
            std::printf("Renderer: %s\n", rt_last_error());
            return;
        }
        std::printf("\nDONE, seconds elapsed: %g\n", last_stats.ms_total / 1000.0);
        if (write_output && write_image(options.output_name, width, height, final_buffer.data()))
            std::printf("Output written: %s\n", options.output_name.c_str());
        return;
    }
    if (!scene.gpu()) {
        std::printf("Renderer: scene is not on a GPU (call build_top_level_bvh first)\n");
        return;
    }
    rt_camera c = cam.flatten();
    rt_options o{};
    o.width = width;
    o.height = height;
    o.spp_begin = 0;
    o.spp_count = samples;
    o.spp_total = samples;
    o.max_depth = options.max_depth;
    o.x0 = 0;
    o.y0 = 0;
    o.x1 = width;
    o.y1 = height;
    o.seed = options.seed;
    o.spp_per_pass = options.spp_per_pass;
    o.flags = RT_RENDER_CLEAR;
    last_stats = rt_stats{};
    if (rt_render(scene.gpu(), &c, &o, radiance_sum.data(), &last_stats) != RT_OK) {
        std::printf("Renderer: %s\n", rt_last_error());
        return;
    }
    if (rt_resolve(scene.gpu(), radiance_sum.data(), width, height, samples, options.gamma, final_buffer.data()) != RT_OK) {
        std::printf("Renderer: %s\n", rt_last_error());
        return;
    }
    std::printf("\nDONE, seconds elapsed: %g\n", last_stats.ms_total / 1000.0);
    if (write_output && write_image(options.output_name, width, height, final_buffer.data()))
        std::printf("Output written: %s\n", options.output_name.c_str());
}

bool write_bmp(const std::string& path, int w, int h, const uint8_t* rgb) {
    FILE* fp = std::fopen(path.c_str(), "wb");
    if (!fp) return false;
    const int row = (w * 3 + 3) & ~3;
    const uint32_t size = 54 + (uint32_t)row * h;
    uint8_t hdr[54] = {'B', 'M'};
    auto put32 = [&](int at, uint32_t v) {
        for (int i = 0; i < 4; i++) hdr[at + i] = (uint8_t)(v >> (8 * i));
    };
    put32(2, size);
    put32(10, 54);
    put32(14, 40);
    put32(18, (uint32_t)w);
    put32(22, (uint32_t)h);
    hdr[26] = 1;
    hdr[28] = 24;
    put32(34, (uint32_t)row * h);
    std::fwrite(hdr, 1, 54, fp);
    std::vector<uint8_t> line(row, 0);
    for (int y = h - 1; y >= 0; y--) { // BMP rows are bottom-up; buffer row 0 is the top (main.cpp:990)
        for (int x = 0; x < w; x++) {
            line[3 * x] = rgb[((size_t)y * w + x) * 3 + 2];
            line[3 * x + 1] = rgb[((size_t)y * w + x) * 3 + 1];
            line[3 * x + 2] = rgb[((size_t)y * w + x) * 3];
        }
        std::fwrite(line.data(), 1, row, fp);
    }
    std::fclose(fp);
    return true;
}

// PNG with the image data in stored deflate blocks: no compressor needed, every decoder reads it.  Rows top-down, one
// filter byte (0 = none) before each.
bool write_png(const std::string& path, int w, int h, const uint8_t* rgb) {
    if (w <= 0 || h <= 0) return false;
    FILE* fp = std::fopen(path.c_str(), "wb");
    if (!fp) return false;
    uint32_t crc_table[256];
    for (uint32_t n = 0; n < 256; n++) {
        uint32_t c = n;
        for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        crc_table[n] = c;
    }
    auto be32 = [](uint8_t* p, uint32_t v) {
        p[0] = (uint8_t)(v >> 24);
        p[1] = (uint8_t)(v >> 16);
        p[2] = (uint8_t)(v >> 8);
        p[3] = (uint8_t)v;
    };
    auto chunk = [&](const char* type, const std::vector<uint8_t>& data) {
        uint8_t head[8];
        be32(head, (uint32_t)data.size());
        std::memcpy(head + 4, type, 4);
        uint32_t c = 0xFFFFFFFFu;
        for (int i = 4; i < 8; i++) c = crc_table[(c ^ head[i]) & 0xFFu] ^ (c >> 8);
        for (uint8_t b : data) c = crc_table[(c ^ b) & 0xFFu] ^ (c >> 8);
        uint8_t tail[4];
        be32(tail, c ^ 0xFFFFFFFFu);
        std::fwrite(head, 1, 8, fp);
        if (!data.empty()) std::fwrite(data.data(), 1, data.size(), fp);
        std::fwrite(tail, 1, 4, fp);
    };
    const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::fwrite(sig, 1, 8, fp);
    std::vector<uint8_t> ihdr(13, 0);
    be32(&ihdr[0], (uint32_t)w);
    be32(&ihdr[4], (uint32_t)h);
    ihdr[8] = 8; // bits per channel
    ihdr[9] = 2; // truecolour
    chunk("IHDR", ihdr);
    const size_t row = (size_t)w * 3 + 1, raw = row * (size_t)h;
    std::vector<uint8_t> z;
    z.reserve(raw + raw / 65535 * 5 + 16);
    z.push_back(0x78); // zlib header: deflate, 32 KiB window, no preset dictionary
    z.push_back(0x01);
    uint32_t a = 1, b = 0; // Adler-32 of the raw stream
    std::vector<uint8_t> line(row, 0);
    std::vector<uint8_t> stream;
    stream.reserve(raw);
    for (int y = 0; y < h; y++) {
        std::memcpy(&line[1], rgb + (size_t)y * w * 3, (size_t)w * 3);
        stream.insert(stream.end(), line.begin(), line.end());
    }
    for (uint8_t v : stream) {
        a = (a + v) % 65521u;
        b = (b + a) % 65521u;
    }
    for (size_t at = 0; at < raw;) {
        const size_t n = std::min<size_t>(65535, raw - at);
        z.push_back(at + n == raw ? 1 : 0); // BFINAL, BTYPE = 00 (stored)
        z.push_back((uint8_t)(n & 0xFF));
        z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xFF));
        z.push_back((uint8_t)((~n >> 8) & 0xFF));
        z.insert(z.end(), stream.begin() + at, stream.begin() + at + n);
        at += n;
    }
    uint8_t adler[4];
    be32(adler, (b << 16) | a);
    z.insert(z.end(), adler, adler + 4);
    chunk("IDAT", z);
    chunk("IEND", {});
    const bool ok = !std::ferror(fp);
    std::fclose(fp);
    return ok;
}

bool write_image(const std::string& path, int w, int h, const uint8_t* rgb) {
    std::string ext = path.size() >= 4 ? path.substr(path.size() - 4) : "";
    for (char& c : ext) c = (char)std::tolower((unsigned char)c);
    return ext == ".png" ? write_png(path, w, h, rgb) : write_bmp(path, w, h, rgb);
}

// -------------------------------------------------------------- materials
Material* material_from_record(const rt_material& r) {
    vec3 c0(r.c0[0], r.c0[1], r.c0[2]), c1(r.c1[0], r.c1[1], r.c1[2]);
    switch (r.type) {
    case RT_MAT_MATTE:
        if (r.s[0] != 0.f) return new MatteMaterial(CheckeredTexture::from_record(c0, c1, r.s[0]));
        return new MatteMaterial(new ConstantTexture(c0));
    case RT_MAT_MICROFACET_GGX: return new Microfacet(nullptr, r.s[0], r.s[1]);
    case RT_MAT_MICROFACET_BECKMANN: return new BeckmannMicrofacet(nullptr, r.s[0], r.s[1]);
    case RT_MAT_EMISSIVE: return new EmissiveMaterial(c0);
    case RT_MAT_DISNEY_DIFFUSE: return new DisneyDiffuse(c0, r.s[0], r.s[1]);
    case RT_MAT_DISNEY_METAL: return new DisneyMetal(c0, r.s[0], r.s[1]);
    case RT_MAT_DISNEY_CLEARCOAT: return new DisneyClearcoat(r.s[0]);
    case RT_MAT_DISNEY_GLASS: return new DisneyGlass(c0, r.s[0], r.s[1], r.s[2]);
    case RT_MAT_ROUGH_DIELECTRIC: return new RoughDielectric(r.s[0], r.s[1], c0, c1);
    case RT_MAT_DISNEY_SHEEN: return new DisneySheen(c0, r.s[0]);
    case RT_MAT_DISNEY_UBER: {
        DisneyUber* u = new DisneyUber;
        u->base_color = c0;
        u->specular_transmission = r.s[0]; u->metallic = r.s[1]; u->subsurface = r.s[2]; u->specular = r.s[3];
        u->roughness = r.s[4]; u->specular_tint = r.s[5]; u->anisotropic = r.s[6]; u->sheen = r.s[7];
        u->sheen_tint = r.s[8]; u->clearcoat = r.s[9]; u->clearcoat_gloss = r.s[10]; u->ior = r.s[11];
        return u;
    }
    case RT_MAT_GLASS: return new GlassMaterial(r.s[0]);
    default: return nullptr;
    }
}

} // namespace rt
