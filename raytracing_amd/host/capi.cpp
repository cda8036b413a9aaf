// capi.cpp — flat C interface over the host Scene/Mesh/BVH/Renderer classes
// (librt_host.so), so scripts and tests can drive the same host code a C++
// application would link.  Not the GPU boundary: that is include/rt_hip.h.
#include "Scene.h"

#include <cstring>

using namespace rt;

struct rth_scene {
    Scene scene;
    // two-level scenes: the meshes TriangleMesh geometries borrow, and those geometries (owned by the scene's instances
    // once placed; a geometry never placed is freed here)
    std::vector<Mesh*> meshes;
    std::vector<TriangleMesh*> mesh_geometry;
    std::vector<bool> mesh_placed;
    ~rth_scene() {
        for (size_t i = 0; i < mesh_geometry.size(); i++)
            if (!mesh_placed[i]) delete mesh_geometry[i];
    }
    // (the Mesh objects outlive the Scene's destructor body: freed by free_meshes() after the scene)
};

extern "C" {

// BVH::build (BVH.cpp:131-149): bounds6 = n x (bmin, bmax); nodes_out holds up
// to 2n-1 32-byte BVHNode records, indices_out n ints.  Returns the node count.
int32_t rth_bvh_build(const float* bounds6, int32_t n, int32_t max_per_node, int32_t strategy, void* nodes_out, int32_t* indices_out) {
    std::vector<Bounds> b((size_t)n);
    for (int i = 0; i < n; i++) {
        b[i].bmin = vec3(bounds6[6 * i], bounds6[6 * i + 1], bounds6[6 * i + 2]);
        b[i].bmax = vec3(bounds6[6 * i + 3], bounds6[6 * i + 4], bounds6[6 * i + 5]);
    }
    BVH bvh = BVH::build(b, max_per_node, strategy == 0 ? BVH_MIDDLE : BVH_SAH);
    std::memcpy(nodes_out, bvh.nodes.data(), bvh.nodes.size() * sizeof(BVHNode));
    std::memcpy(indices_out, bvh.indicies.data(), bvh.indicies.size() * sizeof(int));
    return (int32_t)bvh.nodes.size();
}

rth_scene* rth_scene_new(void) { return new rth_scene; }
void rth_scene_free(rth_scene* s) {
    std::vector<Mesh*> meshes = s->meshes;
    delete s;
    for (Mesh* m : meshes) delete m;
}

// ---- legacy two-level scenes: Scene::instances (Object.h:364-423) ------------------------------------------------
// TriangleMesh(import_mesh(...)) of the scene scripts (main.cpp:456,647): builds the mesh's own BVH once; returns a
// handle to place it with rth_scene_add_mesh_instance any number of times.
int32_t rth_scene_new_triangle_mesh(rth_scene* s, const void* verts, int64_t n_verts, const void* tris, int64_t n_tris, int32_t max_per_leaf) {
    Mesh* m = new Mesh;
    const Vertex* v = (const Vertex*)verts;
    const MeshTriangle* t = (const MeshTriangle*)tris;
    m->verts.assign(v, v + n_verts);
    m->triangles.assign(t, t + n_tris);
    s->meshes.push_back(m);
    s->mesh_geometry.push_back(new TriangleMesh(m, max_per_leaf));
    s->mesh_placed.push_back(false);
    return (int32_t)s->mesh_geometry.size() - 1;
}
// instances.push_back(Instance(mesh geometry, materials[material], transform)); returns the instance index
int32_t rth_scene_add_mesh_instance(rth_scene* s, int32_t mesh, int32_t material, const float* transform16) {
    if (mesh < 0 || (size_t)mesh >= s->mesh_geometry.size() || material < 0 || (size_t)material >= s->scene.materials.size()) return -1;
    s->scene.instances.push_back(Instance(s->mesh_geometry[mesh], s->scene.materials[material], mat4::from_columns(transform16)));
    s->mesh_placed[mesh] = true;
    return (int32_t)s->scene.instances.size() - 1;
}
// instances.push_back(Instance(new <Geometry>(params), materials[material], transform)).  params by kind (RT_GEO_*):
// SPHERE radius | RECTANGLE normal[3], u_length, v_length | DISK normal[3], outer_radius, inner_radius | BOX size[3] |
// CYLINDER radius, height.  is_light: a second Instance of the same rectangle goes into Scene::lights (main.cpp:801-813).
int32_t rth_scene_add_primitive_instance(rth_scene* s, int32_t kind, const float* p, int32_t material, const float* transform16, int32_t is_light) {
    if (material < 0 || (size_t)material >= s->scene.materials.size()) return -1;
    auto make = [&]() -> Geometry* {
        switch (kind) {
        case RT_GEO_SPHERE: return new Sphere(p[0]);
        case RT_GEO_RECTANGLE: return new Rectangle(vec3(p[0], p[1], p[2]), p[3], p[4]);
        case RT_GEO_DISK: return new Disk(vec3(p[0], p[1], p[2]), p[3], p[4]);
        case RT_GEO_BOX: return new Box(vec3(p[0], p[1], p[2]));
        case RT_GEO_CYLINDER: return new Cylinder(p[0], p[1]);
        default: return nullptr;
        }
    };
    Geometry* g = make();
    if (!g) return -1;
    if (is_light && kind != RT_GEO_RECTANGLE) {
        delete g;
        return -1;
    }
    const mat4 m = mat4::from_columns(transform16);
    s->scene.instances.push_back(Instance(g, s->scene.materials[material], m));
    if (is_light) s->scene.lights.push_back(Instance(make(), s->scene.materials[material], m));
    return (int32_t)s->scene.instances.size() - 1;
}
// which: 0 verts, 1 triangles, 2 BVH nodes, 3 BVH indices of mesh handle `mesh`
const void* rth_scene_mesh_array(rth_scene* s, int32_t mesh, int32_t which, int64_t* count) {
    *count = 0;
    if (mesh < 0 || (size_t)mesh >= s->mesh_geometry.size()) return nullptr;
    const Mesh* m = s->meshes[mesh];
    const BVH& b = s->mesh_geometry[mesh]->bvh;
    switch (which) {
    case 0: *count = (int64_t)m->verts.size(); return m->verts.data();
    case 1: *count = (int64_t)m->triangles.size(); return m->triangles.data();
    case 2: *count = (int64_t)b.nodes.size(); return b.nodes.data();
    case 3: *count = (int64_t)b.indicies.size(); return b.indicies.data();
    default: return nullptr;
    }
}
// Instance::get_world_bounds (Object.h:412-416) of instance i -> out6 = (bmin, bmax)
int32_t rth_scene_instance_bounds(rth_scene* s, int32_t i, float* out6) {
    if (i < 0 || (size_t)i >= s->scene.instances.size()) return 0;
    const Bounds b = s->scene.instances[i].get_world_bounds();
    out6[0] = b.bmin.x; out6[1] = b.bmin.y; out6[2] = b.bmin.z;
    out6[3] = b.bmax.x; out6[4] = b.bmax.y; out6[5] = b.bmax.z;
    return 1;
}

// material_mode: GltfMaterialMode; emissive_scale multiplies emissiveFactor.
int32_t rth_scene_load_gltf(rth_scene* s, const char* path, int32_t material_mode, float emissive_scale) {
    s->scene.gltf_material_mode = (GltfMaterialMode)material_mode;
    s->scene.gltf_emissive_scale = emissive_scale;
    return s->scene.load_from_gltf(path) ? 1 : 0;
}

// Appends geometry to sceneMergedMesh.  verts = n x Vertex (32 B), tris = n x MeshTriangle (16 B),
// triangle vertex indices are relative to this call's first vertex.
void rth_scene_add_mesh(rth_scene* s, const void* verts, int64_t n_verts, const void* tris, int64_t n_tris) {
    Mesh& m = s->scene.sceneMergedMesh;
    const uint32_t base = (uint32_t)m.verts.size();
    const Vertex* v = (const Vertex*)verts;
    const MeshTriangle* t = (const MeshTriangle*)tris;
    m.verts.insert(m.verts.end(), v, v + n_verts);
    for (int64_t i = 0; i < n_tris; i++) {
        MeshTriangle tri = t[i];
        for (int c = 0; c < 3; c++) tri.indicies[c] += base;
        m.triangles.push_back(tri);
    }
}

// Scene::add_instance: the mesh (same layouts as rth_scene_add_mesh) placed with a column-major model-to-world matrix.
void rth_scene_add_instance(rth_scene* s, const void* verts, int64_t n_verts, const void* tris, int64_t n_tris, const float* transform16,
                            int32_t material_override) {
    Mesh m;
    const Vertex* v = (const Vertex*)verts;
    const MeshTriangle* t = (const MeshTriangle*)tris;
    m.verts.assign(v, v + n_verts);
    m.triangles.assign(t, t + n_tris);
    s->scene.add_instance(m, mat4::from_columns(transform16), material_override);
}

int32_t rth_scene_add_material(rth_scene* s, const rt_material* rec) {
    Material* m = material_from_record(*rec);
    if (!m) return -1;
    s->scene.materials.push_back(m);
    return (int32_t)s->scene.materials.size() - 1;
}

int32_t rth_scene_set_material(rth_scene* s, int32_t index, const rt_material* rec) {
    if (index < 0 || (size_t)index >= s->scene.materials.size()) return 0;
    Material* m = material_from_record(*rec);
    if (!m) return 0;
    return s->scene.replace_material((size_t)index, m) ? 1 : 0; // instances placed with the old material follow it
}

void rth_scene_add_rect_light(rth_scene* s, const float* normal3, float u_length, float v_length, const float* emit3, const float* transform16) {
    s->scene.add_rectangle_light(vec3(normal3[0], normal3[1], normal3[2]), u_length, v_length, vec3(emit3[0], emit3[1], emit3[2]),
                                 mat4::from_columns(transform16));
}

// load_hdr + EnviormentLight set-up as main.cpp:1028-1036 does: 1 = loaded, 0 = the file could not be read.
int32_t rth_scene_load_env_hdr(rth_scene* s, const char* path, float degrees_offset) {
    HDRImage* img = load_hdr(path);
    if (!img) return 0;
    s->scene.set_environment(img, degrees_offset);
    return 1;
}

// the environment image as the scene holds it: dims2 = {w, h}; rgb (w*h*3 floats) may be NULL to query the size
int32_t rth_scene_get_env(rth_scene* s, int32_t* dims2, float* rgb) {
    const HDRImage* img = s->scene.environment;
    if (!img) return 0;
    dims2[0] = img->width;
    dims2[1] = img->height;
    if (rgb) std::memcpy(rgb, img->imageData.data(), (size_t)img->width * img->height * 3 * sizeof(float));
    return 1;
}

// RGB_32f lat-long environment (copied).  rgb == NULL removes it.
void rth_scene_set_env(rth_scene* s, int32_t w, int32_t h, const float* rgb, float degrees_offset) {
    if (!rgb) {
        s->scene.set_environment(nullptr, 0);
        return;
    }
    HDRImage* img = new HDRImage;
    img->width = w;
    img->height = h;
    img->internalFormat = Image::Format::RGB_32f;
    img->imageData.resize((size_t)w * h * 3 * sizeof(float));
    std::memcpy(img->imageData.data(), rgb, img->imageData.size());
    s->scene.set_environment(img, degrees_offset);
}

// Mesh::build + light tables + flattening (no GPU needed).
void rth_scene_build(rth_scene* s, int32_t max_per_leaf, int32_t strategy) {
    s->scene.build_acceleration(max_per_leaf, strategy == 0 ? BVH_MIDDLE : BVH_SAH);
    s->scene.flatten();
}

// Installs a previously built BVH (cache) instead of building; then flattens.
int32_t rth_scene_set_bvh(rth_scene* s, const void* nodes, int64_t n_nodes, const int32_t* indices, int64_t n_indices) {
    Mesh& m = s->scene.sceneMergedMesh;
    if ((size_t)n_indices != m.triangles.size()) return 0;
    const BVHNode* nd = (const BVHNode*)nodes;
    m.accelStructure.nodes.assign(nd, nd + n_nodes);
    m.accelStructure.indicies.assign(indices, indices + n_indices);
    s->scene.flatten();
    return 1;
}

int32_t rth_scene_upload(rth_scene* s, int32_t device) { return s->scene.upload(device) ? 1 : 0; }
rt_scene* rth_scene_gpu(rth_scene* s) { return s->scene.gpu(); }

enum {
    RTH_VERTS = 0, RTH_TRIS, RTH_NODES, RTH_INDICES, RTH_MATERIALS, RTH_LIGHTS, RTH_EMISSIVE, RTH_TRI_LIGHT,
    RTH_ENV_RGB, RTH_ENV_CDF, RTH_FLAT_PAIRS, RTH_FLAT_TRIS, RTH_FLAT_NORMALS, RTH_EMISSIVE_ORIG,
    RTH_TLAS_NODES, RTH_TLAS_INDICES, RTH_TLAS_PAIRS, RTH_INSTANCES
};

// Read-only views of the host arrays; *count = number of elements.
const void* rth_scene_array(rth_scene* s, int32_t which, int64_t* count) {
    const Scene& sc = s->scene;
    const Mesh& m = sc.sceneMergedMesh;
    const FlatScene& f = sc.flat();
    switch (which) {
    case RTH_VERTS: *count = (int64_t)m.verts.size(); return m.verts.data();
    case RTH_TRIS: *count = (int64_t)m.triangles.size(); return m.triangles.data();
    case RTH_NODES: *count = (int64_t)m.accelStructure.nodes.size(); return m.accelStructure.nodes.data();
    case RTH_INDICES: *count = (int64_t)m.accelStructure.indicies.size(); return m.accelStructure.indicies.data();
    case RTH_MATERIALS: *count = (int64_t)f.materials.size(); return f.materials.data();
    case RTH_LIGHTS: *count = (int64_t)f.lights.size(); return f.lights.data();
    case RTH_EMISSIVE: *count = (int64_t)f.emissive.size(); return f.emissive.data();
    case RTH_TRI_LIGHT: *count = (int64_t)f.tri_light.size(); return f.tri_light.data();
    case RTH_ENV_RGB:
        if (!sc.environment) { *count = 0; return nullptr; }
        *count = (int64_t)sc.environment->width * sc.environment->height * 3;
        return sc.environment->imageData.data();
    case RTH_ENV_CDF: *count = (int64_t)f.env_cdf.size(); return f.env_cdf.data();
    case RTH_FLAT_PAIRS: *count = (int64_t)f.pairs.size(); return f.pairs.data();
    case RTH_FLAT_TRIS: *count = (int64_t)f.tris.size(); return f.tris.data();
    case RTH_FLAT_NORMALS: *count = (int64_t)f.normals.size(); return f.normals.data();
    case RTH_EMISSIVE_ORIG: *count = (int64_t)f.emissive_orig.size(); return f.emissive_orig.data();
    case RTH_TLAS_NODES: *count = (int64_t)sc.tlas.nodes.size(); return sc.tlas.nodes.data();
    case RTH_TLAS_INDICES: *count = (int64_t)sc.tlas.indicies.size(); return sc.tlas.indicies.data();
    case RTH_TLAS_PAIRS: *count = (int64_t)f.tlas_pairs.size(); return f.tlas_pairs.data();
    case RTH_INSTANCES: *count = (int64_t)f.instances.size(); return f.instances.data();
    default: *count = 0; return nullptr;
    }
}

// info[0]=env_w, [1]=env_h, [2]=root_ref, [3]=hasCamera
void rth_scene_info(rth_scene* s, int32_t* info4, float* env_total, float* env_offset, float* cam_view16, float* cam_fov) {
    const Scene& sc = s->scene;
    info4[0] = sc.environment ? sc.environment->width : 0;
    info4[1] = sc.environment ? sc.environment->height : 0;
    info4[2] = sc.flat().root_ref;
    info4[3] = sc.hasCamera ? 1 : 0;
    *env_total = sc.flat().env_total;
    *env_offset = sc.environment_offset;
    std::memcpy(cam_view16, sc.sceneCameraView.m, sizeof(float) * 16);
    *cam_fov = sc.sceneCameraFov;
}

// inverse / determinant of a column-major 4x4 (Math.h:425-542); returns the determinant
float rth_mat4_inverse(const float* m16, float* out16) {
    const mat4 a = mat4::from_columns(m16);
    const mat4 r = inverse(a);
    for (int i = 0; i < 16; i++) out16[i] = r.m[i];
    return determinant(a);
}

void rth_look_at(const float* eye3, const float* at3, const float* up3, float* view16) {
    mat4 v = look_at(vec3(eye3[0], eye3[1], eye3[2]), vec3(at3[0], at3[1], at3[2]), vec3(up3[0], up3[1], up3[2]));
    std::memcpy(view16, v.m, sizeof(v.m));
}

// Camera constructor (Scene.h:14-44); filter: RT_FILTER_*.
void rth_camera_init(const float* view16, float fov, int32_t w, int32_t h, int32_t filter, float filter_param, rt_camera* out) {
    Filter* f = filter == RT_FILTER_BOX ? (Filter*)new BoxFilter(filter_param) : (Filter*)new GaussianFilter(filter_param);
    Camera cam(mat4::from_columns(view16), fov, w, h, f);
    *out = cam.flatten();
}

// Camera::get_ray on the host (Scene.h:61-75).
void rth_camera_ray(const float* view16, float fov, int32_t w, int32_t h, float raster_x, float raster_y, rt_ray* out) {
    Camera cam(mat4::from_columns(view16), fov, w, h);
    Ray r = cam.get_ray(raster_x, raster_y);
    out->o[0] = r.pos.x; out->o[1] = r.pos.y; out->o[2] = r.pos.z;
    out->d[0] = r.dir.x; out->d[1] = r.dir.y; out->d[2] = r.dir.z;
}

// the Renderer's output writer on its own (main.cpp:1015): .png by extension, else BMP; rgb = w*h*3 u8, row 0 = top
int32_t rth_write_image(const char* path, int32_t w, int32_t h, const uint8_t* rgb) { return write_image(path, w, h, rgb) ? 1 : 0; }

// Renderer::render_image (main.cpp:950-1020) through the C++ classes.
// rgb_out = w*h*3 u8 (row 0 = top), radiance_out (optional) = w*h*3 fp32 sums.
int32_t rth_render_image(rth_scene* s, const float* view16, float fov, int32_t w, int32_t h, int32_t spp, int32_t max_depth,
                         uint32_t seed, const char* bmp_path, uint8_t* rgb_out, float* radiance_out, rt_stats* stats) {
    Camera cam(mat4::from_columns(view16), fov, w, h);
    Options o;
    o.width = w;
    o.height = h;
    o.samples_per_pixel = spp;
    o.max_depth = max_depth;
    o.seed = seed;
    Renderer r;
    r.write_output = bmp_path != nullptr;
    if (bmp_path) o.output_name = bmp_path;
    r.render_image(cam, s->scene, o);
    if (r.last_stats.paths == 0) return 0;
    if (rgb_out) std::memcpy(rgb_out, r.pixels().data(), r.pixels().size());
    if (radiance_out) std::memcpy(radiance_out, r.radiance().data(), r.radiance().size() * sizeof(float));
    if (stats) *stats = r.last_stats;
    return 1;
}

// Renderer::render_image over several GPUs (Options::devices; sharding 0 = sample windows, 1 = interleaved tiles): the scene
// is placed on the listed devices first (Scene::upload_replicas).
int32_t rth_render_image_sharded(rth_scene* s, const float* view16, float fov, int32_t w, int32_t h, int32_t spp, int32_t max_depth,
                                 uint32_t seed, const int32_t* devices, int32_t n_devices, int32_t sharding, int32_t tile_size,
                                 uint8_t* rgb_out, float* radiance_out, rt_stats* stats) {
    Camera cam(mat4::from_columns(view16), fov, w, h);
    Options o;
    o.width = w;
    o.height = h;
    o.samples_per_pixel = spp;
    o.max_depth = max_depth;
    o.seed = seed;
    o.devices.assign(devices, devices + n_devices);
    o.sharding = sharding;
    if (tile_size > 0) o.gpu_tile_size = tile_size;
    if (!s->scene.upload_replicas(o.devices)) return 0;
    Renderer r;
    r.write_output = false;
    r.render_image(cam, s->scene, o);
    if (r.last_stats.paths == 0) return 0;
    if (rgb_out) std::memcpy(rgb_out, r.pixels().data(), r.pixels().size());
    if (radiance_out) std::memcpy(radiance_out, r.radiance().data(), r.radiance().size() * sizeof(float));
    if (stats) *stats = r.last_stats;
    return 1;
}

// the two partitions (Scene.h): for scripts that shard by process (bench.py) and for the tests
void rth_shard_sample_window(int32_t rank, int32_t world, int32_t spp, int32_t* begin_count2) {
    shard_sample_window(rank, world, spp, &begin_count2[0], &begin_count2[1]);
}
int32_t rth_shard_tiles(int32_t rank, int32_t world, int32_t width, int32_t height, int32_t tile, int32_t* out_xyxy, int32_t capacity_tiles) {
    const std::vector<int32_t> t = shard_tiles(rank, world, width, height, tile);
    const int32_t n = (int32_t)(t.size() / 4);
    if (out_xyxy && n <= capacity_tiles) std::memcpy(out_xyxy, t.data(), t.size() * sizeof(int32_t));
    return n;
}

// Scene::closest_hit / Scene::any_hit for one ray (host API smoke).
int32_t rth_scene_closest_hit(rth_scene* s, const rt_ray* ray, float* t, float* point3, float* normal3, int32_t* front_face) {
    Intersection si;
    si.index = -1;
    Ray r(vec3(ray->o[0], ray->o[1], ray->o[2]), vec3(ray->d[0], ray->d[1], ray->d[2]));
    if (!s->scene.closest_hit(r, &si)) return 0;
    *t = si.t;
    point3[0] = si.point.x; point3[1] = si.point.y; point3[2] = si.point.z;
    normal3[0] = si.normal.x; normal3[1] = si.normal.y; normal3[2] = si.normal.z;
    *front_face = (si.front_face ? 1 : 0) | (si.index << 1); // bit 0 front_face, the rest Intersection::index
    return 1;
}
int32_t rth_scene_any_hit(rth_scene* s, const rt_ray* ray) {
    return s->scene.any_hit(Ray(vec3(ray->o[0], ray->o[1], ray->o[2]), vec3(ray->d[0], ray->d[1], ray->d[2]))) ? 1 : 0;
}

} // extern "C"
