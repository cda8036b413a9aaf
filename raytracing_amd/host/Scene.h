// Scene.h — host mirror of the reference's scene-side API (target design):
// Camera (Scene.h:10-81), Scene (Scene.h:103-153), Intersection (Def.h:11-40),
// Filter (Filter.h), Image (Image.h), the Rectangle light Instance
// (Object.h:61-135,364-423) and Options/Renderer (main.cpp:836-1020).
// Queries and rendering are forwarded to librt_hip.so through include/rt_hip.h.
#pragma once
#include "../../include/rt_hip.h"
#include "Material.h"
#include "Mesh.h"
#include <string>
#include <vector>

namespace rt {

struct Intersection { // Def.h:11-40
    vec3 point, normal, w0;
    float t = 0;
    bool front_face = true;
    float u = 0, v = 0;
    int index = 0;
    const Material* material = nullptr;
};

class Filter { // Filter.h:5-10
public:
    virtual ~Filter() {}
    virtual vec2 sample(float dx, float dy) const = 0;
    virtual void describe(int32_t* kind, float* param) const = 0;
};
class BoxFilter : public Filter { // Filter.h:12-27
public:
    explicit BoxFilter(float width_) : width(width_) {}
    vec2 sample(float dx, float dy) const override;
    void describe(int32_t* kind, float* param) const override { *kind = RT_FILTER_BOX; *param = width; }
    float width;
};
class GaussianFilter : public Filter { // Filter.h:28-42
public:
    explicit GaussianFilter(float stddev_) : stddev(stddev_) {}
    vec2 sample(float dx, float dy) const override;
    void describe(int32_t* kind, float* param) const override { *kind = RT_FILTER_GAUSSIAN; *param = stddev; }
    float stddev;
};

class Camera { // Scene.h:10-81
public:
    Camera() {}
    Camera(const mat4& view_matrix, float fov, int width, int height, Filter* filter = nullptr);
    ~Camera() { delete filter; }
    Camera(const Camera&) = delete;
    Camera& operator=(const Camera&) = delete;
    Camera& operator=(Camera&& other);
    Ray get_ray(float raster_x, float raster_y) const;
    rt_camera flatten() const;

    Filter* filter = nullptr;
    vec3 camera_pos;
    mat4 raster_to_world;
};

class Image { // Image.h:7-69 (float RGB only on this path)
public:
    enum class Format { None, RGBA_32f, RGB_32f, RGBA_u8, RGB_u8 };
    vec3 get(int x, int y) const;
    int width = 0, height = 0;
    std::vector<char> imageData;
    Format internalFormat = Format::None;
};
using HDRImage = Image;
HDRImage* make_uniform_environment(int width, int height, vec3 radiance);
HDRImage* load_hdr(const std::string& filename); // Image.cpp:6-30: Radiance RGBE -> RGB_32f, nullptr + message on failure

// Geometry (Object.h:9-24) and its subclasses.  intersect() runs on the GPU (rt_device.h: prim_intersect,
// walk_two_level); the host classes hold the parameters, the object bounds the TLAS is built from, and describe
// themselves as an rt_instance.
class Geometry {
public:
    virtual ~Geometry() {}
    virtual Bounds object_bounds() const { return Bounds(); }
    virtual float area() const { return 0.f; }
    virtual void describe(rt_instance* out) const = 0; // geometry tag + parameters
};
class Sphere : public Geometry { // Object.h:26-60
public:
    explicit Sphere(float radius_) : radius(radius_) {}
    Bounds object_bounds() const override { return Bounds(vec3(-radius), vec3(radius)); }
    void describe(rt_instance* out) const override;
    float radius;
};
class Rectangle : public Geometry { // Object.h:61-135
public:
    Rectangle(vec3 normal, vec3 tangent, float u_length, float v_length);
    Rectangle(vec3 normal, float u_length, float v_length);
    Bounds object_bounds() const override; // Object.h:81-92
    float area() const override { return (2 * u) * (2 * v); }
    void describe(rt_instance* out) const override;
    vec3 N, T, B;
    float u, v; // half side lengths
};
// The reference leaves Disk, Box and Cylinder at Geometry's empty default bounds (a TLAS over them is meaningless: the
// empty box becomes NaN under transform_bounds).  Defined here: the box that encloses the primitive.
class Disk : public Geometry { // Object.h:137-168
public:
    Disk(vec3 normal_, float outer_radius_, float inner_radius_ = -1.f) : normal(normal_), outer_radius(outer_radius_), inner_radius(inner_radius_) {}
    Bounds object_bounds() const override { return Bounds(vec3(-outer_radius), vec3(outer_radius)); }
    void describe(rt_instance* out) const override;
    vec3 normal;
    float outer_radius, inner_radius;
};
class Box : public Geometry { // Object.h:170-209
public:
    explicit Box(vec3 box_size) : bmin(box_size / -2.f), bmax(box_size / 2) {}
    Bounds object_bounds() const override { return Bounds(bmin, bmax); }
    void describe(rt_instance* out) const override;
    vec3 bmin, bmax;
};
class Cylinder : public Geometry { // Object.h:211-243
public:
    Cylinder(float radius_, float height) : radius(radius_), ymin(height / -2.f), ymax(height / 2.f) {}
    Bounds object_bounds() const override { return Bounds(vec3(-radius, ymin, -radius), vec3(radius, ymax, radius)); }
    void describe(rt_instance* out) const override;
    float radius, ymin, ymax;
};
class TriangleMesh : public Geometry { // Object.h:266-362, Object.cpp:11-29: borrows the mesh, owns its BVH
public:
    // max_per_node: BVH::build's parameter; the reference passes 1 (Object.cpp:21)
    explicit TriangleMesh(const Mesh* mesh, int max_per_node = 1);
    Bounds object_bounds() const override { return mesh_bounds; }
    void describe(rt_instance* out) const override;
    const Mesh* mesh;
    BVH bvh;
    Bounds mesh_bounds;
};
class Instance { // Object.h:364-423
public:
    Instance(Geometry* geo, Material* mat, vec3 offset) : geometry(geo), material(mat), transform(translate(mat4(1.f), offset)) {}
    Instance(Geometry* geo, Material* mat, mat4 m) : geometry(geo), material(mat), transform(m) {}
    void free_data() {
        delete geometry;
        delete material;
    }
    float get_area() const { return geometry->area(); }
    Bounds get_world_bounds() const { return transform_bounds(geometry->object_bounds(), transform); } // Object.h:412-416
    const Material* get_material() const { return material; }
    Material* get_material_ptr() const { return material; } // to place the same material in a second Instance (a light in both lists)
    const Geometry* get_geometry() const { return geometry; }
    const Transform& get_transform() const { return transform; }

private:
    friend class Scene; // ~Scene frees every geometry and material exactly once
    Geometry* geometry = nullptr;
    Material* material = nullptr;
    Transform transform;
};

// How glTF pbrMetallicRoughness materials become reference materials
// (Scene::load_gltf_materials is declared but unwritten: Scene.cpp:174).
enum GltfMaterialMode {
    GLTF_MATERIALS_UBER = 0,  // DisneyUber{base, metallic, roughness, transmission}
    GLTF_MATERIALS_MATTE = 1, // MatteMaterial(baseColor)
    GLTF_MATERIALS_GGX = 2    // roughness>0 && !transmission -> Microfacet (even material index) / BeckmannMicrofacet (odd)
};

// The flattened, GPU-ready arrays (what rt_scene_create uploads).
struct FlatScene {
    int32_t root_ref = -1;
    std::vector<rt_bvh_pair> pairs;
    std::vector<rt_tri> tris;
    std::vector<rt_tri_normals> normals;
    std::vector<rt_material> materials;
    std::vector<rt_light> lights;
    std::vector<rt_emissive_tri> emissive;
    std::vector<int32_t> tri_light;       // per ORIGINAL triangle: light index or -1
    std::vector<int32_t> emissive_orig;   // emissive[i].slot's original triangle index
    std::vector<float> env_cdf;
    float env_total = 0;
    // legacy two-level scenes (Scene::instances not empty): pairs/tris/normals above hold every TriangleMesh's own BVH
    // in model space, one after the other; these hold the TLAS
    int32_t tlas_root_ref = -1;
    std::vector<rt_bvh_pair> tlas_pairs;
    std::vector<rt_instance> instances;
};

class Scene { // Scene.h:103-153
public:
    Scene() {}
    ~Scene();
    Scene(const Scene&) = delete;
    Scene& operator=(const Scene&) = delete;

    bool load_from_gltf(const std::string& gltfPath);
    bool closest_hit(Ray r, Intersection* res) const;
    bool any_hit(Ray r) const;
    // Scene.cpp:6-15 in the legacy design built the TLAS.  In the target design
    // (one merged mesh) it builds the mesh BVH, derives the light tables,
    // flattens everything and uploads it to the GPU once.
    void build_top_level_bvh();

    // Adds the Instance(Rectangle, EmissiveMaterial, transform) pair the
    // reference pushes into both `instances` and `lights` (main.cpp:801-813):
    // two emissive triangles into sceneMergedMesh + the sampled light.
    void add_rectangle_light(vec3 normal, float u_length, float v_length, vec3 emit, const mat4& transform);
    void set_environment(HDRImage* img, float degrees_offset = 0.f); // takes ownership
    // materials[index] = m (takes ownership): every Instance in `instances` and `lights` that holds the old material is
    // moved to the new one before the old one is freed; false (and m freed) when index is out of range
    bool replace_material(size_t index, Material* m);
    // Instance(TriangleMesh, material, transform) of the legacy design (Object.h:364-423, main.cpp:456-470), flattened
    // at scene set-up: the mesh's triangles enter sceneMergedMesh in world space — positions through
    // Transform::to_world_point, normals through normalize(to_world_normal) exactly as Instance::intersect reports
    // them (Object.h:388-391) — so the GPU walks ONE tree (BASELINE config 5: "instanced mesh, flattened at upload").
    // material_override >= 0 replaces every triangle's materialIndex.
    void add_instance(const Mesh& mesh, const mat4& model_to_world, int material_override = -1);

    // stages of build_top_level_bvh, exposed for tests and for hosts without a GPU
    void build_acceleration(int max_per_leaf = 1, PartitionStrategy strat = BVH_SAH);
    void flatten();
    bool upload(int device);
    rt_scene* gpu() const { return gpu_scene; }
    // Multi-GPU: one copy of the flattened scene per listed device (a device may be listed twice: rehearsal on one GPU).
    // Renderer::render_image uses replica i as rank i when Options::devices names the same list.
    bool upload_replicas(const std::vector<int>& devices);
    size_t replica_count() const { return replicas.size(); }
    rt_scene* replica(size_t i) const { return replicas[i].second; }
    int replica_device(size_t i) const { return replicas[i].first; }
    const FlatScene& flat() const { return flat_scene; }

    bool hasCamera = false;
    Camera sceneCamera;
    mat4 sceneCameraView = mat4(1.f); // view matrix + vertical fov the camera was built from,
    float sceneCameraFov = 39.6f;     // so a host can rebuild it at another resolution
    Mesh sceneMergedMesh;
    std::vector<Material*> materials;
    std::vector<Image*> images;
    std::vector<Instance> lights;
    vec3 background_color = vec3(0);
    // The legacy design (Scene.h:137-138, commented out at HEAD; Scene.cpp:6-77 still uses them): when `instances` is
    // not empty the scene is two-level — build_top_level_bvh builds `tlas` over the instances' world bounds
    // (Scene.cpp:6-15), every TriangleMesh keeps its own BVH, and rays are moved into model space per instance on the
    // GPU.  sceneMergedMesh is then unused; an Instance's material must be one of `materials` (the scene frees it).
    // A light is an Instance(Rectangle, EmissiveMaterial) pushed into BOTH lists, as main.cpp:801-813 does.
    std::vector<Instance> instances;
    BVH tlas;

    HDRImage* environment = nullptr;
    float environment_offset = 0.f; // degrees / 360 (main.cpp:94)
    GltfMaterialMode gltf_material_mode = GLTF_MATERIALS_UBER;
    float gltf_emissive_scale = 1.f;
    int device = 0;

private:
    FlatScene flat_scene;
    rt_scene* gpu_scene = nullptr;
    std::vector<std::pair<int, rt_scene*>> replicas;
    rt_scene* create_on(int dev) const;
    std::vector<int> light_first_tri; // per rect light: first of its two triangles
    void flatten_two_level();
};

struct Options { // main.cpp:836-850
    std::string output_name = "Output.bmp";
    int num_threads = 0; // unused on the GPU path
    int width = 512, height = 512;
    int samples_per_pixel = 1;
    int max_depth = 4;
    int direct_samples = 1;
    int tile_size = 16; // unused on the GPU path
    float gamma = 2.2f;
    unsigned seed = 0;
    int spp_per_pass = 0;
    // Multi-GPU (SURVEY 8 e; no counterpart in the reference, whose parallelism is std::thread over 16x16 tiles): the GPUs to
    // render on, one host thread each; how the frame is cut; the GPU tile edge.  Empty devices = the scene's own GPU.
    std::vector<int> devices;
    enum Sharding { SHARD_SAMPLES = 0, SHARD_TILES = 1 };
    int sharding = SHARD_SAMPLES; // SAMPLES: rank r renders its window of the ONE spp-sample set; TILES: interleaved tiles, t -> rank t mod G
    int gpu_tile_size = 64;
};

// The two partitions of a frame over G ranks (also what bench.py and raytracing_amd/sharding.py compute).
void shard_sample_window(int rank, int world, int spp, int* begin, int* count); // disjoint windows covering [0, spp)
std::vector<int32_t> shard_tiles(int rank, int world, int width, int height, int tile); // x0,y0,x1,y1 per tile, tile t -> rank t % world

class Renderer { // main.cpp:853-903,950-1020
public:
    void initalize(const Options& options);
    // Drop-in for the reference's render call: same signature, the thread pool
    // replaced by rt_render + rt_resolve on the scene's GPU.
    void render_image(const Camera& cam, const Scene& scene, const Options& options);

    const std::vector<uint8_t>& pixels() const { return final_buffer; }
    const std::vector<float>& radiance() const { return radiance_sum; }
    rt_stats last_stats{};
    bool write_output = true;

private:
    bool render_sharded(const Camera& cam, const Scene& scene, const Options& options);
    int width = 0, height = 0, samples = 0;
    std::vector<uint8_t> final_buffer;
    std::vector<float> radiance_sum;
};

bool write_bmp(const std::string& path, int w, int h, const uint8_t* rgb);
bool write_png(const std::string& path, int w, int h, const uint8_t* rgb); // 8-bit RGB, stored (uncompressed) deflate blocks
bool write_image(const std::string& path, int w, int h, const uint8_t* rgb); // by extension: .png, else BMP (main.cpp:1015 writes BMP)

} // namespace rt
