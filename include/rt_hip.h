/*
 * rt_hip.h — C-ABI of the MI355X path-tracing hot path (librt_hip.so).
 *
 * The reference (charlesrw1/Raytracing) has no plugin/FFI seam: the GPU path
 * slots in behind plain C++ calls inside one executable.  Every entry point
 * below names the reference call it replaces (file:line relative to the
 * reference tree).  Plain C, plain pointers and sizes, no C++/torch types.
 *
 * Conventions
 *   - every function returns RT_OK (0) or an RT_ERR_* code; the message for
 *     the calling thread's last failure is rt_last_error().  (The reference
 *     reports errors with bool/nullptr + printf: Scene.cpp:203-206,
 *     BVH.cpp:138-141.)
 *   - one rt_scene per GPU; calls on one handle are not re-entrant; different
 *     handles may be driven from different host threads.
 *   - "host" pointers are ordinary malloc'ed memory, "device" pointers are
 *     HBM addresses on the scene's GPU (e.g. a torch tensor's data_ptr()).
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK 0
#define RT_ERR_INVALID 1   /* bad argument / inconsistent description        */
#define RT_ERR_NO_DEVICE 2 /* no usable HIP device                           */
#define RT_ERR_HIP 3       /* a HIP runtime call failed (see rt_last_error)  */
#define RT_ERR_LIMIT 4     /* a documented size limit was exceeded           */

/* ---- plain data ------------------------------------------------------- */

typedef struct rt_scene rt_scene; /* opaque: one uploaded scene on one GPU */

/* Ray: Math.h:622-633 (pos, dir), 24 B. */
typedef struct rt_ray {
    float o[3];
    float d[3];
} rt_ray;

/* Closest-hit record.  tri = index into the mesh's triangle array
 * (Mesh.h:15-19 order), -1 on a miss; (u,v) = bary_out.x/.y of
 * IntersectTriRay2 (Object.h:245-264); t as accepted at Object.h:312-315. */
typedef struct rt_hit {
    int32_t tri;
    float t;
    float u;
    float v;
} rt_hit;

/* One inner BVH node flattened as a 64-byte child-pair record: the two
 * adjacent children (BVH.cpp:122-125 allocates left, left+1) of a branch
 * node (BVH.h:15-21), so one traversal step (Object.h:322-326) reads one
 * aligned 64-byte line.
 * child ref >= 0 : index of the child's own pair record (child is a branch)
 * child ref <  0 : child is a leaf; its first triangle slot is ~ref         */
typedef struct rt_bvh_pair {
    float lmin[3];
    float lmax[3];
    float rmin[3];
    float rmax[3];
    int32_t lref;
    int32_t rref;
    int32_t reserved[2];
} rt_bvh_pair;

#define RT_TRI_LEAF_END 0x80000000u

/* One triangle in leaf order (slot k holds triangle BVH::indicies[k],
 * BVH.h:32), pre-differenced exactly as IntersectTriRay2 does
 * (edge1 = v1 - v0, edge2 = v2 - v0: Object.h:247-248), 48 B.
 * tri_id: original triangle index, bit 31 (RT_TRI_LEAF_END) set on the last
 * triangle of its leaf.  material: MeshTriangle::materialIndex (Mesh.h:18).
 * light: index into rt_scene_desc.lights or -1. */
typedef struct rt_tri {
    float v0[3];
    uint32_t tri_id;
    float e1[3];
    int32_t material;
    float e2[3];
    int32_t light;
} rt_tri;

/* Shading normals of slot k's three corners (Vertex::normal, Mesh.h:8-13;
 * interpolated at Object.h:351-353), 48 B. */
typedef struct rt_tri_normals {
    float n0[3];
    float pad0;
    float n1[3];
    float pad1;
    float n2[3];
    float pad2;
} rt_tri_normals;

/* Material table entry (Material.h / Material.cpp classes flattened to a tag
 * + parameters), 80 B. */
enum {
    RT_MAT_MATTE = 0,            /* MatteMaterial      Material.h:64-92   c0 = albedo (ConstantTexture); CheckeredTexture
                                    (Material.h:25-41): c0 = even, c1 = odd, s0 = its grid = PI / the constructor's argument (0 = constant) */
    RT_MAT_MICROFACET_GGX = 1,   /* Microfacet         Material.cpp:4-109 s0 = roughness               */
    RT_MAT_MICROFACET_BECKMANN = 2, /* (new, no reference arithmetic)     s0 = roughness               */
    RT_MAT_EMISSIVE = 3,         /* EmissiveMaterial   Material.h:274-284 c0 = emit                    */
    RT_MAT_DISNEY_DIFFUSE = 4,   /* Material.cpp:114-153  c0 albedo, s0 roughness, s1 subsurface       */
    RT_MAT_DISNEY_METAL = 5,     /* Material.cpp:160-252  c0 albedo, s0 roughness, s1 anisotropic      */
    RT_MAT_DISNEY_CLEARCOAT = 6, /* Material.cpp:256-329  s0 clearcoat_gloss                           */
    RT_MAT_DISNEY_GLASS = 7,     /* Material.cpp:333-460  c0 albedo, s0 roughness, s1 aniso, s2 eta    */
    RT_MAT_ROUGH_DIELECTRIC = 8, /* Material.cpp:463-542  c0 refl, c1 trans, s0 eta, s1 roughness      */
    RT_MAT_DISNEY_SHEEN = 9,     /* Material.cpp:545-567  c0 base, s0 sheen_tint                       */
    RT_MAT_DISNEY_UBER = 10,     /* Material.cpp:571-722  c0 base, s[0..12] = Material.h:184-196 order */
    RT_MAT_GLASS = 11            /* GlassMaterial      Material.h:240-273 s0 = index_r                 */
};
typedef struct rt_material {
    int32_t type;
    float c0[3];
    float c1[3];
    float s[13];
} rt_material;

enum { RT_LIGHT_RECT = 0, RT_LIGHT_TRIMESH = 1 };

/* Area light.  RECT restates an Instance(Rectangle, EmissiveMaterial, mat4)
 * pushed into Scene::lights (Object.h:61-135,364-423; main.cpp:801-813).
 * TRIMESH (extension: the reference cannot sample mesh emitters) is a range of
 * rt_emissive_tri entries with an area CDF. */
typedef struct rt_light {
    int32_t kind;
    int32_t material; /* index of the emissive material                      */
    int32_t first;    /* TRIMESH: first entry in emissive_tris               */
    int32_t count;    /* TRIMESH: number of entries                          */
    float area;       /* Instance::get_area (Object.h:404-407) / mesh area   */
    float n[3];       /* RECT: Rectangle N, T, B (model space)               */
    float t[3];
    float b[3];
    float hu;         /* RECT: half side lengths u, v (Object.h:68-69)       */
    float hv;
    float m[16];      /* RECT: model-to-world, column-major (Math.h:338-347) */
    float inv_m[16];  /* RECT: world-to-model                                */
} rt_light;

typedef struct rt_emissive_tri {
    int32_t slot; /* triangle slot (leaf order) */
    float cdf;    /* cumulative area fraction up to and including this entry */
} rt_emissive_tri;

/* Everything rt_scene_create copies to HBM once ("upload once"); the caller
 * keeps ownership of every host array. */
typedef struct rt_scene_desc {
    int32_t root_ref;              /* child-ref encoding of BVH node 0       */
    int64_t n_pairs;
    const rt_bvh_pair* pairs;
    int64_t n_tris;
    const rt_tri* tris;
    const rt_tri_normals* normals;
    int32_t n_materials;
    const rt_material* materials;
    int32_t n_lights;
    const rt_light* lights;
    int32_t n_emissive_tris;
    const rt_emissive_tri* emissive_tris;
    /* EnviormentLight (main.cpp:88-193): RGB_32f lat-long image, its flat CDF
     * (build_cdf, main.cpp:97-120) and total_sum; offset = degrees/360. */
    int32_t env_w;
    int32_t env_h;
    const float* env_rgb;
    const float* env_cdf;
    float env_total;
    float env_offset;
} rt_scene_desc;

/* ---- legacy two-level scenes: Scene::instances under a TLAS (Scene.cpp:6-77), Instance (Object.h:364-423) over the
 * Geometry classes (Object.h:26-243 analytic, 266-362 TriangleMesh).  SURVEY 8 f-2. */
enum {
    RT_GEO_MESH = 0,      /* TriangleMesh  Object.h:266-362  blas_root = its BVH within rt_scene_desc.pairs / tris (model space) */
    RT_GEO_SPHERE = 1,    /* Sphere        Object.h:26-60    p0 = radius                                            */
    RT_GEO_RECTANGLE = 2, /* Rectangle     Object.h:61-135   p0-2 N, p3-5 T, p6-8 B, p9 u, p10 v (half side lengths) */
    RT_GEO_DISK = 3,      /* Disk          Object.h:137-168  p0-2 normal, p3 outer_radius, p4 inner_radius          */
    RT_GEO_BOX = 4,       /* Box           Object.h:170-209  p0-2 bmin, p3-5 bmax                                   */
    RT_GEO_CYLINDER = 5   /* Cylinder      Object.h:211-243  p0 radius, p1 ymin, p2 ymax                            */
};
typedef struct rt_instance {
    int32_t geometry;
    int32_t material;  /* Instance::material: replaces the triangles' own (Object.h:387) */
    int32_t blas_root; /* RT_GEO_MESH: child-ref encoding of the mesh BVH's node 0 (refs index rt_scene_desc.pairs / tris) */
    int32_t reserved;
    float area;        /* Instance::get_area (Object.h:404-407): Rectangle (2u)(2v), every other Geometry 0 */
    float p[11];
    float m[16];       /* Transform: model-to-world, column-major (Math.h:641-698) */
    float inv_m[16];   /* world-to-model */
} rt_instance;

/* The TLAS (BVH::build(world bounds, 1, BVH_SAH), Scene.cpp:6-15) as child-pair records; a leaf ref is ~instance index
 * (one instance per leaf). */
typedef struct rt_tlas_desc {
    int32_t root_ref;
    int64_t n_pairs;
    const rt_bvh_pair* pairs;
    int32_t n_instances;
    const rt_instance* instances;
} rt_tlas_desc;

enum { RT_FILTER_GAUSSIAN = 0, RT_FILTER_BOX = 1 };

/* Camera (Scene.h:10-81): raster_to_world and camera_pos as built by the
 * constructor, plus the pixel filter sampled at ray generation
 * (Filter.h:12-42; param = stddev or width). */
typedef struct rt_camera {
    float raster_to_world[16];
    float pos[3];
    int32_t filter;
    float filter_param;
} rt_camera;

#define RT_RENDER_CLEAR 1u      /* zero radiance_sum before accumulating      */
#define RT_RENDER_DEVICE_OUT 2u /* radiance_sum is a device pointer (see rt_render: stream ordering is the caller's) */
#define RT_RENDER_TIMING 4u     /* bracket every kernel with HIP events       */
#define RT_RENDER_SERIAL 8u     /* one sample pass at a time on one stream (per-kernel times then are stand-alone times) */

/* Options (main.cpp:836-850) + the sample window this call renders.
 * Sample i of pixel (x,y) is hammersley_2d(i, spp_total) (Utils.h:148-151);
 * this call renders i in [spp_begin, spp_begin + spp_count) for pixels in
 * [x0,x1) x [y0,y1) of a width x height raster. */
typedef struct rt_options {
    int32_t width;
    int32_t height;
    int32_t spp_begin;
    int32_t spp_count;
    int32_t spp_total;
    int32_t max_depth;     /* MAX_DEPTH, main.cpp:33,335 */
    int32_t x0, y0, x1, y1;
    uint32_t seed;
    int32_t spp_per_pass;  /* samples per pixel traced together in one wavefront pass; 0 = auto */
    uint32_t flags;
} rt_options;

typedef struct rt_stats {
    uint64_t rays_closest; /* closest-hit queries issued (main.cpp:338)      */
    uint64_t rays_shadow;  /* shadow queries issued (main.cpp:250,281)       */
    uint64_t paths;        /* camera samples started                         */
    uint64_t launches_closest;
    uint64_t launches_shadow;
    float ms_total;        /* whole call, HIP events on the scene's stream   */
    float ms_generate;     /* per kernel, only with RT_RENDER_TIMING         */
    float ms_closest;
    float ms_shade;
    float ms_shadow;
    float ms_accumulate;
} rt_stats;

/* ---- entry points ----------------------------------------------------- */

const char* rt_last_error(void);

/* Number of HIP devices visible (0 when none). */
int rt_device_count(void);

/* Replaces Scene::build_top_level_bvh + the per-mesh BVH ownership
 * (Scene.cpp:6-15, Object.cpp:24-29): uploads the flattened scene to
 * `device` and returns an opaque handle. */
int rt_scene_create(const rt_scene_desc* desc, int device, rt_scene** out);
void rt_scene_destroy(rt_scene* scene);

/* The legacy two-level scene: replaces Scene::build_top_level_bvh (Scene.cpp:6-15) + every TriangleMesh's own BVH
 * (Object.cpp:11-29).  `geometry` holds the model-space BVHs and triangles of all meshes, concatenated (root_ref unused;
 * n_tris may be 0; the triangles' material / light fields and emissive_tris are ignored; lights must be RT_LIGHT_RECT),
 * plus materials, lights and environment as in rt_scene_create.  Every entry point below works on such a scene:
 * closest hit walks the TLAS and, inside a mesh instance, the mesh's BVH with the ray in model space (Object.h:380-394);
 * shadow queries answer Scene::trace_scene as a boolean (Scene.h:113-129), walking the TLAS instead of the reference's
 * loop over all instances (an instance whose world bounds the segment misses is skipped). */
int rt_scene_create_instanced(const rt_scene_desc* geometry, const rt_tlas_desc* tlas, int device, rt_scene** out);

/* Replaces bool Scene::closest_hit(Ray, Intersection*) const
 * (Scene.cpp:17-77 over TriangleMesh::intersect, Object.h:273-357) for a
 * batch of n rays; host pointers.  The records are copied to the device as they are and laid out there; results return
 * straight into `hits`.  The scene keeps one device allocation for these calls (grown on demand, kept up to 1 GiB):
 * calls on one scene must not overlap. */
int rt_trace_closest(rt_scene* scene, const rt_ray* rays, int64_t n, rt_hit* hits);

/* The same with Intersection::index (Scene.cpp:42): index[i] = instance hit, -1 on a miss (0 on a hit in a scene without
 * instances).  In a two-level scene hits[i].tri is the triangle's index within the instance's mesh, 0 for an analytic
 * primitive; (u, v) are the Rectangle's si->u / si->v (Object.h:116-117) and 0 for the other analytic primitives. */
int rt_trace_closest_indexed(rt_scene* scene, const rt_ray* rays, int64_t n, rt_hit* hits, int32_t* index);

/* Replaces Scene::trace_scene(Ray, 0, tmax, &tmp) used as a boolean
 * occlusion test (Scene.h:113-129; call sites main.cpp:250,281) — what the
 * declared-but-undefined Scene::any_hit (Scene.h:133) was to be.
 * occluded[i] = 1 if any triangle is hit with 1e-4 < t <= tmax[i].
 * The walk skips boxes entered beyond tmax * 1.0001 + 1e-6 (the reference enters them and rejects their triangles one
 * by one): same flags unless rounding puts a triangle's computed t at or below tmax while a box around it starts beyond
 * that margin (never observed, compared flag for flag with the reference's procedure in tests/). */
int rt_trace_any(rt_scene* scene, const rt_ray* rays, const float* tmax, int64_t n,
                 uint8_t* occluded);

/* How the scene's occlusion queries (rt_trace_any, the shadow rays of rt_render) walk the tree: 0 = the child-pair records
 * (the default), n > 0 = n compressed four-wide records built from the same pairs (RT_TUNE_QUADS=1 in the environment of
 * rt_scene_create; DESIGN.md section 4: same flags bit for bit, measured slower on the headline workload).  No counterpart
 * in the reference (its trace_scene has one walk, Object.h:273-344). */
int64_t rt_scene_quad_records(const rt_scene* scene);

/* The records behind rt_scene_quad_records, built on the host WITHOUT a GPU (what rt_scene_create builds under RT_TUNE_QUADS=1),
 * for inspection and for tests: 16 words per record, `*n_quads` quad records followed by one leaf record per triangle slot
 * (layout: raytracing_amd/csrc/rt_device.h, "four-wide occlusion walk").  out_words may be null to ask for the size: the call
 * then only sets *n_quads.  Returns RT_ERR_INVALID when the tree cannot be collapsed (one leaf, boxes that do not nest, a
 * worst-case stack beyond 96 entries: *stack_need says how deep), RT_ERR_LIMIT when capacity_words is too small.  No
 * counterpart in the reference. */
int rt_collapse_quads(const rt_bvh_pair* pairs, int64_t n_pairs, int32_t root_ref, const rt_tri* tris, int64_t n_tris,
                      uint32_t* out_words, int64_t capacity_words, int64_t* n_quads, int32_t* qroot, int32_t* stack_need);

/* rt_trace_closest for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE: it exists for callers that cannot
 * call hipMalloc — tight serving loops, harnesses that forbid allocation, code that sets up HIP graph capture around its OWN
 * streams (the call itself synchronises the scene's stream and asks for pointer attributes: it cannot run INSIDE a capture).  Same walk, same
 * kernels and same results as rt_trace_closest, but `rays` and `hits` are DEVICE pointers on the scene's GPU (hits
 * 16-byte aligned; a miss is tri = -1, t = inf, u = v = 0) and the walk's scratch (two float4 ray streams, shard counters
 * and fetch cursors, the second hit record of two-level scenes) is a static device array of the library: 24 MiB + 16 KiB
 * per GPU, held until the process ends.  Limit: n <= RT_NOALLOC_MAX_RAYS, beyond it RT_ERR_LIMIT.  The call makes no
 * hipMalloc / hipFree and no allocating library call (its error messages are string literals); it runs on the scene's
 * own stream and returns when the hits are written, so the caller must have finished writing `rays` (synchronise the
 * producing stream first).  All calls on one GPU share that GPU's array, whatever the scene handle: calls that overlap
 * on one GPU race and must be serialised by the caller. */
#define RT_NOALLOC_MAX_RAYS (1 << 19)
int rt_trace_closest_noalloc(rt_scene* scene, const rt_ray* rays, int64_t n, rt_hit* hits);

/* Replaces the body of Renderer::render_image (main.cpp:950-1005): for every
 * pixel of the tile and every sample of the window, generate the camera ray
 * (Scene.h:61-75), run get_ray_color (main.cpp:325-441) as a wavefront
 * (generate / closest-hit / shade+NEE / shadow / accumulate kernels) and add
 * the NaN-guarded sample (main.cpp:984-987) into radiance_sum[(y*width+x)*3],
 * y in raster order (the flip of main.cpp:990 happens in rt_resolve).
 * RT_RENDER_DEVICE_OUT: the library works on its own non-blocking streams, which nothing orders after work the caller has
 * queued on other streams — fills or copies into radiance_sum (e.g. a torch.zeros) must have FINISHED before the call
 * (synchronise that stream first); the call returns after its own work is complete, so the buffer may be read on any
 * stream afterwards. */
int rt_render(rt_scene* scene, const rt_camera* cam, const rt_options* opt,
              float* radiance_sum, rt_stats* stats);

/* rt_render over a LIST of tiles in one call: tiles_xyxy = n_tiles x (x0, y0, x1, y1), pairwise disjoint rectangles inside
 * the raster (overlap is refused: RT_ERR_INVALID); opt->x0..y1 are ignored.  The reference hands 16x16 tiles to its threads
 * one at a time (TileQueue, main.cpp:904-935); a GPU needs millions of paths in flight, so a rank of an image-tile-sharded
 * render (SURVEY 8 e: interleaved 64x64 tiles, tile t -> GPU t mod G) passes ALL its tiles at once.  Every pixel gets
 * exactly the sums rt_render gives it (the random streams are keyed by raster position and sample index). */
int rt_render_tiles(rt_scene* scene, const rt_camera* cam, const rt_options* opt, const int32_t* tiles_xyxy, int32_t n_tiles,
                    float* radiance_sum, rt_stats* stats);

/* Replaces Renderer::add_to_buffer (main.cpp:865-877) for the whole image:
 * 1/spp, pow(1/gamma), clamp*255, row H-y-1.  Host pointers. */
int rt_resolve(rt_scene* scene, const float* radiance_sum, int32_t width, int32_t height,
               int32_t spp, float gamma, uint8_t* rgb_out);

/* Blocks until all work queued on the scene's stream has finished. */
int rt_scene_sync(rt_scene* scene);

/* ---- multi-GPU: the frame's one reduce (SURVEY 8 e).  The reference has no counterpart (one process, std::thread over
 * tiles: main.cpp:954-1005); these close a frame whose sample passes or image tiles were rendered by several GPUs, each
 * into its own W*H*3 fp32 buffer: ONE sum into the root's buffer over RCCL / xGMI.  RCCL is loaded at run time; nothing
 * here is needed, or touched, for one GPU. */
typedef struct rt_comm rt_comm; /* opaque: one rank of a communicator, bound to one GPU */
#define RT_COMM_ID_BYTES 128

/* Rank 0 makes the id, every rank receives it by whatever channel the host has (a file, MPI, torch.distributed, ...). */
int rt_comm_unique_id(uint8_t* id128);
/* One rank per process (or thread): collective over all `world` ranks. */
int rt_comm_create(const uint8_t* id128, int rank, int world, int device, rt_comm** out);
/* All ranks in this process: out[i] = rank i on devices[i].  Distinct devices: an RCCL communicator.  One device listed
 * n times: a REHEARSAL communicator (host barrier + a device add in rank order) for one-GPU boxes; anything else is refused. */
int rt_comm_create_local(const int* devices, int n_devices, rt_comm** out);
int rt_comm_rank(const rt_comm* comm);
int rt_comm_world(const rt_comm* comm);
/* buf (device memory on the rank's GPU, n floats): on return the root's buf holds the sum over all ranks, the other
 * ranks' bufs are unchanged.  Collective: every rank calls it, each from its own host thread; returns when this rank's
 * part has finished.  The caller's writes to buf must have completed (rt_render returns only then). */
int rt_reduce(rt_comm* comm, float* buf, size_t n, int root);
void rt_comm_destroy(rt_comm* comm);

/* Device frames (the buffers rt_render(RT_RENDER_DEVICE_OUT) accumulates into and rt_reduce sums) for hosts that are not
 * compiled with HIP: allocate on `device`, copy to host memory, clear, free. */
int rt_buffer_alloc(int device, size_t bytes, void** out);
int rt_buffer_download(int device, const void* src, void* host_dst, size_t bytes);
int rt_buffer_zero(int device, void* p, size_t bytes);
void rt_buffer_free(int device, void* p);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
