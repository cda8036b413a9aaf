"""raytracing_amd — script-side bindings of the MI355X path-tracing hot path.

Two native libraries do the work:
  librt_hip.so   the gfx950 kernels behind the C-ABI of include/rt_hip.h
  librt_host.so  the C++ host mirror of the reference's Scene/Mesh/Material/BVH API
This module only marshals numpy arrays into them with ctypes.  There is no CPU
fallback: when librt_hip.so is missing, or no GPU is visible, GPU calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_hip = None
_host = None

# ---- numpy views of the C structs (include/rt_hip.h) -------------------------
RAY = np.dtype([("o", "<f4", 3), ("d", "<f4", 3)])
HIT = np.dtype([("tri", "<i4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4")])
VERTEX = np.dtype([("pos", "<f4", 3), ("normal", "<f4", 3), ("uv", "<f4", 2)])
TRIANGLE = np.dtype([("idx", "<u4", 3), ("material", "<u4")])
NODE = np.dtype([("bmin", "<f4", 3), ("bmax", "<f4", 3), ("left_node", "<i4"), ("count", "<i4")])
MATERIAL = np.dtype([("type", "<i4"), ("c0", "<f4", 3), ("c1", "<f4", 3), ("s", "<f4", 13)])
LIGHT = np.dtype([("kind", "<i4"), ("material", "<i4"), ("first", "<i4"), ("count", "<i4"), ("area", "<f4"),
                  ("n", "<f4", 3), ("t", "<f4", 3), ("b", "<f4", 3), ("hu", "<f4"), ("hv", "<f4"),
                  ("m", "<f4", 16), ("inv_m", "<f4", 16)])
EMISSIVE = np.dtype([("slot", "<i4"), ("cdf", "<f4")])
PAIR = np.dtype([("lmin", "<f4", 3), ("lmax", "<f4", 3), ("rmin", "<f4", 3), ("rmax", "<f4", 3),
                 ("lref", "<i4"), ("rref", "<i4"), ("reserved", "<i4", 2)])
FLAT_TRI = np.dtype([("v0", "<f4", 3), ("tri_id", "<u4"), ("e1", "<f4", 3), ("material", "<i4"),
                     ("e2", "<f4", 3), ("light", "<i4")])
FLAT_NORMALS = np.dtype([("n0", "<f4", 3), ("pad0", "<f4"), ("n1", "<f4", 3), ("pad1", "<f4"),
                         ("n2", "<f4", 3), ("pad2", "<f4")])
assert RAY.itemsize == 24 and HIT.itemsize == 16 and VERTEX.itemsize == 32 and TRIANGLE.itemsize == 16
assert NODE.itemsize == 32 and MATERIAL.itemsize == 80 and PAIR.itemsize == 64 and FLAT_TRI.itemsize == 48
INSTANCE = np.dtype([("geometry", "<i4"), ("material", "<i4"), ("blas_root", "<i4"), ("reserved", "<i4"), ("area", "<f4"),
                     ("p", "<f4", 11), ("m", "<f4", 16), ("inv_m", "<f4", 16)])
assert LIGHT.itemsize == 192 and FLAT_NORMALS.itemsize == 48 and INSTANCE.itemsize == 192
GEO_MESH, GEO_SPHERE, GEO_RECTANGLE, GEO_DISK, GEO_BOX, GEO_CYLINDER = range(6)

MAT_MATTE, MAT_GGX, MAT_BECKMANN, MAT_EMISSIVE, MAT_DISNEY_DIFFUSE, MAT_DISNEY_METAL = 0, 1, 2, 3, 4, 5
MAT_DISNEY_CLEARCOAT, MAT_DISNEY_GLASS, MAT_ROUGH_DIELECTRIC, MAT_DISNEY_SHEEN, MAT_DISNEY_UBER, MAT_GLASS = 6, 7, 8, 9, 10, 11
GLTF_UBER, GLTF_MATTE, GLTF_GGX = 0, 1, 2
FILTER_GAUSSIAN, FILTER_BOX = 0, 1
RENDER_CLEAR, RENDER_DEVICE_OUT, RENDER_TIMING, RENDER_SERIAL = 1, 2, 4, 8

(ARR_VERTS, ARR_TRIS, ARR_NODES, ARR_INDICES, ARR_MATERIALS, ARR_LIGHTS, ARR_EMISSIVE, ARR_TRI_LIGHT, ARR_ENV_RGB,
 ARR_ENV_CDF, ARR_FLAT_PAIRS, ARR_FLAT_TRIS, ARR_FLAT_NORMALS, ARR_EMISSIVE_ORIG, ARR_TLAS_NODES, ARR_TLAS_INDICES,
 ARR_TLAS_PAIRS, ARR_INSTANCES) = range(18)
_ARR_DTYPES = {ARR_VERTS: VERTEX, ARR_TRIS: TRIANGLE, ARR_NODES: NODE, ARR_INDICES: np.dtype("<i4"),
               ARR_MATERIALS: MATERIAL, ARR_LIGHTS: LIGHT, ARR_EMISSIVE: EMISSIVE, ARR_TRI_LIGHT: np.dtype("<i4"),
               ARR_ENV_RGB: np.dtype("<f4"), ARR_ENV_CDF: np.dtype("<f4"), ARR_FLAT_PAIRS: PAIR,
               ARR_FLAT_TRIS: FLAT_TRI, ARR_FLAT_NORMALS: FLAT_NORMALS, ARR_EMISSIVE_ORIG: np.dtype("<i4"),
               ARR_TLAS_NODES: NODE, ARR_TLAS_INDICES: np.dtype("<i4"), ARR_TLAS_PAIRS: PAIR, ARR_INSTANCES: INSTANCE}


class RtCamera(C.Structure):
    _fields_ = [("raster_to_world", C.c_float * 16), ("pos", C.c_float * 3), ("filter", C.c_int32),
                ("filter_param", C.c_float)]


class RtOptions(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("spp_begin", C.c_int32), ("spp_count", C.c_int32),
                ("spp_total", C.c_int32), ("max_depth", C.c_int32), ("x0", C.c_int32), ("y0", C.c_int32),
                ("x1", C.c_int32), ("y1", C.c_int32), ("seed", C.c_uint32), ("spp_per_pass", C.c_int32),
                ("flags", C.c_uint32)]


class RtStats(C.Structure):
    _fields_ = [("rays_closest", C.c_uint64), ("rays_shadow", C.c_uint64), ("paths", C.c_uint64),
                ("launches_closest", C.c_uint64), ("launches_shadow", C.c_uint64), ("ms_total", C.c_float),
                ("ms_generate", C.c_float), ("ms_closest", C.c_float), ("ms_shade", C.c_float),
                ("ms_shadow", C.c_float), ("ms_accumulate", C.c_float)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class RtError(RuntimeError):
    pass


HIP_SYMBOLS = ["rt_last_error", "rt_device_count", "rt_scene_create", "rt_scene_create_instanced", "rt_scene_destroy",
               "rt_trace_closest", "rt_trace_closest_indexed", "rt_trace_any", "rt_trace_closest_noalloc", "rt_render", "rt_render_tiles",
               "rt_resolve", "rt_scene_sync", "rt_comm_unique_id", "rt_comm_create", "rt_comm_create_local", "rt_comm_rank",
               "rt_comm_world", "rt_reduce", "rt_comm_destroy", "rt_buffer_alloc", "rt_buffer_download", "rt_buffer_zero",
               "rt_buffer_free", "rt_scene_quad_records", "rt_collapse_quads"]
NOALLOC_MAX_RAYS = 1 << 19


def load_libraries():
    """dlopen both libraries (idempotent).  Raises when a library is missing."""
    global _hip, _host
    if _hip is not None:
        return _hip, _host
    hip_path = os.path.join(_HERE, "librt_hip.so")
    host_path = os.path.join(_HERE, "librt_host.so")
    for p in (hip_path, host_path):
        if not os.path.exists(p):
            raise RtError(f"{p} is missing: run `python __graft_entry__.py` to build the HIP extension")
    hip = C.CDLL(hip_path, mode=C.RTLD_GLOBAL)
    host = C.CDLL(host_path)
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    hip.rt_last_error.restype = C.c_char_p
    hip.rt_device_count.restype = C.c_int
    hip.rt_trace_closest.argtypes = [vp, vp, i64, vp]
    hip.rt_trace_closest_indexed.argtypes = [vp, vp, i64, vp, vp]
    hip.rt_trace_any.argtypes = [vp, vp, vp, i64, vp]
    hip.rt_trace_closest_noalloc.argtypes = [vp, vp, i64, vp]
    hip.rt_render.argtypes = [vp, C.POINTER(RtCamera), C.POINTER(RtOptions), vp, C.POINTER(RtStats)]
    hip.rt_render_tiles.argtypes = [vp, C.POINTER(RtCamera), C.POINTER(RtOptions), vp, i32, vp, C.POINTER(RtStats)]
    hip.rt_comm_unique_id.argtypes = [vp]
    hip.rt_comm_create.argtypes = [vp, i32, i32, i32, C.POINTER(vp)]
    hip.rt_comm_create_local.argtypes = [vp, i32, vp]
    hip.rt_comm_rank.argtypes = [vp]
    hip.rt_comm_world.argtypes = [vp]
    hip.rt_reduce.argtypes = [vp, vp, C.c_size_t, i32]
    hip.rt_comm_destroy.argtypes = [vp]
    hip.rt_comm_destroy.restype = None
    hip.rt_resolve.argtypes = [vp, vp, i32, i32, i32, f32, vp]
    hip.rt_scene_sync.argtypes = [vp]
    hip.rt_scene_quad_records.argtypes = [vp]
    hip.rt_scene_quad_records.restype = i64
    hip.rt_collapse_quads.argtypes = [vp, i64, i32, vp, i64, vp, i64, vp, vp, vp]
    host.rth_bvh_build.argtypes = [vp, i32, i32, i32, vp, vp]
    host.rth_bvh_build.restype = i32
    host.rth_scene_new.restype = vp
    host.rth_scene_free.argtypes = [vp]
    host.rth_scene_load_gltf.argtypes = [vp, C.c_char_p, i32, f32]
    host.rth_scene_add_mesh.argtypes = [vp, vp, i64, vp, i64]
    host.rth_scene_add_instance.argtypes = [vp, vp, i64, vp, i64, vp, i32]
    host.rth_scene_load_env_hdr.argtypes = [vp, C.c_char_p, f32]
    host.rth_scene_load_env_hdr.restype = i32
    host.rth_scene_get_env.argtypes = [vp, vp, vp]
    host.rth_scene_get_env.restype = i32
    host.rth_scene_add_instance.restype = None
    host.rth_scene_add_material.argtypes = [vp, vp]
    host.rth_scene_set_material.argtypes = [vp, i32, vp]
    host.rth_scene_add_rect_light.argtypes = [vp, vp, f32, f32, vp, vp]
    host.rth_scene_set_env.argtypes = [vp, i32, i32, vp, f32]
    host.rth_scene_build.argtypes = [vp, i32, i32]
    host.rth_scene_set_bvh.argtypes = [vp, vp, i64, vp, i64]
    host.rth_scene_upload.argtypes = [vp, i32]
    host.rth_scene_gpu.argtypes = [vp]
    host.rth_scene_gpu.restype = vp
    host.rth_scene_array.argtypes = [vp, i32, C.POINTER(i64)]
    host.rth_scene_array.restype = vp
    host.rth_scene_info.argtypes = [vp, vp, vp, vp, vp, vp]
    host.rth_look_at.argtypes = [vp, vp, vp, vp]
    host.rth_mat4_inverse.argtypes = [vp, vp]
    host.rth_mat4_inverse.restype = f32
    host.rth_camera_init.argtypes = [vp, f32, i32, i32, i32, f32, C.POINTER(RtCamera)]
    host.rth_camera_ray.argtypes = [vp, f32, i32, i32, f32, f32, vp]
    host.rth_write_image.argtypes = [C.c_char_p, i32, i32, vp]
    host.rth_write_image.restype = i32
    host.rth_render_image.argtypes = [vp, vp, f32, i32, i32, i32, i32, C.c_uint32, C.c_char_p, vp, vp, C.POINTER(RtStats)]
    host.rth_render_image_sharded.argtypes = [vp, vp, f32, i32, i32, i32, i32, C.c_uint32, vp, i32, i32, i32, vp, vp, C.POINTER(RtStats)]
    host.rth_shard_sample_window.argtypes = [i32, i32, i32, vp]
    host.rth_shard_tiles.argtypes = [i32, i32, i32, i32, i32, vp, i32]
    host.rth_shard_tiles.restype = i32
    host.rth_scene_closest_hit.argtypes = [vp, vp, vp, vp, vp, vp]
    host.rth_scene_new_triangle_mesh.argtypes = [vp, vp, i64, vp, i64, i32]
    host.rth_scene_new_triangle_mesh.restype = i32
    host.rth_scene_add_mesh_instance.argtypes = [vp, i32, i32, vp]
    host.rth_scene_add_mesh_instance.restype = i32
    host.rth_scene_add_primitive_instance.argtypes = [vp, i32, vp, i32, vp, i32]
    host.rth_scene_add_primitive_instance.restype = i32
    host.rth_scene_mesh_array.argtypes = [vp, i32, i32, C.POINTER(i64)]
    host.rth_scene_mesh_array.restype = vp
    host.rth_scene_instance_bounds.argtypes = [vp, i32, vp]
    host.rth_scene_instance_bounds.restype = i32
    host.rth_scene_any_hit.argtypes = [vp, vp]
    _hip, _host = hip, host
    return _hip, _host


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a, n=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if n is not None and a.size != n:
        raise ValueError(f"expected {n} floats, got {a.size}")
    return a


def _check(rc):
    if rc != 0:
        raise RtError(f"rt error {rc}: {_hip.rt_last_error().decode()}")


def device_count():
    hip, _ = load_libraries()
    return int(hip.rt_device_count())


def last_error():
    hip, _ = load_libraries()
    return hip.rt_last_error().decode()


def write_image(path, rgb):
    """The Renderer's output writer (main.cpp:1015) on its own: rgb = (h, w, 3) uint8, row 0 = top; `.png` by
    extension (stored deflate blocks), anything else is written as a 24-bit BMP."""
    _, host = load_libraries()
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    if rgb.ndim != 3 or rgb.shape[2] != 3 or rgb.shape[0] < 1 or rgb.shape[1] < 1:
        raise ValueError("write_image: rgb must be (h, w, 3) uint8")
    if not host.rth_write_image(os.fsencode(path), rgb.shape[1], rgb.shape[0], _ptr(rgb)):
        raise OSError(f"write_image: cannot write {path}")


def look_at(eye, at, up):
    """Math.h:595-610 -> column-major 16 floats."""
    _, host = load_libraries()
    out = np.zeros(16, np.float32)
    host.rth_look_at(_ptr(_f32(eye, 3)), _ptr(_f32(at, 3)), _ptr(_f32(up, 3)), _ptr(out))
    return out


def mat4_inverse(m16):
    """inverse(mat4) of the host mirror (Math.h:454-542: fp32 adjugate over determinant; a singular matrix comes back
    unchanged) -> (inverse as 16 column-major floats, determinant)."""
    _, host = load_libraries()
    out = np.zeros(16, np.float32)
    det = host.rth_mat4_inverse(_ptr(_f32(m16, 16)), _ptr(out))
    return out, float(det)


def camera(view16, fov_deg, width, height, filter=FILTER_GAUSSIAN, filter_param=0.5):
    """Camera constructor (Scene.h:14-44) -> RtCamera."""
    _, host = load_libraries()
    cam = RtCamera()
    host.rth_camera_init(_ptr(_f32(view16, 16)), fov_deg, width, height, filter, filter_param, C.byref(cam))
    return cam


def camera_ray(view16, fov_deg, width, height, raster_x, raster_y):
    """Camera::get_ray on the host (Scene.h:61-75)."""
    _, host = load_libraries()
    out = np.zeros(1, RAY)
    host.rth_camera_ray(_ptr(_f32(view16, 16)), fov_deg, width, height, raster_x, raster_y, _ptr(out))
    return out[0]


def shard_sample_window(rank, world, spp):
    """The host mirror's sample-pass partition: rank's (begin, count) window of the one spp-sample set."""
    _, host = load_libraries()
    out = np.zeros(2, np.int32)
    host.rth_shard_sample_window(rank, world, spp, _ptr(out))
    return int(out[0]), int(out[1])


def shard_tiles(rank, world, width, height, tile=64):
    """The host mirror's image-tile partition: interleaved tile x tile tiles, tile t -> rank t mod world; n x (x0, y0, x1, y1)."""
    _, host = load_libraries()
    n = host.rth_shard_tiles(rank, world, width, height, tile, None, 0)
    out = np.zeros((max(n, 0), 4), np.int32)
    if n > 0:
        host.rth_shard_tiles(rank, world, width, height, tile, _ptr(out), n)
    return out


def collapse_quads(pairs, root_ref, flat_tris):
    """rt_collapse_quads: the occlusion walk's compressed four-wide records of a pair tree, built on the host (no GPU).
    -> (words [n_records, 16] uint32: quad records then one leaf record per triangle slot, n_quads, qroot, worst-case stack)"""
    hip, _ = load_libraries()
    pairs = np.ascontiguousarray(pairs, dtype=PAIR)
    tris = np.ascontiguousarray(flat_tris, dtype=FLAT_TRI)
    nq, root, need = C.c_int64(0), C.c_int32(0), C.c_int32(0)
    _check(hip.rt_collapse_quads(_ptr(pairs), len(pairs), int(root_ref), _ptr(tris), len(tris), None, 0, C.byref(nq), C.byref(root), C.byref(need)))
    out = np.zeros((nq.value + len(tris), 16), np.uint32)
    _check(hip.rt_collapse_quads(_ptr(pairs), len(pairs), int(root_ref), _ptr(tris), len(tris), _ptr(out), out.size, C.byref(nq), C.byref(root), C.byref(need)))
    return out, int(nq.value), int(root.value), int(need.value)


def bvh_build(bounds6, max_per_node=1, strategy=1):
    """BVH::build (BVH.cpp:131-149) on the host. strategy: 0 = BVH_MIDDLE, 1 = BVH_SAH."""
    _, host = load_libraries()
    b = _f32(bounds6).reshape(-1, 6)
    n = b.shape[0]
    nodes = np.zeros(max(2 * n, 1), NODE)
    idx = np.zeros(n, np.int32)
    k = host.rth_bvh_build(_ptr(b), n, max_per_node, strategy, _ptr(nodes), _ptr(idx))
    return nodes[:k].copy(), idx


def material(type, c0=(0, 0, 0), c1=(0, 0, 0), s=()):
    m = np.zeros(1, MATERIAL)
    m["type"] = type
    m["c0"] = c0
    m["c1"] = c1
    m["s"][0, : len(s)] = s
    return m


class Comm:
    """One rank of a frame-reduce communicator (rt_comm_*, include/rt_hip.h)."""

    def __init__(self, handle):
        self._h = handle

    @staticmethod
    def unique_id():
        load_libraries()
        buf = np.zeros(128, np.uint8)
        _check(_hip.rt_comm_unique_id(_ptr(buf)))
        return buf

    @staticmethod
    def create(unique_id, rank, world, device):
        load_libraries()
        uid = np.ascontiguousarray(unique_id, dtype=np.uint8)
        h = C.c_void_p()
        _check(_hip.rt_comm_create(_ptr(uid), rank, world, device, C.byref(h)))
        return Comm(h)

    @staticmethod
    def create_local(devices):
        load_libraries()
        dev = np.ascontiguousarray(devices, dtype=np.int32)
        out = (C.c_void_p * len(dev))()
        _check(_hip.rt_comm_create_local(_ptr(dev), len(dev), out))
        return [Comm(C.c_void_p(h)) for h in out]

    @property
    def rank(self):
        return int(_hip.rt_comm_rank(self._h))

    @property
    def world(self):
        return int(_hip.rt_comm_world(self._h))

    def reduce(self, device_ptr, n, root=0):
        """Sum of every rank's n-float device buffer into the root's (collective; blocks until this rank's part is done)."""
        _check(_hip.rt_reduce(self._h, C.c_void_p(device_ptr), n, root))

    def close(self):
        if self._h:
            _hip.rt_comm_destroy(self._h)
            self._h = None


class Scene:
    """Thin handle on the C++ rt::Scene (librt_host.so)."""

    def __init__(self):
        _, host = load_libraries()
        self._host = host
        self._h = host.rth_scene_new()
        self.n_triangle_meshes = 0   # handles returned by new_triangle_mesh
        self.instance_mesh = {}      # instance index -> mesh handle (mesh instances only)

    def close(self):
        if self._h:
            self._host.rth_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- construction
    def load_gltf(self, path, material_mode=GLTF_UBER, emissive_scale=1.0):
        if not self._host.rth_scene_load_gltf(self._h, os.fsencode(path), material_mode, emissive_scale):
            raise RtError(f"load_from_gltf failed for {path}")

    def add_mesh(self, verts, tris):
        verts = np.ascontiguousarray(verts, dtype=VERTEX)
        tris = np.ascontiguousarray(tris, dtype=TRIANGLE)
        if len(tris) and int(tris["idx"].max()) >= len(verts):
            raise ValueError("triangle index out of range")
        self._host.rth_scene_add_mesh(self._h, _ptr(verts), len(verts), _ptr(tris), len(tris))

    def add_instance(self, verts, tris, transform16, material=-1):
        """Scene::add_instance: the mesh placed with a column-major model-to-world matrix, flattened into the merged mesh."""
        verts = np.ascontiguousarray(verts, dtype=VERTEX)
        tris = np.ascontiguousarray(tris, dtype=TRIANGLE)
        if len(tris) and int(tris["idx"].max()) >= len(verts):
            raise ValueError("triangle index out of range")
        self._host.rth_scene_add_instance(self._h, _ptr(verts), len(verts), _ptr(tris), len(tris), _ptr(_f32(transform16, 16)), int(material))

    # ---- legacy two-level scenes: Scene::instances (Object.h:364-423).  A scene with instances is two-level: build()
    # builds the TLAS over their world bounds (Scene.cpp:6-15) and the merged mesh is unused.
    def new_triangle_mesh(self, verts, tris, max_per_leaf=1):
        """TriangleMesh(mesh): builds the mesh's own BVH once (max_per_leaf = BVH::build's max_per_node; the reference
        passes 1, Object.cpp:21); returns a handle for add_mesh_instance."""
        verts = np.ascontiguousarray(verts, dtype=VERTEX)
        tris = np.ascontiguousarray(tris, dtype=TRIANGLE)
        if len(tris) == 0 or int(tris["idx"].max()) >= len(verts):
            raise ValueError("empty mesh or triangle index out of range")
        self.n_triangle_meshes += 1
        return int(self._host.rth_scene_new_triangle_mesh(self._h, _ptr(verts), len(verts), _ptr(tris), len(tris), int(max_per_leaf)))

    def add_mesh_instance(self, mesh, material, transform16):
        i = self._host.rth_scene_add_mesh_instance(self._h, int(mesh), int(material), _ptr(_f32(transform16, 16)))
        if i < 0:
            raise ValueError("bad mesh handle or material index")
        self.instance_mesh[i] = int(mesh)
        return i

    def add_primitive_instance(self, kind, params, material, transform16, light=False):
        """Instance(new Sphere/Rectangle/Disk/Box/Cylinder(params), materials[material], transform); kind = GEO_*.
        params: SPHERE (radius) | RECTANGLE (nx, ny, nz, u_length, v_length) | DISK (nx, ny, nz, outer, inner) |
        BOX (sx, sy, sz) | CYLINDER (radius, height).  light=True also pushes it into Scene::lights (rectangles only)."""
        p = np.zeros(8, np.float32)
        params = np.asarray(params, np.float32).ravel()
        p[: len(params)] = params
        i = self._host.rth_scene_add_primitive_instance(self._h, int(kind), _ptr(p), int(material), _ptr(_f32(transform16, 16)), int(light))
        if i < 0:
            raise ValueError("bad geometry kind, material index, or a light that is not a rectangle")
        return i

    def mesh_array(self, mesh, which):
        """which: 0 verts, 1 triangles, 2 BVH nodes, 3 BVH indices of a new_triangle_mesh handle."""
        n = C.c_int64(0)
        p = self._host.rth_scene_mesh_array(self._h, int(mesh), which, C.byref(n))
        dt = [VERTEX, TRIANGLE, NODE, np.dtype("<i4")][which]
        if not p or n.value == 0:
            return np.zeros(0, dt)
        return np.frombuffer((C.c_char * (n.value * dt.itemsize)).from_address(p), dtype=dt).copy()

    def instance_bounds(self, i):
        out = np.zeros(6, np.float32)
        if not self._host.rth_scene_instance_bounds(self._h, int(i), _ptr(out)):
            raise ValueError("bad instance index")
        return out

    def add_material(self, rec):
        rec = np.ascontiguousarray(rec, dtype=MATERIAL)
        i = self._host.rth_scene_add_material(self._h, _ptr(rec))
        if i < 0:
            raise ValueError("unknown material type")
        return i

    def set_material(self, index, rec):
        rec = np.ascontiguousarray(rec, dtype=MATERIAL)
        if not self._host.rth_scene_set_material(self._h, index, _ptr(rec)):
            raise ValueError("bad material index or type")

    def add_rect_light(self, normal, u_length, v_length, emit, transform16):
        self._host.rth_scene_add_rect_light(self._h, _ptr(_f32(normal, 3)), u_length, v_length, _ptr(_f32(emit, 3)),
                                            _ptr(_f32(transform16, 16)))

    def set_env(self, rgb, degrees_offset=0.0):
        if rgb is None:
            self._host.rth_scene_set_env(self._h, 0, 0, None, 0.0)
            return
        rgb = _f32(rgb)
        h, w, _ = rgb.shape
        self._host.rth_scene_set_env(self._h, w, h, _ptr(rgb), degrees_offset)

    def load_env_hdr(self, path, degrees_offset=0.0):
        """load_hdr (Radiance .hdr) + environment light; False (and a message on stdout) when the file cannot be read."""
        return bool(self._host.rth_scene_load_env_hdr(self._h, os.fsencode(path), degrees_offset))

    def env(self):
        """The environment image the scene holds, (h, w, 3) float32, or None."""
        dims = np.zeros(2, np.int32)
        if not self._host.rth_scene_get_env(self._h, _ptr(dims), None):
            return None
        rgb = np.zeros((int(dims[1]), int(dims[0]), 3), np.float32)
        self._host.rth_scene_get_env(self._h, _ptr(dims), _ptr(rgb))
        return rgb

    def build(self, max_per_leaf=1, strategy=1):
        self._host.rth_scene_build(self._h, max_per_leaf, strategy)

    def set_bvh(self, nodes, indices):
        nodes = np.ascontiguousarray(nodes, dtype=NODE)
        indices = np.ascontiguousarray(indices, dtype=np.int32)
        if not self._host.rth_scene_set_bvh(self._h, _ptr(nodes), len(nodes), _ptr(indices), len(indices)):
            raise ValueError("BVH does not match the mesh")

    def upload(self, device=0):
        if not self._host.rth_scene_upload(self._h, device):
            raise RtError("scene upload failed: " + last_error())

    # ---- views
    def array(self, which):
        n = C.c_int64(0)
        p = self._host.rth_scene_array(self._h, which, C.byref(n))
        dt = _ARR_DTYPES[which]
        if not p or n.value == 0:
            return np.zeros(0, dt)
        buf = (C.c_char * (n.value * dt.itemsize)).from_address(p)
        return np.frombuffer(buf, dtype=dt).copy()

    def info(self):
        info = np.zeros(4, np.int32)
        tot, off, fov = np.zeros(1, np.float32), np.zeros(1, np.float32), np.zeros(1, np.float32)
        view = np.zeros(16, np.float32)
        self._host.rth_scene_info(self._h, _ptr(info), _ptr(tot), _ptr(off), _ptr(view), _ptr(fov))
        return {"env_w": int(info[0]), "env_h": int(info[1]), "root_ref": int(info[2]), "has_camera": bool(info[3]),
                "env_total": float(tot[0]), "env_offset": float(off[0]), "camera_view": view, "camera_fov": float(fov[0])}

    @property
    def gpu(self):
        g = self._host.rth_scene_gpu(self._h)
        if not g:
            raise RtError("scene is not on a GPU: call upload() first")
        return g

    # ---- GPU calls (straight through the C-ABI of include/rt_hip.h)
    def quad_records(self):
        """rt_scene_quad_records: 0 = occlusion queries walk the pair records, n = the compressed four-wide records."""
        return int(_hip.rt_scene_quad_records(self.gpu))

    def trace_closest(self, rays):
        rays = np.ascontiguousarray(rays, dtype=RAY)
        hits = np.zeros(len(rays), HIT)
        _check(_hip.rt_trace_closest(self.gpu, _ptr(rays), len(rays), _ptr(hits)))
        return hits

    def trace_closest_indexed(self, rays):
        """rt_trace_closest_indexed -> (hits, Intersection::index per ray)."""
        rays = np.ascontiguousarray(rays, dtype=RAY)
        hits = np.zeros(len(rays), HIT)
        index = np.zeros(len(rays), np.int32)
        _check(_hip.rt_trace_closest_indexed(self.gpu, _ptr(rays), len(rays), _ptr(hits), _ptr(index)))
        return hits, index

    def trace_closest_noalloc(self, rays_dev, n, hits_dev):
        """rt_trace_closest_noalloc: device addresses of n rt_ray records in and n rt_hit records out (the caller's memory);
        the walk's scratch is the library's static device array, nothing is allocated.  Returns the C status code."""
        return int(_hip.rt_trace_closest_noalloc(self.gpu, C.c_void_p(rays_dev), int(n), C.c_void_p(hits_dev)))

    def trace_any(self, rays, tmax):
        rays = np.ascontiguousarray(rays, dtype=RAY)
        tmax = _f32(tmax, len(rays))
        occ = np.zeros(len(rays), np.uint8)
        _check(_hip.rt_trace_any(self.gpu, _ptr(rays), _ptr(tmax), len(rays), _ptr(occ)))
        return occ

    def render(self, cam, width, height, spp, max_depth, seed=0, spp_begin=0, spp_total=None, tile=None,
               spp_per_pass=0, out=None, device_out=None, timing=False, clear=None, serial=False, tiles=None):
        """rt_render, or rt_render_tiles when `tiles` (n x (x0, y0, x1, y1), disjoint) is given.
        Returns (radiance_sum [H,W,3] fp32 or None with device_out, stats dict)."""
        opt = RtOptions()
        opt.width, opt.height = width, height
        opt.spp_begin, opt.spp_count = spp_begin, spp
        opt.spp_total = spp_total if spp_total is not None else spp_begin + spp
        opt.max_depth = max_depth
        opt.x0, opt.y0, opt.x1, opt.y1 = tile if tile is not None else (0, 0, width, height)
        opt.seed = seed
        opt.spp_per_pass = spp_per_pass
        flags = (RENDER_TIMING if timing else 0) | (RENDER_SERIAL if serial else 0)
        stats = RtStats()
        if tiles is not None:
            tiles = np.ascontiguousarray(tiles, dtype=np.int32).reshape(-1, 4)

        def call(target):
            if tiles is None:
                return _hip.rt_render(self.gpu, C.byref(cam), C.byref(opt), target, C.byref(stats))
            return _hip.rt_render_tiles(self.gpu, C.byref(cam), C.byref(opt), _ptr(tiles), len(tiles), target, C.byref(stats))

        if device_out is not None:
            flags |= RENDER_DEVICE_OUT
            if clear:
                flags |= RENDER_CLEAR
            opt.flags = flags
            _check(call(C.c_void_p(device_out)))
            return None, stats.as_dict()
        if out is None:
            out = np.zeros((height, width, 3), np.float32)
            flags |= RENDER_CLEAR
        elif clear:
            flags |= RENDER_CLEAR
        opt.flags = flags
        _check(call(_ptr(out)))
        return out, stats.as_dict()

    def resolve(self, radiance_sum, spp, gamma=2.2):
        radiance_sum = _f32(radiance_sum)
        h, w, _ = radiance_sum.shape
        out = np.zeros((h, w, 3), np.uint8)
        _check(_hip.rt_resolve(self.gpu, _ptr(radiance_sum), w, h, spp, gamma, _ptr(out)))
        return out

    def render_image(self, view16, fov_deg, width, height, spp, max_depth, seed=0, bmp_path=None):
        """Renderer::render_image through the C++ host classes -> (u8 image, fp32 sums, stats)."""
        rgb = np.zeros((height, width, 3), np.uint8)
        rad = np.zeros((height, width, 3), np.float32)
        stats = RtStats()
        ok = self._host.rth_render_image(self._h, _ptr(_f32(view16, 16)), fov_deg, width, height, spp, max_depth, seed,
                                         os.fsencode(bmp_path) if bmp_path else None, _ptr(rgb), _ptr(rad), C.byref(stats))
        if not ok:
            raise RtError("render_image failed: " + last_error())
        return rgb, rad, stats.as_dict()

    def render_image_sharded(self, view16, fov_deg, width, height, spp, max_depth, devices, sharding=0, tile_size=64, seed=0):
        """Renderer::render_image with Options::devices = devices (one host thread and one scene replica per entry; a device
        listed twice rehearses on one GPU), sharding 0 = sample windows of the one spp-sample set, 1 = interleaved tiles;
        the frames meet in rt_reduce.  -> (u8 image, fp32 sums, stats)."""
        rgb = np.zeros((height, width, 3), np.uint8)
        rad = np.zeros((height, width, 3), np.float32)
        stats = RtStats()
        dev = np.ascontiguousarray(devices, dtype=np.int32)
        ok = self._host.rth_render_image_sharded(self._h, _ptr(_f32(view16, 16)), fov_deg, width, height, spp, max_depth, seed, _ptr(dev),
                                                 len(dev), int(sharding), int(tile_size), _ptr(rgb), _ptr(rad), C.byref(stats))
        if not ok:
            raise RtError("render_image_sharded failed: " + last_error())
        return rgb, rad, stats.as_dict()

    def closest_hit(self, ray):
        """Scene::closest_hit for one ray -> dict or None."""
        ray = np.ascontiguousarray(ray, dtype=RAY).reshape(1)
        t = np.zeros(1, np.float32)
        p, n = np.zeros(3, np.float32), np.zeros(3, np.float32)
        ff = np.zeros(1, np.int32)
        if not self._host.rth_scene_closest_hit(self._h, _ptr(ray), _ptr(t), _ptr(p), _ptr(n), _ptr(ff)):
            return None
        return {"t": float(t[0]), "point": p, "normal": n, "front_face": bool(ff[0] & 1), "index": int(ff[0] >> 1)}

    def any_hit(self, ray):
        ray = np.ascontiguousarray(ray, dtype=RAY).reshape(1)
        return bool(self._host.rth_scene_any_hit(self._h, _ptr(ray)))
