"""Scene builders shared by the tests, smoke() and bench.py (synthetic data only)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CORNELL_GLB = os.path.join(ROOT, "tests", "golden", "cornellbox.glb")


def _quad(rta, corners, normal, material):
    v = np.zeros(4, rta.VERTEX)
    v["pos"] = np.asarray(corners, np.float32)
    v["normal"] = np.asarray(normal, np.float32)
    v["uv"] = [(0, 0), (1, 0), (1, 1), (0, 1)]
    t = np.zeros(2, rta.TRIANGLE)
    t["idx"] = [(0, 1, 2), (0, 2, 3)]
    t["material"] = material
    return v, t


def _box(rta, lo, hi, material):
    lo, hi = np.asarray(lo, np.float32), np.asarray(hi, np.float32)
    x0, y0, z0 = lo
    x1, y1, z1 = hi
    faces = [
        ([(x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1)], (0, 0, 1)),
        ([(x1, y0, z0), (x0, y0, z0), (x0, y1, z0), (x1, y1, z0)], (0, 0, -1)),
        ([(x1, y0, z1), (x1, y0, z0), (x1, y1, z0), (x1, y1, z1)], (1, 0, 0)),
        ([(x0, y0, z0), (x0, y0, z1), (x0, y1, z1), (x0, y1, z0)], (-1, 0, 0)),
        ([(x0, y1, z1), (x1, y1, z1), (x1, y1, z0), (x0, y1, z0)], (0, 1, 0)),
        ([(x0, y0, z0), (x1, y0, z0), (x1, y0, z1), (x0, y0, z1)], (0, -1, 0)),
    ]
    return [_quad(rta, c, n, material) for c, n in faces]


def quad_room(rta, box_material=None, light=True, env=0.05, tall_material=None):
    """A 2x2x2 Cornell-style room (open towards +z) with a rectangle light under the ceiling, one box with a
    configurable material (default GGX roughness 0.4) and a second, tall, Lambert box.  82 triangles."""
    sc = rta.Scene()
    white = sc.add_material(rta.material(rta.MAT_MATTE, (0.725, 0.71, 0.68)))
    red = sc.add_material(rta.material(rta.MAT_MATTE, (0.63, 0.065, 0.05)))
    green = sc.add_material(rta.material(rta.MAT_MATTE, (0.14, 0.45, 0.091)))
    boxm = sc.add_material(box_material if box_material is not None else rta.material(rta.MAT_GGX, s=(0.4, 0.0)))
    tallm = sc.add_material(tall_material if tall_material is not None else rta.material(rta.MAT_MATTE, (0.6, 0.6, 0.7)))
    parts = [
        _quad(rta, [(-1, 0, 1), (1, 0, 1), (1, 0, -1), (-1, 0, -1)], (0, 1, 0), white),     # floor
        _quad(rta, [(-1, 2, -1), (1, 2, -1), (1, 2, 1), (-1, 2, 1)], (0, -1, 0), white),    # ceiling
        _quad(rta, [(-1, 0, -1), (1, 0, -1), (1, 2, -1), (-1, 2, -1)], (0, 0, 1), white),   # back
        _quad(rta, [(-1, 0, 1), (-1, 0, -1), (-1, 2, -1), (-1, 2, 1)], (1, 0, 0), red),     # left
        _quad(rta, [(1, 0, -1), (1, 0, 1), (1, 2, 1), (1, 2, -1)], (-1, 0, 0), green),      # right
    ]
    parts += _box(rta, (0.15, 0.0, -0.1), (0.7, 0.6, 0.45), boxm)
    parts += _box(rta, (-0.7, 0.0, -0.65), (-0.15, 1.2, -0.1), tallm)
    # a strip of many small triangles on the floor so the BVH has some depth
    n = 6
    for i in range(n):
        for j in range(n):
            x0, z0 = -0.9 + 0.1 * i, 0.3 + 0.1 * j
            parts.append(_quad(rta, [(x0, 0.01, z0 + 0.08), (x0 + 0.08, 0.01, z0 + 0.08), (x0 + 0.08, 0.01, z0), (x0, 0.01, z0)],
                               (0, 1, 0), white if (i + j) % 2 else red))
    for v, t in parts:
        sc.add_mesh(v, t)
    if light:
        m = np.eye(4, dtype=np.float32)
        m[:3, 3] = (0.0, 1.98, 0.0)
        sc.add_rect_light((0, -1, 0), 0.6, 0.5, (17, 12, 4), m.T.reshape(-1))  # column-major
    if env is not None:
        sc.set_env(np.full((8, 16, 3), env, np.float32))
    return sc


def random_triangles(rta, n, seed=1234, material=None):
    """BASELINE config 3 geometry: n triangles, centres uniform in [0,1]^3, vertices = centre + U(-s/2, s/2)^3,
    s = 0.5 n^(-1/3); one Lambert material, one rect light at the box top."""
    rng = np.random.default_rng(seed)
    s = 0.5 * n ** (-1.0 / 3.0)
    c = rng.random((n, 1, 3), dtype=np.float32)
    p = (c + (rng.random((n, 3, 3), dtype=np.float32) - 0.5) * s).astype(np.float32)
    v = np.zeros(3 * n, rta.VERTEX)
    v["pos"] = p.reshape(-1, 3)
    e1, e2 = p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]
    nn = np.cross(e1, e2)
    nn /= np.maximum(np.linalg.norm(nn, axis=1, keepdims=True), 1e-20)
    v["normal"] = np.repeat(nn, 3, axis=0)
    t = np.zeros(n, rta.TRIANGLE)
    t["idx"] = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    sc = rta.Scene()
    sc.add_material(material if material is not None else rta.material(rta.MAT_MATTE, (0.7, 0.7, 0.7)))
    sc.add_mesh(v, t)
    m = np.eye(4, dtype=np.float32)
    m[:3, 3] = (0.5, 1.2, 0.5)
    sc.add_rect_light((0, -1, 0), 0.5, 0.5, (20, 20, 20), m.T.reshape(-1))
    return sc


def uv_sphere(rta, n_lat, n_lon, material=0):
    """Indexed unit sphere: 2 * n_lon * (n_lat - 1) triangles, smooth normals."""
    th = np.linspace(0.0, np.pi, n_lat + 1)[1:-1]
    ph = np.linspace(0.0, 2.0 * np.pi, n_lon, endpoint=False)
    ring = np.stack([np.outer(np.sin(th), np.cos(ph)), np.outer(np.cos(th), np.ones_like(ph)), np.outer(np.sin(th), np.sin(ph))], -1)
    pos = np.concatenate([[[0.0, 1.0, 0.0]], ring.reshape(-1, 3), [[0.0, -1.0, 0.0]]]).astype(np.float32)
    v = np.zeros(len(pos), rta.VERTEX)
    v["pos"] = pos
    v["normal"] = pos
    idx = []
    r = lambda i, j: 1 + i * n_lon + (j % n_lon)
    for j in range(n_lon):
        idx.append((0, r(0, j + 1), r(0, j)))
        idx.append((len(pos) - 1, r(n_lat - 2, j), r(n_lat - 2, j + 1)))
    for i in range(n_lat - 2):
        for j in range(n_lon):
            idx.append((r(i, j), r(i, j + 1), r(i + 1, j + 1)))
            idx.append((r(i, j), r(i + 1, j + 1), r(i + 1, j)))
    t = np.zeros(len(idx), rta.TRIANGLE)
    t["idx"] = np.asarray(idx, np.uint32)
    t["material"] = material
    return v, t


def instance_transforms(copies, seed=99, lo=(-1.4, 0.3, -1.4), hi=(1.4, 3.0, 1.4), scale=(0.04, 0.12)):
    """Column-major model-to-world matrices: random rotation, uniform scale in `scale`, translation in the box."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(copies):
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        if np.linalg.det(q) < 0:
            q[:, 0] = -q[:, 0]
        m = np.eye(4)
        m[:3, :3] = q * rng.uniform(*scale)
        m[:3, 3] = np.asarray(lo) + rng.random(3) * (np.asarray(hi) - np.asarray(lo))
        out.append(m.T.reshape(-1).astype(np.float32))
    return out


def cornell_with_instances(rta, n_lat, n_lon, copies, seed=99):
    """BASELINE config 5 in shape: cornellbox.glb plus ONE mesh (a UV sphere of 2 n_lon (n_lat - 1) triangles) instanced
    `copies` times inside the room (100 000 triangles x 50 = the configuration's 5 M), flattened at scene set-up."""
    sc = cornellbox(rta)
    mat = sc.add_material(rta.material(rta.MAT_GGX, s=(0.35, 0.0)))
    v, t = uv_sphere(rta, n_lat, n_lon)
    for m in instance_transforms(copies, seed):
        sc.add_instance(v, t, m, material=mat)
    return sc


def sphere_field(rta, n_spheres=10_000, n_lat=16, n_lon=32, seed=1234, grid=(22, 22, 21)):
    """BASELINE config 4 geometry: n_spheres tessellated spheres (2 n_lon (n_lat - 1) triangles each: 960 -> 9.6 M for
    10 000) on a jittered 22 x 22 x 21 grid in the unit box, one Lambert and one GGX material, a rectangle light on top."""
    rng = np.random.default_rng(seed)
    sc = rta.Scene()
    sc.add_material(rta.material(rta.MAT_MATTE, (0.7, 0.7, 0.7)))
    sc.add_material(rta.material(rta.MAT_GGX, s=(0.4, 0.0)))
    v, t = uv_sphere(rta, n_lat, n_lon)
    nx, ny, nz = grid  # (a smaller grid = fewer, larger spheres: the same picture in small)
    cells = rng.permutation(nx * ny * nz)[:n_spheres]
    for k, c in enumerate(cells):
        ix, iy, iz = c % nx, (c // nx) % ny, c // (nx * ny)
        centre = (np.array([ix / nx, iy / ny, iz / nz]) + (0.25 + 0.5 * rng.random(3)) / np.array([nx, ny, nz]))
        m = np.eye(4)
        m[:3, :3] *= 0.2 / nx * (0.6 + 0.4 * rng.random())
        m[:3, 3] = centre
        sc.add_instance(v, t, m.T.reshape(-1).astype(np.float32), material=k & 1)
    lm = np.eye(4, dtype=np.float32)
    lm[:3, 3] = (0.5, 1.2, 0.5)
    sc.add_rect_light((0, -1, 0), 0.5, 0.5, (20, 20, 20), lm.T.reshape(-1))
    return sc


def random_triangles_camera(rta, width, height):
    """Config 3 view: the unit box of random_triangles() seen from +z."""
    return rta.camera(rta.look_at((0.5, 0.5, 2.6), (0.5, 0.5, 0.5), (0, 1, 0)), 39.6, width, height)


def cornellbox(rta, mode=None, emissive_scale=15.0, env=0.02):
    """cornellbox.glb (data asset of the reference, committed under tests/golden/) through Scene::load_from_gltf.
    mode: a raytracing_amd GLTF_* material mode (default GLTF_GGX: the BASELINE config-2 GGX + Beckmann mix)."""
    sc = rta.Scene()
    sc.load_gltf(CORNELL_GLB, rta.GLTF_GGX if mode is None else mode, emissive_scale)
    if env is not None:
        sc.set_env(np.full((8, 16, 3), env, np.float32))
    return sc


def cornell_camera(rta, sc, width, height):
    info = sc.info()
    return rta.camera(info["camera_view"], info["camera_fov"], width, height)


def random_rays(n, seed, lo=(-0.2, -0.2, -0.2), hi=(1.2, 1.2, 1.2)):
    """Incoherent rays: origins uniform in a box, directions uniform on the sphere."""
    rng = np.random.default_rng(seed)
    rays = np.zeros(n, np.dtype([("o", "<f4", 3), ("d", "<f4", 3)]))
    rays["o"] = (np.asarray(lo) + rng.random((n, 3)) * (np.asarray(hi) - np.asarray(lo))).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays["d"] = d.astype(np.float32)
    return rays


def _trs(translate=(0, 0, 0), scale=1.0, rot_y_deg=0.0, rot_x_deg=0.0):
    """Column-major model-to-world: translate * rot_y * rot_x * scale."""
    cy, sy = np.cos(np.radians(rot_y_deg)), np.sin(np.radians(rot_y_deg))
    cx, sx = np.cos(np.radians(rot_x_deg)), np.sin(np.radians(rot_x_deg))
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    m = np.eye(4)
    m[:3, :3] = ry @ rx * np.asarray(scale, np.float64)
    m[:3, 3] = translate
    return m.T.reshape(-1).astype(np.float32)


def legacy_two_level(rta, env=0.05, light=True, mesh_lat=12, mesh_lon=24, mesh_copies=3, extra_spheres=0, seed=5, mesh_max_per_leaf=1):
    """A scene of the legacy design, in the shape of the reference's BallTestScene / Conference scripts
    (main.cpp:725-833): Scene::instances of Rectangle, Sphere, Box, Cylinder, Disk and one TriangleMesh placed several
    times with scale + rotation + translation, a Rectangle light in both `instances` and `lights`, under a TLAS."""
    sc = rta.Scene()
    floor = sc.add_material(rta.material(rta.MAT_DISNEY_DIFFUSE, (0.7, 0.7, 0.7), s=(0.4, 0.5)))
    red = sc.add_material(rta.material(rta.MAT_DISNEY_UBER, (0.6, 0.05, 0.05), s=(0.0, 0.0, 0.0, 1.0, 0.1, 0.0, 0.0, 0, 0, 0, 0, 1.5)))
    glass = sc.add_material(rta.material(rta.MAT_DISNEY_GLASS, (1, 1, 1), s=(0.05, 0.0, 1.5)))
    metal = sc.add_material(rta.material(rta.MAT_DISNEY_METAL, (0.7, 0.7, 0.7), s=(0.2, 0.0)))
    ggx = sc.add_material(rta.material(rta.MAT_GGX, s=(0.4, 0.0)))
    matte = sc.add_material(rta.material(rta.MAT_MATTE, (0.2, 0.4, 0.8)))
    emit = sc.add_material(rta.material(rta.MAT_EMISSIVE, (17, 12, 4)))
    sc.add_primitive_instance(rta.GEO_RECTANGLE, (0, 1, 0, 3.0, 3.0), floor, _trs())
    for k, (z, mat) in enumerate([(0.0, red), (0.25, glass), (-0.25, metal), (-0.5, floor), (0.5, ggx)]):
        sc.add_primitive_instance(rta.GEO_SPHERE, (0.1,), mat, _trs((0.0, 0.1, z)))
    sc.add_primitive_instance(rta.GEO_BOX, (0.2, 0.3, 0.15), matte, _trs((0.45, 0.15, 0.3), rot_y_deg=25.0))
    sc.add_primitive_instance(rta.GEO_CYLINDER, (0.06, 0.3), ggx, _trs((0.45, 0.15, -0.3), rot_x_deg=10.0))
    sc.add_primitive_instance(rta.GEO_DISK, (0, 1, 0, 0.2, 0.07), metal, _trs((-0.4, 0.25, 0.0)))
    if light:
        sc.add_primitive_instance(rta.GEO_RECTANGLE, (0, -1, 0, 0.3, 1.3), emit, _trs((0.0, 0.9, 0.0)), light=True)
    if mesh_copies:
        v, t = uv_sphere(rta, mesh_lat, mesh_lon)
        v["pos"][:, 1] *= 1.6  # an ellipsoid, so rotations matter
        mesh = sc.new_triangle_mesh(v, t, mesh_max_per_leaf)
        rng = np.random.default_rng(seed)
        for k in range(mesh_copies):
            m = _trs((-0.35 + 0.1 * k, 0.2 + 0.12 * k, -0.6 + 0.55 * k), scale=0.07 + 0.02 * k, rot_y_deg=float(rng.uniform(0, 360)),
                     rot_x_deg=float(rng.uniform(-40, 40)))
            sc.add_mesh_instance(mesh, [matte, red, metal][k % 3], m)
    rng = np.random.default_rng(seed + 1)
    for _ in range(extra_spheres):
        p = rng.uniform((-1.2, 0.05, -1.2), (1.2, 0.8, 1.2))
        sc.add_primitive_instance(rta.GEO_SPHERE, (float(rng.uniform(0.02, 0.06)),), int(rng.integers(0, 6)), _trs(p))
    if env:
        sc.set_env(np.full((8, 16, 3), env, np.float32))
    return sc


def legacy_camera(rta, width, height):
    return rta.camera(rta.look_at((-1.4, 0.7, 0.0), (0.0, 0.2, 0.0), (0, 1, 0)), 45.0, width, height)


def legacy_instanced(rta, n_lat=159, n_lon=316, copies=50, seed=99, env=0.05, max_per_leaf=1):
    """BASELINE config 5 as the legacy two-level design holds it: ONE mesh (a UV sphere of 2 n_lon (n_lat - 1) triangles:
    99 856 by default) placed `copies` times by Instance transforms — one BVH of 100k triangles in memory instead of the
    5 M flattened ones — inside a room of five Rectangle instances, under a Rectangle light."""
    sc = rta.Scene()
    white = sc.add_material(rta.material(rta.MAT_DISNEY_DIFFUSE, (0.725, 0.71, 0.68), s=(0.5, 0.5)))
    red = sc.add_material(rta.material(rta.MAT_MATTE, (0.63, 0.065, 0.05)))
    green = sc.add_material(rta.material(rta.MAT_MATTE, (0.14, 0.45, 0.091)))
    ggx = sc.add_material(rta.material(rta.MAT_GGX, s=(0.35, 0.0)))
    metal = sc.add_material(rta.material(rta.MAT_DISNEY_METAL, (0.8, 0.6, 0.3), s=(0.3, 0.0)))
    emit = sc.add_material(rta.material(rta.MAT_EMISSIVE, (17, 12, 4)))
    R = rta.GEO_RECTANGLE
    sc.add_primitive_instance(R, (0, 1, 0, 3.2, 3.2), white, _trs((0, 0, 0)))
    sc.add_primitive_instance(R, (0, -1, 0, 3.2, 3.2), white, _trs((0, 3.2, 0)))
    sc.add_primitive_instance(R, (0, 0, 1, 3.2, 3.2), white, _trs((0, 1.6, -1.6)))
    sc.add_primitive_instance(R, (1, 0, 0, 3.2, 3.2), red, _trs((-1.6, 1.6, 0)))
    sc.add_primitive_instance(R, (-1, 0, 0, 3.2, 3.2), green, _trs((1.6, 1.6, 0)))
    sc.add_primitive_instance(R, (0, -1, 0, 0.8, 0.8), emit, _trs((0, 3.19, 0)), light=True)
    v, t = uv_sphere(rta, n_lat, n_lon)
    mesh = sc.new_triangle_mesh(v, t, max_per_leaf)
    for k, m in enumerate(instance_transforms(copies, seed, scale=(0.12, 0.3))):
        sc.add_mesh_instance(mesh, [ggx, metal, white][k % 3], m)
    if env:
        sc.set_env(np.full((8, 16, 3), env, np.float32))
    return sc


def legacy_instanced_camera(rta, width, height):
    return rta.camera(rta.look_at((0.0, 1.6, 5.4), (0.0, 1.5, 0.0), (0, 1, 0)), 40.0, width, height)


def legacy_instanced_flat(rta, n_lat=159, n_lon=316, copies=50, seed=99, env=0.05):
    """The geometry of legacy_instanced() flattened at set-up into ONE world-space mesh (Scene::add_instance): what the
    target design does with config 5.  Same surfaces, materials, light and camera, 5 M triangles in memory."""
    sc = rta.Scene()
    white = sc.add_material(rta.material(rta.MAT_DISNEY_DIFFUSE, (0.725, 0.71, 0.68), s=(0.5, 0.5)))
    red = sc.add_material(rta.material(rta.MAT_MATTE, (0.63, 0.065, 0.05)))
    green = sc.add_material(rta.material(rta.MAT_MATTE, (0.14, 0.45, 0.091)))
    ggx = sc.add_material(rta.material(rta.MAT_GGX, s=(0.35, 0.0)))
    metal = sc.add_material(rta.material(rta.MAT_DISNEY_METAL, (0.8, 0.6, 0.3), s=(0.3, 0.0)))
    a = 1.6
    for corners, n, mat in [
        ([(-a, 0, a), (a, 0, a), (a, 0, -a), (-a, 0, -a)], (0, 1, 0), white),
        ([(-a, 2 * a, -a), (a, 2 * a, -a), (a, 2 * a, a), (-a, 2 * a, a)], (0, -1, 0), white),
        ([(-a, 0, -a), (a, 0, -a), (a, 2 * a, -a), (-a, 2 * a, -a)], (0, 0, 1), white),
        ([(-a, 0, a), (-a, 0, -a), (-a, 2 * a, -a), (-a, 2 * a, a)], (1, 0, 0), red),
        ([(a, 0, -a), (a, 0, a), (a, 2 * a, a), (a, 2 * a, -a)], (-1, 0, 0), green),
    ]:
        sc.add_mesh(*_quad(rta, corners, n, mat))
    sc.add_rect_light((0, -1, 0), 0.8, 0.8, (17, 12, 4), _trs((0, 3.19, 0)))
    v, t = uv_sphere(rta, n_lat, n_lon)
    for k, m in enumerate(instance_transforms(copies, seed, scale=(0.12, 0.3))):
        sc.add_instance(v, t, m, material=[ggx, metal, white][k % 3])
    if env:
        sc.set_env(np.full((8, 16, 3), env, np.float32))
    return sc
