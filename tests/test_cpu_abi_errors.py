"""CPU (-m "not gpu"): argument and description checks of the C-ABI (include/rt_hip.h).  Every reference a kernel would
follow is validated on the host before anything touches a GPU, so a bad description is an error code with a message,
never a GPU memory fault; and without a GPU a good description fails loudly (RT_ERR_NO_DEVICE): there is no CPU path."""
import ctypes as C

import numpy as np
import pytest

import scenes

RT_OK, RT_ERR_INVALID, RT_ERR_NO_DEVICE, RT_ERR_HIP, RT_ERR_LIMIT = 0, 1, 2, 3, 4


class SceneDesc(C.Structure):
    _fields_ = [("root_ref", C.c_int32), ("n_pairs", C.c_int64), ("pairs", C.c_void_p), ("n_tris", C.c_int64), ("tris", C.c_void_p),
                ("normals", C.c_void_p), ("n_materials", C.c_int32), ("materials", C.c_void_p), ("n_lights", C.c_int32),
                ("lights", C.c_void_p), ("n_emissive_tris", C.c_int32), ("emissive_tris", C.c_void_p), ("env_w", C.c_int32),
                ("env_h", C.c_int32), ("env_rgb", C.c_void_p), ("env_cdf", C.c_void_p), ("env_total", C.c_float), ("env_offset", C.c_float)]


class TlasDesc(C.Structure):
    _fields_ = [("root_ref", C.c_int32), ("n_pairs", C.c_int64), ("pairs", C.c_void_p), ("n_instances", C.c_int32), ("instances", C.c_void_p)]


def _flat(rta):
    sc = scenes.quad_room(rta)
    sc.build()
    a = {k: sc.array(getattr(rta, "ARR_" + k)) for k in ("FLAT_PAIRS", "FLAT_TRIS", "FLAT_NORMALS", "MATERIALS", "LIGHTS", "EMISSIVE")}
    d = SceneDesc()
    d.root_ref = sc.info()["root_ref"]
    d.n_pairs, d.pairs = len(a["FLAT_PAIRS"]), a["FLAT_PAIRS"].ctypes.data
    d.n_tris, d.tris, d.normals = len(a["FLAT_TRIS"]), a["FLAT_TRIS"].ctypes.data, a["FLAT_NORMALS"].ctypes.data
    d.n_materials, d.materials = len(a["MATERIALS"]), a["MATERIALS"].ctypes.data
    d.n_lights, d.lights = len(a["LIGHTS"]), a["LIGHTS"].ctypes.data
    d.n_emissive_tris, d.emissive_tris = len(a["EMISSIVE"]), (a["EMISSIVE"].ctypes.data if len(a["EMISSIVE"]) else None)
    return sc, a, d


def _two_level(rta):
    sc = scenes.legacy_two_level(rta, env=0)
    sc.build()
    a = {k: sc.array(getattr(rta, "ARR_" + k)) for k in ("FLAT_PAIRS", "FLAT_TRIS", "FLAT_NORMALS", "MATERIALS", "LIGHTS", "TLAS_PAIRS", "INSTANCES")}
    d = SceneDesc()
    d.n_pairs, d.pairs = len(a["FLAT_PAIRS"]), a["FLAT_PAIRS"].ctypes.data
    d.n_tris, d.tris, d.normals = len(a["FLAT_TRIS"]), a["FLAT_TRIS"].ctypes.data, a["FLAT_NORMALS"].ctypes.data
    d.n_materials, d.materials = len(a["MATERIALS"]), a["MATERIALS"].ctypes.data
    d.n_lights, d.lights = len(a["LIGHTS"]), a["LIGHTS"].ctypes.data
    t = TlasDesc()
    t.root_ref = 0
    t.n_pairs, t.pairs = len(a["TLAS_PAIRS"]), a["TLAS_PAIRS"].ctypes.data
    t.n_instances, t.instances = len(a["INSTANCES"]), a["INSTANCES"].ctypes.data
    return sc, a, d, t


def _create(rta, d, tl=None):
    hip, _ = rta.load_libraries()
    out = C.c_void_p()
    rc = hip.rt_scene_create(C.byref(d), 0, C.byref(out)) if tl is None else hip.rt_scene_create_instanced(C.byref(d), C.byref(tl), 0, C.byref(out))
    msg = rta.last_error()
    if rc == RT_OK:
        hip.rt_scene_destroy.argtypes = [C.c_void_p]
        hip.rt_scene_destroy(out)
    return rc, msg


def _good(rta, rc):
    """A valid description: created where a GPU is present, RT_ERR_NO_DEVICE (loudly) where none is."""
    return rc == (RT_OK if rta.device_count() > 0 else RT_ERR_NO_DEVICE)


def test_flat_description_checks(rta):
    sc, a, d = _flat(rta)
    rc, msg = _create(rta, d)
    assert _good(rta, rc), msg
    hip, _ = rta.load_libraries()
    assert hip.rt_scene_create(None, 0, None) == RT_ERR_INVALID

    def broken(field, value, where=None):
        sc2, a2, d2 = _flat(rta)
        if where is None:
            setattr(d2, field, value)
        else:
            a2[where][field][value[0]] = value[1]
        rc, msg = _create(rta, d2)
        return rc, msg

    assert broken("root_ref", 10 ** 6)[0] == RT_ERR_INVALID
    assert broken("n_tris", 0)[0] == RT_ERR_INVALID
    assert broken("n_materials", 0)[0] == RT_ERR_INVALID
    assert broken("n_lights", 16)[0] == RT_ERR_LIMIT
    rc, msg = broken("lref", (3, 10 ** 6), "FLAT_PAIRS")
    assert rc == RT_ERR_INVALID and "child ref" in msg
    rc, msg = broken("material", (5, 99), "FLAT_TRIS")
    assert rc == RT_ERR_INVALID and "material" in msg
    rc, msg = broken("tri_id", (len(a["FLAT_TRIS"]) - 1, 7), "FLAT_TRIS")  # the last triangle no longer ends a leaf
    assert rc == RT_ERR_INVALID and "leaf" in msg
    rc, msg = broken("type", (0, 42), "MATERIALS")
    assert rc == RT_ERR_INVALID and "material type" in msg
    rc, msg = broken("kind", (0, 9), "LIGHTS")
    assert rc == RT_ERR_INVALID and "light" in msg


def _chain(rta, depth):
    """A degenerate BVH: `depth` pair records in a chain (left child = a leaf, right child = the next record), one
    triangle per leaf — what a caller-built tree over sorted, nested primitives can look like."""
    n_tris = depth + 1
    pairs = np.zeros(depth, rta.PAIR)
    pairs["lmin"], pairs["lmax"], pairs["rmin"], pairs["rmax"] = -1, 1, -1, 1
    for k in range(depth):
        pairs["lref"][k] = ~k
        pairs["rref"][k] = k + 1 if k + 1 < depth else ~depth
    tris = np.zeros(n_tris, rta.FLAT_TRI)
    tris["e1"], tris["e2"] = (1, 0, 0), (0, 1, 0)
    tris["tri_id"] = np.arange(n_tris, dtype=np.uint32) | np.uint32(0x80000000)
    tris["light"] = -1
    normals = np.zeros(n_tris, rta.FLAT_NORMALS)
    mats = rta.material(rta.MAT_MATTE, (0.5, 0.5, 0.5))
    d = SceneDesc()
    d.root_ref = 0
    d.n_pairs, d.pairs = depth, pairs.ctypes.data
    d.n_tris, d.tris, d.normals = n_tris, tris.ctypes.data, normals.ctypes.data
    d.n_materials, d.materials = 1, mats.ctypes.data
    return (pairs, tris, normals, mats), d


def test_tree_depth_and_shape_checks(rta):
    """The walks keep the reference's 64-entry stack (Object.h:275): a tree deeper than that is refused with RT_ERR_LIMIT
    (it would have overrun the reference's own stack too), a record reached twice (cycle / shared subtree) with
    RT_ERR_INVALID — error codes on the host, not a wild read in a kernel."""
    keep, d = _chain(rta, 64)
    rc, msg = _create(rta, d)
    assert _good(rta, rc), msg
    keep, d = _chain(rta, 70)
    rc, msg = _create(rta, d)
    assert rc == RT_ERR_LIMIT and "deeper than 64" in msg
    keep, d = _chain(rta, 10)
    keep[0]["rref"][9] = 3  # a cycle
    rc, msg = _create(rta, d)
    assert rc == RT_ERR_INVALID and "tree" in msg
    keep, d = _chain(rta, 10)
    keep[0]["lref"][2] = 7  # record 7 has two parents
    rc, msg = _create(rta, d)
    assert rc == RT_ERR_INVALID and "tree" in msg
    # two-level: the TLAS may be 63 deep (its entries share the stack with a mesh's exit marker and the mesh's 64)
    sc, a, dd, t = _two_level(rta)
    deep = np.zeros(70, rta.PAIR)
    for k in range(70):
        deep["lref"][k] = ~0
        deep["rref"][k] = k + 1 if k + 1 < 70 else ~1
    t.n_pairs, t.pairs = 70, deep.ctypes.data
    rc, msg = _create(rta, dd, t)
    assert rc == RT_ERR_LIMIT and "TLAS deeper" in msg
    sc, a, dd, t = _two_level(rta)
    mesh_pairs = a["FLAT_PAIRS"]
    assert len(mesh_pairs) > 4
    mesh_pairs["rref"][len(mesh_pairs) - 1] = 0  # the mesh tree bites its own root
    rc, msg = _create(rta, dd, t)
    assert rc == RT_ERR_INVALID and "trees" in msg


def test_two_level_description_checks(rta):
    sc, a, d, t = _two_level(rta)
    rc, msg = _create(rta, d, t)
    assert _good(rta, rc), msg

    def broken(edit):
        sc2, a2, d2, t2 = _two_level(rta)
        edit(a2, d2, t2)
        return _create(rta, d2, t2)

    rc, msg = broken(lambda a2, d2, t2: setattr(t2, "n_instances", 0))
    assert rc == RT_ERR_INVALID and "instances" in msg
    rc, msg = broken(lambda a2, d2, t2: a2["TLAS_PAIRS"]["rref"].__setitem__(0, ~(10 ** 6)))
    assert rc == RT_ERR_INVALID and "TLAS child ref" in msg
    rc, msg = broken(lambda a2, d2, t2: setattr(t2, "root_ref", 10 ** 6))
    assert rc == RT_ERR_INVALID and "root_ref" in msg
    rc, msg = broken(lambda a2, d2, t2: a2["INSTANCES"]["geometry"].__setitem__(1, 17))
    assert rc == RT_ERR_INVALID and "geometry" in msg
    rc, msg = broken(lambda a2, d2, t2: a2["INSTANCES"]["material"].__setitem__(2, 99))
    assert rc == RT_ERR_INVALID and "material" in msg
    mesh_i = int(np.argmax(a["INSTANCES"]["geometry"] == 0))
    rc, msg = broken(lambda a2, d2, t2: a2["INSTANCES"]["blas_root"].__setitem__(mesh_i, 10 ** 6))
    assert rc == RT_ERR_INVALID and "blas_root" in msg
    rc, msg = broken(lambda a2, d2, t2: a2["LIGHTS"]["kind"].__setitem__(0, 1))
    assert rc == RT_ERR_INVALID and "rectangles" in msg
    hip, _ = rta.load_libraries()
    out = C.c_void_p()
    assert hip.rt_scene_create_instanced(C.byref(d), None, 0, C.byref(out)) == RT_ERR_INVALID


def test_calls_on_null_scene(rta):
    hip, _ = rta.load_libraries()
    rays = np.zeros(4, rta.RAY)
    hits = np.zeros(4, rta.HIT)
    assert hip.rt_trace_closest(None, rays.ctypes.data, 4, hits.ctypes.data) == RT_ERR_INVALID
    assert hip.rt_trace_closest_indexed(None, rays.ctypes.data, 4, hits.ctypes.data, None) == RT_ERR_INVALID
    assert hip.rt_trace_any(None, rays.ctypes.data, None, 4, None) == RT_ERR_INVALID
    assert hip.rt_render(None, None, None, None, None) == RT_ERR_INVALID
    assert hip.rt_resolve(None, None, 0, 0, 0, 2.2, None) == RT_ERR_INVALID
    assert hip.rt_scene_sync(None) == RT_ERR_INVALID
    assert "null" in rta.last_error() or "bad" in rta.last_error()


@pytest.mark.gpu
def test_render_argument_checks(rta, gpu):
    sc = scenes.quad_room(rta)
    sc.build()
    sc.upload(gpu)
    cam = rta.camera(rta.look_at((0.0, 1.0, 3.2), (0.0, 1.0, 0.0), (0, 1, 0)), 45.0, 32, 32)
    for kw, code in (({"tile": (0, 0, 40, 32)}, RT_ERR_INVALID), ({"tile": (8, 8, 8, 16)}, RT_ERR_INVALID), ({"spp_total": 2}, RT_ERR_INVALID),
                     ({"max_depth": 0}, RT_ERR_LIMIT), ({"max_depth": 64}, RT_ERR_LIMIT)):
        args = dict(spp=4, max_depth=4)
        args.update(kw)
        with pytest.raises(rta.RtError, match=f"rt error {code}"):
            sc.render(cam, 32, 32, **args)
    img, st = sc.render(cam, 32, 32, spp=4, max_depth=4)  # the handle is still usable after the refusals
    assert st["paths"] == 32 * 32 * 4 and np.isfinite(img).all()


@pytest.mark.gpu
def test_render_refuses_what_does_not_fit_the_gpu(rta, gpu):
    """A pass that would need more device memory than is free is an RT_ERR_LIMIT with the sizes in the message (2^30 paths
    per pass at ~400 B each), not a failed allocation half-way; the handle keeps working."""
    sc = scenes.quad_room(rta)
    sc.build()
    sc.upload(gpu)
    cam = rta.camera(rta.look_at((0.0, 1.0, 3.2), (0.0, 1.0, 0.0), (0, 1, 0)), 45.0, 8192, 8192)
    with pytest.raises(rta.RtError, match="rt error 4.*MiB"):
        sc.render(cam, 8192, 8192, spp=16, max_depth=2, spp_per_pass=15, device_out=1 << 20)
    small = rta.camera(rta.look_at((0.0, 1.0, 3.2), (0.0, 1.0, 0.0), (0, 1, 0)), 45.0, 32, 32)
    img, st = sc.render(small, 32, 32, spp=2, max_depth=2)
    assert st["paths"] == 2048


def _py_valid(a, root_ref):
    """An independent statement of what rt_scene_create must accept (include/rt_hip.h): every reference in range, the pair
    records a tree no deeper than 64 in which every record is reached once, leaves ending at a flagged triangle."""
    pairs, tris, mats, lights = a["FLAT_PAIRS"], a["FLAT_TRIS"], a["MATERIALS"], a["LIGHTS"]
    n_p, n_t = len(pairs), len(tris)
    if not (all(0 <= int(m) < len(mats) for m in tris["material"]) and all(int(l) < len(lights) for l in tris["light"])):  # (any negative light index means "no light")
        return False
    if not all(0 <= int(t) < 13 - 1 for t in mats["type"]):  # RT_KEY_MISS = 12: material types are 0..11
        return False
    for L in lights:
        if int(L["kind"]) != 0 or not (0 <= int(L["material"]) < len(mats)):  # RT_LIGHT_RECT in this scene
            return False
    if not (int(tris["tri_id"][-1]) & 0x80000000):
        return False

    def ok_ref(r):
        return (0 <= r < n_p) if r >= 0 else (0 <= ~r < n_t)

    if not ok_ref(int(root_ref)):
        return False
    seen, stack = set(), [(int(root_ref), 1)]
    while stack:
        r, depth = stack.pop()
        if r < 0:
            continue
        if r in seen or depth > 64:
            return False
        seen.add(r)
        for c in (int(pairs["lref"][r]), int(pairs["rref"][r])):
            if not ok_ref(c):
                return False
            stack.append((c, depth + 1))
    return True


def test_fuzzed_references_are_refused_or_harmless(rta):
    """400 random single-word corruptions of a valid description (child refs, material / light indices, leaf flags, material
    types, light records, the root): the call returns an error code, or — when an independent checker finds the description
    still valid — what a valid description returns.  It never accepts what the checker refuses: a kernel follows every one
    of these words."""
    rng = np.random.default_rng(20260211)
    sc, a0, d0 = _flat(rta)
    rc, _ = _create(rta, d0)
    assert _good(rta, rc) and _py_valid(a0, d0.root_ref)
    targets = [("FLAT_PAIRS", "lref"), ("FLAT_PAIRS", "rref"), ("FLAT_TRIS", "material"), ("FLAT_TRIS", "light"), ("FLAT_TRIS", "tri_id"),
               ("MATERIALS", "type"), ("LIGHTS", "kind"), ("LIGHTS", "material"), (None, "root_ref")]
    refused = 0
    for it in range(400):
        sc2, a, d = _flat(rta) if it % 50 == 0 else (sc2, a, d)  # a fresh description now and then; restored below otherwise
        arr, field = targets[int(rng.integers(len(targets)))]
        kind = int(rng.integers(4))
        n_p, n_t = len(a["FLAT_PAIRS"]), len(a["FLAT_TRIS"])
        value = [int(rng.integers(-2 ** 31, 2 ** 31 - 1)), int(rng.integers(-n_t - 2, n_p + 2)), int(rng.integers(0, 16)), -1][kind]
        if arr is None:
            old = d.root_ref
            d.root_ref = int(np.int32(value))
        else:
            i = int(rng.integers(len(a[arr])))
            old = a[arr][field][i].copy()
            a[arr][field][i] = np.array(value).astype(np.int64).astype(a[arr][field].dtype)
        rc, msg = _create(rta, d)
        valid = _py_valid(a, d.root_ref)
        assert rc in (RT_OK, RT_ERR_INVALID, RT_ERR_LIMIT, RT_ERR_NO_DEVICE), (rc, msg)
        if _good(rta, rc):
            assert valid, f"accepted a description the checker refuses: {arr}.{field} <- {value}"
        else:
            refused += 1
        if arr is None:
            d.root_ref = old
        else:
            a[arr][field][i] = old
    assert refused > 200


def test_fuzzed_two_level_references_are_refused_or_harmless(rta):
    """The same for rt_scene_create_instanced: TLAS child refs (leaves name instances), an instance's geometry kind,
    material and blas_root.  Out-of-range values must come back as error codes; values the independent range check allows
    may be accepted."""
    rng = np.random.default_rng(77)
    sc, a, d, t = _two_level(rta)
    rc, msg = _create(rta, d, t)
    assert _good(rta, rc), msg
    n_inst, n_tp, n_p, n_t, n_m = len(a["INSTANCES"]), len(a["TLAS_PAIRS"]), len(a["FLAT_PAIRS"]), len(a["FLAT_TRIS"]), len(a["MATERIALS"])

    def in_range():
        for r in list(a["TLAS_PAIRS"]["lref"]) + list(a["TLAS_PAIRS"]["rref"]):
            r = int(r)
            if not ((0 <= r < n_tp) if r >= 0 else (0 <= ~r < n_inst)):
                return False
        for I in a["INSTANCES"]:
            g, m, b = int(I["geometry"]), int(I["material"]), int(I["blas_root"])
            if not (0 <= g <= 5 and 0 <= m < n_m):
                return False
            if g == 0 and not ((0 <= b < n_p) if b >= 0 else (0 <= ~b < n_t)):
                return False
        return True

    targets = [("TLAS_PAIRS", "lref"), ("TLAS_PAIRS", "rref"), ("INSTANCES", "geometry"), ("INSTANCES", "material"), ("INSTANCES", "blas_root")]
    refused = 0
    for it in range(300):
        arr, field = targets[int(rng.integers(len(targets)))]
        value = [int(rng.integers(-2 ** 31, 2 ** 31 - 1)), int(rng.integers(-n_t - 2, n_p + 2)), int(rng.integers(0, 16)), -1][int(rng.integers(4))]
        i = int(rng.integers(len(a[arr])))
        old = a[arr][field][i].copy()
        a[arr][field][i] = np.array(value).astype(np.int64).astype(np.int32)
        rc, msg = _create(rta, d, t)
        assert rc in (RT_OK, RT_ERR_INVALID, RT_ERR_LIMIT, RT_ERR_NO_DEVICE), (rc, msg)
        if _good(rta, rc):
            assert in_range(), f"accepted an out-of-range reference: {arr}.{field}[{i}] <- {value}"
        else:
            refused += 1
        a[arr][field][i] = old
    assert refused > 100
