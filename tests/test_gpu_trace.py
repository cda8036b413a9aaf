"""-m gpu: closest-hit / any-hit parity of the HIP traversal kernels against the oracle, through the C-ABI.
Bar: triangle ids and occlusion flags bit-exact; t, u, v bit-exact too (same fp32 recipe, fp-contract off) —
stricter than the north star's 1e-5 relative."""
import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu


def _secondary_rays(orc, osc, rays, hits, seed):
    """Bounce rays leaving the primary hit points in random directions (incoherent)."""
    ok = hits["tri"] >= 0
    p = rays["o"][ok] + rays["d"][ok] * hits["t"][ok, None]
    rng = np.random.default_rng(seed)
    d = rng.normal(size=p.shape)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    out = np.zeros(len(p), orc.RAY)
    out["o"] = (p + 1e-3 * d).astype(np.float32)
    out["d"] = d.astype(np.float32)
    return out


def _check_closest(sc, orc, osc, rays):
    got = sc.trace_closest(rays)
    want, _ = orc.closest_hit(osc, rays)
    assert np.array_equal(got["tri"], want["tri"]), f"{np.count_nonzero(got['tri'] != want['tri'])} triangle ids differ"
    hit = want["tri"] >= 0
    for f in ("t", "u", "v"):
        assert np.array_equal(got[f][hit], want[f][hit]), f"{f} differs"
    return want


@pytest.fixture(scope="module")
def room(rta, orc, gpu):
    sc = scenes.quad_room(rta)
    sc.build()
    sc.upload(gpu)
    return sc, orc.scene_from_host(sc, own_bvh=True)


@pytest.fixture(scope="module")
def cornell(rta, orc, gpu):
    sc = scenes.cornellbox(rta)
    sc.build()
    sc.upload(gpu)
    return sc, orc.scene_from_host(sc)


def test_closest_primary_and_secondary_room(rta, orc, room):
    sc, osc = room
    cam = rta.camera(rta.look_at((0.0, 1.0, 3.2), (0.0, 1.0, 0.0), (0, 1, 0)), 45.0, 96, 96)
    rays = orc.camera_rays(cam, 96, 96, 3, 16)
    hits = _check_closest(sc, orc, osc, rays)
    assert (hits["tri"] >= 0).mean() > 0.5
    _check_closest(sc, orc, osc, _secondary_rays(orc, osc, rays, hits, 1))


def test_closest_cornellbox(rta, orc, cornell):
    sc, osc = cornell
    cam = scenes.cornell_camera(rta, sc, 128, 128)
    rays = orc.camera_rays(cam, 128, 128, 0, 4)
    hits = _check_closest(sc, orc, osc, rays)
    sec = _secondary_rays(orc, osc, rays, hits, 2)
    hits2 = _check_closest(sc, orc, osc, sec)
    _check_closest(sc, orc, osc, _secondary_rays(orc, osc, sec, hits2, 3))
    _check_closest(sc, orc, osc, scenes.random_rays(20000, 5, lo=(-1.7, 0, -1.7), hi=(1.7, 3.4, 1.7)))


def test_any_hit(rta, orc, cornell):
    sc, osc = cornell
    rays = scenes.random_rays(30000, 11, lo=(-1.7, 0, -1.7), hi=(1.7, 3.4, 1.7))
    rng = np.random.default_rng(12)
    tmax = rng.random(len(rays)).astype(np.float32) * 3.0
    tmax[::7] = np.inf
    got = sc.trace_any(rays, tmax)
    want = orc.any_hit(osc, rays, tmax)
    assert np.array_equal(got, want)
    assert 0.05 < want.mean() < 0.95


def test_edge_cases(rta, orc, gpu):
    """One-triangle scene (root is a leaf), axis-parallel rays (inf/NaN slabs), rays starting on a surface,
    empty batch, leaves with several triangles."""
    sc = rta.Scene()
    sc.add_material(rta.material(rta.MAT_MATTE, (0.5, 0.5, 0.5)))
    v = np.zeros(3, rta.VERTEX)
    v["pos"] = [(-1, -1, -4), (1, -1, -4), (0, 1, -4)]
    v["normal"] = (0, 0, 1)
    t = np.zeros(1, rta.TRIANGLE)
    t["idx"] = [(0, 1, 2)]
    sc.add_mesh(v, t)
    sc.build()
    sc.upload(gpu)
    osc = orc.scene_from_host(sc)
    rays = np.zeros(3, rta.RAY)
    rays["d"] = [(0, 0, -1), (0, 0, 1), (0.3, 0, -1)]
    hits = _check_closest(sc, orc, osc, rays)
    assert hits["tri"][0] == 0 and hits["t"][0] == 4.0 and hits["tri"][1] == -1
    assert len(sc.trace_closest(np.zeros(0, rta.RAY))) == 0

    room = scenes.quad_room(rta)
    room.build(max_per_leaf=4, strategy=0)  # BVH_MIDDLE, multi-triangle leaves
    room.upload(gpu)
    oroom = orc.scene_from_host(room)
    rays = scenes.random_rays(5000, 3, lo=(-1, 0, -1), hi=(1, 2, 1))
    rays["d"][::5] = (0, -1, 0)   # axis-parallel: 1/0 = inf in the slab test
    rays["d"][1::5] = (1, 0, 0)
    rays["o"][2::5, 1] = 0.0      # origins lying in the floor plane
    _check_closest(room, orc, oroom, rays)
    tmax = np.full(len(rays), np.inf, np.float32)
    assert np.array_equal(room.trace_any(rays, tmax), orc.any_hit(oroom, rays, tmax))


@pytest.mark.parametrize("chunk", [None, 192])
def test_ragged_ray_counts_every_ray_gets_its_own_answer(rta, orc, gpu, chunk):
    """rt_trace_* deal the caller's array out to the 64 queue shards in chunks of RT_TUNE_FETCH_CHUNK rays (default 128; 192: not a
    power of two): counts around one chunk, one round of chunks (64 x 128 = 8192) and several rounds, each against the oracle."""
    import os
    if chunk:
        os.environ["RT_TUNE_FETCH_CHUNK"] = str(chunk)
    try:
        sc = scenes.quad_room(rta)
        sc.build()
        sc.upload(gpu)
    finally:
        os.environ.pop("RT_TUNE_FETCH_CHUNK", None)
    osc = orc.scene_from_host(sc, own_bvh=True)
    rays = scenes.random_rays(40_000, 21, lo=(-1, 0, -1), hi=(1, 2, 1))
    tmax = (np.random.default_rng(22).random(len(rays)) * 2.5).astype(np.float32)
    want, _ = orc.closest_hit(osc, rays)
    want_any = orc.any_hit(osc, rays, tmax)
    for n in (1, 2, 63, 64, 65, 127, 128, 129, 191, 192, 193, 8191, 8192, 8193, 12288, 12289, 24575, 33000, 40_000):
        got = sc.trace_closest(rays[:n])
        assert got.tobytes() == want[:n].tobytes(), n
        assert np.array_equal(sc.trace_any(rays[:n], tmax[:n]), want_any[:n]), n


def test_host_scene_api(rta, orc, room):
    """Scene::closest_hit / Scene::any_hit (the reference's query seam) for single rays."""
    sc, osc = room
    ray = np.zeros(1, rta.RAY)
    ray["o"] = (0, 1, 3.2)
    ray["d"] = (0, 0, -1)
    hit = sc.closest_hit(ray)
    want, _ = orc.closest_hit(osc, ray)
    assert hit is not None and hit["t"] == want["t"][0]
    assert hit["front_face"] and np.allclose(hit["normal"], (0, 0, 1))
    assert sc.any_hit(ray)
    ray["d"] = (0, 0, 1)
    assert sc.closest_hit(ray) is None and not sc.any_hit(ray)


def test_light_directed_segments_match_the_reference_walk(rta, orc, cornell):
    """Shadow segments as next-event estimation makes them: from surface points towards points on the emissive mesh,
    ending 1e-3 in front of it (main.cpp:278-281).  The kernels skip boxes entered beyond the segment's end; the oracle
    walks like the reference (no such skip).  The flags must agree for every one of the ~1.7 million segments."""
    sc, osc = cornell
    tris, verts = sc.array(rta.ARR_TRIS), sc.array(rta.ARR_VERTS)
    emitters = np.flatnonzero(tris["material"] == 5)  # Suzanne (material 5 of cornellbox.glb)
    assert len(emitters) == 968
    rng = np.random.default_rng(61)
    n = 2_000_000
    # surface points: first hits of random rays inside the room
    prim = scenes.random_rays(n, 62, lo=(-1.6, 0.1, -1.6), hi=(1.6, 3.3, 1.6))
    hits, _ = orc.closest_hit(osc, prim)
    ok = hits["tri"] >= 0
    prim, hits = prim[ok], hits[ok]
    p = prim["o"] + prim["d"] * hits["t"][:, None]
    # points on the light: uniform barycentrics on random emissive triangles
    pick = emitters[rng.integers(0, len(emitters), len(p))]
    pv = verts["pos"][tris["idx"][pick].astype(np.int64)]
    b = rng.random((len(p), 2)).astype(np.float32)
    su = np.sqrt(b[:, 0])
    u, v = 1.0 - su, b[:, 1] * su
    target = pv[:, 0] + u[:, None] * (pv[:, 1] - pv[:, 0]) + v[:, None] * (pv[:, 2] - pv[:, 0])
    d = target - p
    length = np.linalg.norm(d, axis=1).astype(np.float32)
    seg = np.zeros(len(p), orc.RAY)
    seg["o"] = (p - 1e-4 * prim["d"]).astype(np.float32)
    seg["d"] = (d / length[:, None]).astype(np.float32)
    tmax = (length - np.float32(0.001)).astype(np.float32)
    got = sc.trace_any(seg, tmax)
    want = orc.any_hit(osc, seg, tmax)
    assert np.array_equal(got, want), np.count_nonzero(got != want)
    assert 0.03 < want.mean() < 0.97  # both answers are well represented (1.7 M segments, ~10 % unoccluded)


def test_two_triangles_per_leaf_same_hits(rta, orc, gpu):
    """bench.py walks cornellbox under BVH::build(bounds, 2, BVH_SAH) (the reference's builder, max_per_node = 2): the hits
    are those of the one-triangle-per-leaf tree its TriangleMesh builds (Object.cpp:21), and the oracle's on either tree."""
    one, two = scenes.cornellbox(rta), scenes.cornellbox(rta)
    one.build(max_per_leaf=1)
    two.build(max_per_leaf=2)
    assert len(two.array(rta.ARR_FLAT_PAIRS)) < 0.75 * len(one.array(rta.ARR_FLAT_PAIRS))
    one.upload(gpu)
    two.upload(gpu)
    cam = scenes.cornell_camera(rta, one, 160, 90)
    rays = np.concatenate([orc.camera_rays(cam, 160, 90, 2, 8), scenes.random_rays(40000, 77, lo=(-1.7, 0, -1.7), hi=(1.7, 3.4, 1.7))])
    h1, h2 = one.trace_closest(rays), two.trace_closest(rays)
    ref, _ = orc.closest_hit(orc.scene_from_host(two), rays)
    assert h2.tobytes() == ref.tobytes()
    assert np.array_equal(h1["tri"], h2["tri"]) and np.array_equal(h1["t"], h2["t"]) and np.array_equal(h1["u"], h2["u"])
    tmax = (np.random.default_rng(5).random(len(rays)) * 3).astype(np.float32)
    assert np.array_equal(one.trace_any(rays, tmax), two.trace_any(rays, tmax))
    assert np.array_equal(two.trace_any(rays, tmax), orc.any_hit(orc.scene_from_host(two), rays, tmax))
    img1, _ = one.render(cam, 160, 90, spp=4, max_depth=6, seed=9)
    img2, _ = two.render(cam, 160, 90, spp=4, max_depth=6, seed=9)
    assert np.array_equal(img1, img2)  # the same paths, the same arithmetic: the tree changes no pixel
