"""-m gpu: BASELINE config 3 geometry (random triangles in the unit box, incoherent rays).
  * 100 000 triangles, SAH tree: bit parity of closest-hit / any-hit with the oracle;
  * the configuration's full 1 000 000 triangles (BVH_MIDDLE tree: 1 s to build; the SAH build takes 40 s and is left to
    bench.py): size-independent properties — occlusion agrees with closest hit, results do not depend on the order of
    the rays in the batch, and a sample of rays agrees with a brute-force test of all triangles."""
import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu


def test_random_triangles_100k_bit_parity(rta, orc, gpu):
    sc = scenes.random_triangles(rta, 100_000)
    sc.build()
    sc.upload(gpu)
    osc = orc.scene_from_host(sc)
    rays = scenes.random_rays(60_000, 41)
    got = sc.trace_closest(rays)
    want, _ = orc.closest_hit(osc, rays)
    assert np.array_equal(got["tri"], want["tri"])
    hit = want["tri"] >= 0
    assert 0.05 < hit.mean() < 0.999
    for f in ("t", "u", "v"):
        assert np.array_equal(got[f][hit], want[f][hit]), f
    tmax = (np.random.default_rng(42).random(len(rays)) * 0.6).astype(np.float32)
    tmax[::3] = np.inf
    assert np.array_equal(sc.trace_any(rays, tmax), orc.any_hit(osc, rays, tmax))


def _brute_force(verts, tris, ray):
    """Moeller-Trumbore over every triangle in float64: nearest t > 1e-4."""
    p = verts["pos"][tris["idx"].astype(np.int64)].astype(np.float64)
    v0, e1, e2 = p[:, 0], p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]
    o, d = ray["o"].astype(np.float64), ray["d"].astype(np.float64)
    h = np.cross(d, e2)
    a = np.einsum("ij,ij->i", e1, h)
    ok = np.abs(a) >= 1e-7
    f = 1.0 / np.where(ok, a, 1.0)
    s = o - v0
    u = f * np.einsum("ij,ij->i", s, h)
    qq = np.cross(s, e1)
    v = f * (qq @ d)
    t = f * np.einsum("ij,ij->i", e2, qq)
    ok &= (u >= 0) & (u <= 1) & (v >= 0) & (u + v <= 1) & (t > 1e-4)
    if not ok.any():
        return -1, np.inf
    t = np.where(ok, t, np.inf)
    i = int(np.argmin(t))
    return i, float(t[i])


def test_random_triangles_1m_properties(rta, gpu):
    n = 1_000_000
    sc = scenes.random_triangles(rta, n)
    sc.build(max_per_leaf=1, strategy=0)  # BVH_MIDDLE (BVH.cpp:66-88)
    sc.upload(gpu)
    rays = scenes.random_rays(400_000, 43)
    hits = sc.trace_closest(rays)
    hit = hits["tri"] >= 0
    assert 0.05 < hit.mean() < 0.9999
    n_all = len(sc.array(rta.ARR_TRIS))  # + the two triangles of the rectangle light
    assert n_all == n + 2 and (hits["tri"][hit] < n_all).all() and (hits["t"][hit] > 1e-4).all()
    # occlusion agrees with closest hit: unbounded segment <=> some hit; segment ending at the hit is blocked (t <= tmax
    # is accepted); segment ending at half the nearest distance is free
    inf = np.full(len(rays), np.inf, np.float32)
    assert np.array_equal(sc.trace_any(rays, inf).astype(bool), hit)
    tm = np.where(hit, hits["t"], 1.0).astype(np.float32)
    assert np.array_equal(sc.trace_any(rays, tm).astype(bool)[hit], np.ones(hit.sum(), bool))
    assert not sc.trace_any(rays, (0.5 * tm).astype(np.float32)).astype(bool)[hit].any()
    # the batch order does not matter (lanes are refilled from a shared queue: a ray's result must not depend on its neighbours)
    perm = np.random.default_rng(44).permutation(len(rays))
    again = sc.trace_closest(rays[perm])
    assert np.array_equal(again.view(np.uint8).reshape(len(rays), -1), hits[perm].view(np.uint8).reshape(len(rays), -1))
    # brute force over all 1e6 triangles for a sample of rays
    verts, tris = sc.array(rta.ARR_VERTS), sc.array(rta.ARR_TRIS)
    for i in range(0, len(rays), len(rays) // 24):
        tri, t = _brute_force(verts, tris, rays[i])
        if tri < 0:
            assert hits["tri"][i] == -1
        else:
            assert abs(hits["t"][i] - t) <= 1e-5 * t, (i, hits[i], tri, t)
            assert hits["tri"][i] == tri or abs(hits["t"][i] - t) <= 1e-6 * t


def test_cornell_with_instances_bit_parity(rta, orc, gpu):
    """BASELINE config 5 in small: cornellbox + one mesh instanced 12 times (flattened at set-up): hits, occlusion and a
    small image against the oracle."""
    sc = scenes.cornell_with_instances(rta, 12, 16, 12)
    sc.build()
    sc.upload(gpu)
    osc = orc.scene_from_host(sc)
    cam = scenes.cornell_camera(rta, sc, 96, 54)
    rays = np.concatenate([orc.camera_rays(cam, 96, 54, 0, 2), scenes.random_rays(20_000, 51, lo=(-1.7, 0, -1.7), hi=(1.7, 3.4, 1.7))])
    got = sc.trace_closest(rays)
    want, _ = orc.closest_hit(osc, rays)
    assert np.array_equal(got["tri"], want["tri"])
    hit = want["tri"] >= 0
    n_cornell = 13170
    assert (want["tri"][hit] >= n_cornell).mean() > 0.005  # the instances are in view
    for f in ("t", "u", "v"):
        assert np.array_equal(got[f][hit], want[f][hit]), f
    tmax = np.full(len(rays), np.inf, np.float32)
    assert np.array_equal(sc.trace_any(rays, tmax), orc.any_hit(osc, rays, tmax))
    img, _ = sc.render(cam, 96, 54, spp=8, max_depth=8, seed=3)
    ref, _ = orc.render(osc, cam, 96, 54, 8, 8, seed=3)
    rel = float(np.sqrt(np.mean((img - ref) ** 2)) / np.sqrt(np.mean(ref ** 2)))
    assert rel < 1e-3, rel


def test_instanced_5m_properties(rta, gpu):
    """The configuration's full size: 100 352-triangle mesh x 50 instances + cornellbox = 5.03 M triangles, checked through
    occlusion <=> closest hit and batch-order independence.  (SAH tree: the BVH_MIDDLE tree of this scene is deeper than the 64
    entries of the reference's traversal stack, Object.h:275, and rt_scene_create refuses such a tree.)"""
    sc = scenes.cornell_with_instances(rta, 225, 224, 50)
    n = len(sc.array(rta.ARR_TRIS))
    assert n == 13170 + 50 * 2 * 224 * 224
    sc.build(max_per_leaf=1, strategy=0)
    with pytest.raises(rta.RtError, match="deeper than 64"):
        sc.upload(gpu)
    sc.build(max_per_leaf=2, strategy=1)
    sc.upload(gpu)
    rays = scenes.random_rays(300_000, 52, lo=(-1.7, 0, -1.7), hi=(1.7, 3.4, 1.7))
    hits = sc.trace_closest(rays)
    hit = hits["tri"] >= 0
    assert hit.mean() > 0.8 and (hits["tri"][hit] < n).all() and (hits["tri"][hit] >= 13170).mean() > 0.005
    inf = np.full(len(rays), np.inf, np.float32)
    assert np.array_equal(sc.trace_any(rays, inf).astype(bool), hit)
    tm = np.where(hit, hits["t"], 1.0).astype(np.float32)
    assert sc.trace_any(rays, tm).astype(bool)[hit].all() and not sc.trace_any(rays, (0.5 * tm).astype(np.float32)).astype(bool)[hit].any()
    perm = np.random.default_rng(53).permutation(len(rays))
    again = sc.trace_closest(rays[perm])
    assert np.array_equal(again.view(np.uint8), hits[perm].view(np.uint8))


def test_sphere_field_10m_properties(rta, gpu):
    """BASELINE config 4 geometry at full size: 10 000 spheres x 960 triangles = 9.6 M triangles (BVH_MIDDLE tree; 614 MB
    of child-pair records, beyond any cache), through occlusion <=> closest hit, batch-order independence and the radius
    of the hit points (every hit lies on some sphere: within its tessellation error of a sphere centre)."""
    sc = scenes.sphere_field(rta)
    n = len(sc.array(rta.ARR_TRIS))
    assert n == 10_000 * 960 + 2
    sc.build(max_per_leaf=1, strategy=0)
    sc.upload(gpu)
    rays = scenes.random_rays(300_000, 71, lo=(-0.1, -0.1, -0.1), hi=(1.1, 1.1, 1.1))
    hits = sc.trace_closest(rays)
    hit = hits["tri"] >= 0
    assert 0.3 < hit.mean() < 0.9999 and (hits["tri"][hit] < n).all()
    inf = np.full(len(rays), np.inf, np.float32)
    assert np.array_equal(sc.trace_any(rays, inf).astype(bool), hit)
    tm = np.where(hit, hits["t"], 1.0).astype(np.float32)
    assert sc.trace_any(rays, tm).astype(bool)[hit].all() and not sc.trace_any(rays, (0.5 * tm).astype(np.float32)).astype(bool)[hit].any()
    perm = np.random.default_rng(72).permutation(len(rays))
    again = sc.trace_closest(rays[perm])
    assert np.array_equal(again.view(np.uint8), hits[perm].view(np.uint8))
    # the hit triangle contains the hit point: |p - (v0 + u e1 + v e2)| is rounding-small
    verts, tris = sc.array(rta.ARR_VERTS), sc.array(rta.ARR_TRIS)
    idx = np.flatnonzero(hit)[:50_000]
    tv = verts["pos"][tris["idx"][hits["tri"][idx]].astype(np.int64)].astype(np.float64)
    u, v = hits["u"][idx].astype(np.float64), hits["v"][idx].astype(np.float64)
    on_tri = tv[:, 0] + u[:, None] * (tv[:, 1] - tv[:, 0]) + v[:, None] * (tv[:, 2] - tv[:, 0])
    p = rays["o"][idx].astype(np.float64) + rays["d"][idx].astype(np.float64) * hits["t"][idx][:, None]
    assert np.abs(p - on_tri).max() < 1e-4
