"""Pins the oracle (CPU restatement) with closed-form known answers.  The reference ships no golden vectors and cannot
be built in this image, so these — plus the SURVEY's recorded observation of the reference (a triangle hit at t = 4
exactly) — are what the oracle stands on ("parity unpinned" against the compiled reference)."""
import math

import numpy as np
import pytest

import scenes


def _single_tri_scene(rta, orc, pts=((-1, -1, -4), (1, -1, -4), (0, 1, -4))):
    sc = rta.Scene()
    sc.add_material(rta.material(rta.MAT_MATTE, (0.5, 0.5, 0.5)))
    v = np.zeros(3, rta.VERTEX)
    v["pos"] = pts
    v["normal"] = (0, 0, 1)
    t = np.zeros(1, rta.TRIANGLE)
    t["idx"] = [(0, 1, 2)]
    sc.add_mesh(v, t)
    sc.build()
    return sc, orc.scene_from_host(sc)


def _ray(orc, o, d):
    r = np.zeros(1, orc.RAY)
    r["o"], r["d"] = o, d
    return r


def test_triangle_hit_known_values(rta, orc):
    _, osc = _single_tri_scene(rta, orc)
    h, vis = orc.closest_hit(osc, _ray(orc, (0, 0, 0), (0, 0, -1)))
    assert h["tri"][0] == 0 and h["t"][0] == 4.0  # "hits at t = 4 exactly"
    # Moller-Trumbore barycentrics: u along v1-v0, v along v2-v0; the point (0,0,-4) = v0 + 0.25 e1 + 0.5 e2
    assert h["u"][0] == pytest.approx(0.25) and h["v"][0] == pytest.approx(0.5)
    assert vis["leaf"][0] == 1 and vis["inner"][0] == 0
    assert orc.closest_hit(osc, _ray(orc, (0, 0, 0), (0, 0, 1)))[0]["tri"][0] == -1     # behind
    assert orc.closest_hit(osc, _ray(orc, (5, 0, 0), (0, 0, -1)))[0]["tri"][0] == -1     # outside
    assert orc.closest_hit(osc, _ray(orc, (0, 0, -4), (1, 0, 0)))[0]["tri"][0] == -1     # parallel: |a| < 1e-7
    # t <= 1e-4 is rejected (Object.h:260): a ray starting 5e-5 in front of the triangle misses, 2e-4 hits
    assert orc.closest_hit(osc, _ray(orc, (0, 0, -4 + 5e-5), (0, 0, -1)))[0]["tri"][0] == -1
    assert orc.closest_hit(osc, _ray(orc, (0, 0, -4 + 2e-4), (0, 0, -1)))[0]["tri"][0] == 0
    # backface is hit too (no culling)
    assert orc.closest_hit(osc, _ray(orc, (0, 0, -8), (0, 0, 1)))[0]["t"][0] == 4.0
    # any-hit honours tmax (main.cpp:281)
    assert orc.any_hit(osc, _ray(orc, (0, 0, 0), (0, 0, -1)), [3.9])[0] == 0
    assert orc.any_hit(osc, _ray(orc, (0, 0, 0), (0, 0, -1)), [4.0])[0] == 1


def test_closest_hit_equals_brute_force(rta, orc):
    """The BVH traversal must return the nearest of all triangle hits (ties aside)."""
    sc = scenes.random_triangles(rta, 400, seed=3)
    sc.build()
    osc = orc.scene_from_host(sc, own_bvh=True)
    rays = scenes.random_rays(300, 4)
    hits, _ = orc.closest_hit(osc, rays)
    verts, tris = sc.array(rta.ARR_VERTS), sc.array(rta.ARR_TRIS)
    p = verts["pos"][tris["idx"]].astype(np.float64)
    for k in range(len(rays)):
        o, d = rays["o"][k].astype(np.float64), rays["d"][k].astype(np.float64)
        e1, e2 = p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]
        hh = np.cross(d, e2)
        a = np.einsum("ij,ij->i", e1, hh)
        ok = np.abs(a) > 1e-7
        f = np.where(ok, 1.0 / np.where(ok, a, 1.0), 0.0)
        s = o - p[:, 0]
        u = f * np.einsum("ij,ij->i", s, hh)
        q = np.cross(s, e1)
        v = f * (q @ d)
        t = f * np.einsum("ij,ij->i", e2, q)
        ok &= (u >= 0) & (u <= 1) & (v >= 0) & (u + v <= 1) & (t > 1e-4)
        if not ok.any():
            assert hits["tri"][k] == -1
        else:
            best = np.where(ok, t, np.inf).min()
            assert hits["tri"][k] >= 0 and hits["t"][k] == pytest.approx(best, rel=1e-4)


def test_hammersley_and_rng(orc):
    import ctypes as C

    cam = orc.camera(orc.look_at((0, 0, 0), (0, 0, -1), (0, 1, 0)), 90.0, 2, 2)
    # sample 0 of 1: dx = 0 -> clamped 1e-8, dy = 0: offset = (0.5*sqrt(-2 ln 1e-8), 0) ~ (3.03, 0) px
    r = orc.camera_rays(cam, 2, 2, 0, 1, tile=(0, 0, 1, 1))[0]
    px = 0.5 + 0.5 * math.sqrt(-2 * math.log(1e-8))
    want = np.array([(2 * px / 2 - 1), (2 * 0.5 / 2 - 1), -1.0])
    want /= np.linalg.norm(want)
    assert np.allclose(r["d"], want, atol=1e-6) and np.all(r["o"] == 0)
    draws = np.array([orc.lib().orc_rng_draw(orc.lib().orc_rng_key(5, 9, 1), c) for c in range(4000)])
    assert draws.min() >= 0.0 and draws.max() < 1.0 and abs(draws.mean() - 0.5) < 0.02
    assert orc.lib().orc_rng_key(5, 9, 1) != orc.lib().orc_rng_key(5, 10, 1) != orc.lib().orc_rng_key(6, 9, 1)


def test_lambert_and_env_closed_forms(rta, orc):
    m = rta.material(rta.MAT_MATTE, (0.2, 0.4, 0.8))
    n = (0, 0, 1)
    out = np.array([0.6, 0.0, 0.8], np.float32)
    f, pdf = orc.material_eval(m, n, 1, (0, 0, -1), out)
    assert pdf == pytest.approx(0.8 / math.pi, rel=1e-6)  # Material.h:86-88
    assert np.allclose(f, np.array([0.2, 0.4, 0.8]) * 0.8 / math.pi, rtol=1e-6)
    f, pdf = orc.material_eval(m, n, 1, (0, 0, -1), (0.6, 0.0, -0.8))
    assert pdf == 0 and not f.any()
    # cosine sampling: r1 = 0, r2 = 0.36 -> (sin,0,cos) = (0.6, 0, 0.8) in the ONB of n = +z (T = normalize(up x N) ...)
    o, d, f, pdf, used = orc.material_sample(m, (0, 0, 0), n, 1, (0, 0, -1), [0.0, 0.36])
    assert used == 2 and d[2] == pytest.approx(0.8, rel=1e-6) and np.linalg.norm(d) == pytest.approx(1, rel=1e-6)
    assert np.allclose(o, (0, 0, 1e-3)) and pdf == pytest.approx(0.8 / math.pi, rel=1e-5)
    # emissive and the base class: pdf = 0 terminates the path (Material.h:56-59)
    e = rta.material(rta.MAT_EMISSIVE, (3, 2, 1))
    assert orc.material_sample(e, (0, 0, 0), n, 1, (0, 0, -1), [0.1, 0.2])[3] == 0.0
    # uniform environment: pdf(dir) = 1 / (2 pi^2 sin(theta)) (main.cpp:155-161), radiance = the constant
    sc = scenes.quad_room(rta, env=0.25)
    sc.build()
    osc = orc.scene_from_host(sc)
    rgb, pdf = np.zeros(3, np.float32), np.zeros(1, np.float32)
    d = np.array([0.0, math.cos(1.0), math.sin(1.0)], np.float32)
    orc.lib().orc_env_eval(orc.C.byref(osc.c), orc._ptr(d), orc._ptr(rgb), orc._ptr(pdf))
    assert np.allclose(rgb, 0.25, rtol=1e-5) and pdf[0] == pytest.approx(1 / (2 * math.pi ** 2 * math.sin(1.0)), rel=1e-4)


def test_white_furnace_plane_under_uniform_sky(rta, orc):
    """A Lambert floor under a uniform sky of radiance L, nothing else: the camera sees albedo * L, times the
    reference's own pole loss: an environment sample drawn from image row 0 has theta = 0, so env_pdf = 0
    (main.cpp:177-179), `(mis*f*c)/env_pdf` is 0/0 = NaN and the whole sample is zeroed by the NaN guard
    (main.cpp:984-985).  With an H-row uniform image that is a fraction 1/H of all samples.  Exercises the path loop,
    env NEE, BSDF sampling and both MIS weights (they must sum to one)."""
    sc = rta.Scene()
    sc.add_material(rta.material(rta.MAT_MATTE, (0.5, 0.25, 0.75)))
    for part in [scenes._quad(rta, [(-50, 0, 50), (50, 0, 50), (50, 0, -50), (-50, 0, -50)], (0, 1, 0), 0)]:
        sc.add_mesh(*part)
    L = 0.8
    H = 64
    sc.set_env(np.full((H, 2 * H, 3), L, np.float32))
    sc.build()
    osc = orc.scene_from_host(sc)
    cam = orc.camera(orc.look_at((0, 1, 0), (0, 0, -0.5), (0, 1, 0)), 20.0, 8, 8)
    spp = 4096
    img, c = orc.render(osc, cam, 8, 8, spp, 2, seed=3)
    got = img.mean(axis=(0, 1)) / spp
    assert np.allclose(got, np.array([0.5, 0.25, 0.75]) * L * (1 - 1 / H), rtol=0.02)
    assert c["rays_closest"] == 2 * 64 * spp  # bounce 0 hits the floor, bounce 1 escapes


def test_resolve_known_values(orc):
    s = np.zeros((2, 1, 3), np.float32)
    s[0, 0] = (4 * 0.5, 4 * 1.0, 4 * 2.0)  # 4 spp: means 0.5, 1, 2
    out = orc.resolve(s, 4)
    assert out[1, 0].tolist() == [int((0.5 ** (1 / 2.2)) * 255), 255, 255]  # row flip: raster row 0 -> bottom row
    assert out[0, 0].tolist() == [0, 0, 0]
