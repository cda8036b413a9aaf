"""-m gpu: the occlusion walk over COMPRESSED FOUR-WIDE records (RT_TUNE_QUADS=1 at rt_scene_create; rt_device.h, "four-wide
occlusion walk") against the oracle's reference-faithful any-hit walk, flag for flag, and against the default pair walk,
pixel for pixel.  The quad walk reaches a superset of the reference's leaves through conservative 8-bit boxes and decides
each leaf by its exact box, so every flag must be the reference's — also for axis-parallel rays (they walk the pair records
inside the same kernel), rays that start in a box plane, segments that end just in front of a triangle."""
import os

import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu


@pytest.fixture()
def quads_on():
    os.environ["RT_TUNE_QUADS"] = "1"
    yield
    del os.environ["RT_TUNE_QUADS"]


def _upload_with_quads(sc, gpu):
    sc.upload(gpu)
    assert sc.quad_records() > 0, "RT_TUNE_QUADS=1 did not build the quad records"
    return sc


def _segments(n, seed, lo, hi, tmax_hi):
    rays = scenes.random_rays(n, seed, lo=lo, hi=hi)
    rng = np.random.default_rng(seed + 1)
    tmax = (rng.random(n) * tmax_hi).astype(np.float32)
    tmax[::7] = np.inf
    tmax[3::11] = 0.0
    tmax[5::13] = -1.0
    return rays, tmax


@pytest.mark.parametrize("per_leaf", [1, 2, 4])
def test_quad_walk_flags_cornellbox(rta, orc, gpu, quads_on, per_leaf):
    sc = scenes.cornellbox(rta)
    sc.build(max_per_leaf=per_leaf)
    _upload_with_quads(sc, gpu)
    osc = orc.scene_from_host(sc)
    rays, tmax = _segments(400_000, 70 + per_leaf, (-1.7, 0.0, -1.7), (1.7, 3.4, 1.7), 3.0)
    rays["d"][::17] = (0, -1, 0)  # select-form rays: these walk the pair records in the same kernel
    rays["d"][1::19] = (1, 0, 0)
    got, want = sc.trace_any(rays, tmax), orc.any_hit(osc, rays, tmax)
    assert np.array_equal(got, want), np.count_nonzero(got != want)
    assert 0.1 < want.mean() < 0.9
    # segments that end AT a surface: tmax = the closest hit's own t, one ulp less, one ulp more
    hits, _ = orc.closest_hit(osc, rays)
    ok = hits["tri"] >= 0
    for t in (hits["t"][ok], np.nextafter(hits["t"][ok], np.float32(0)), np.nextafter(hits["t"][ok], np.float32(np.inf))):
        got, want = sc.trace_any(rays[ok], t), orc.any_hit(osc, rays[ok], t)
        assert np.array_equal(got, want), np.count_nonzero(got != want)


def test_quad_walk_flags_random_triangles_and_planes(rta, orc, gpu, quads_on):
    sc = scenes.random_triangles(rta, 100_000)
    sc.build()
    _upload_with_quads(sc, gpu)
    osc = orc.scene_from_host(sc)
    rays, tmax = _segments(300_000, 81, (-0.1, -0.1, -0.1), (1.1, 1.1, 1.1), 0.5)
    got, want = sc.trace_any(rays, tmax), orc.any_hit(osc, rays, tmax)
    assert np.array_equal(got, want), np.count_nonzero(got != want)
    # a room of axis-aligned quads: flat boxes (grid steps of zero extent), origins lying IN box planes
    room = scenes.quad_room(rta)
    room.build(max_per_leaf=4, strategy=0)
    _upload_with_quads(room, gpu)
    oroom = orc.scene_from_host(room)
    rays, tmax = _segments(100_000, 83, (-1, 0, -1), (1, 2, 1), 3.0)
    rays["o"][2::5, 1] = 0.0
    rays["o"][3::5, 0] = 1.0
    rays["d"][::5] = (0, -1, 0)
    got, want = room.trace_any(rays, tmax), orc.any_hit(oroom, rays, tmax)
    assert np.array_equal(got, want), np.count_nonzero(got != want)


def test_quad_walk_renders_the_same_frame(rta, gpu):
    """Shadow rays through the quad walk: the frame of the default (pair) walk, bit for bit."""
    w, h = 160, 90
    frames = []
    for quads in (False, True):
        if quads:
            os.environ["RT_TUNE_QUADS"] = "1"
        try:
            sc = scenes.cornellbox(rta)
            sc.build(max_per_leaf=2)
            sc.upload(gpu)
        finally:
            os.environ.pop("RT_TUNE_QUADS", None)
        assert (sc.quad_records() > 0) == quads
        cam = scenes.cornell_camera(rta, sc, w, h)
        img, st = sc.render(cam, w, h, spp=8, max_depth=8, seed=3)
        frames.append((img, st["rays_shadow"]))
    assert frames[0][1] == frames[1][1]
    assert np.array_equal(frames[0][0], frames[1][0])


def test_quad_walk_equals_the_pair_walk_at_scale(rta, gpu):
    """BASELINE configs[2] geometry (1 000 000 random triangles: the tree is far beyond the LDS copy and the L2): two million
    segments, the quad walk's flags against the pair walk's (which the other suites hold against the oracle)."""
    scs = []
    for quads in (False, True):
        if quads:
            os.environ["RT_TUNE_QUADS"] = "1"
        try:
            sc = scenes.random_triangles(rta, 1_000_000)
            sc.build()
            sc.upload(gpu)
        finally:
            os.environ.pop("RT_TUNE_QUADS", None)
        assert (sc.quad_records() > 0) == quads
        scs.append(sc)
    assert 300_000 < scs[1].quad_records() < 500_000  # ~ a third of the 999 999 pair records
    rays, tmax = _segments(2_000_000, 91, (-0.1, -0.1, -0.1), (1.1, 1.1, 1.1), 0.3)
    a, b = scs[0].trace_any(rays, tmax), scs[1].trace_any(rays, tmax)
    assert np.array_equal(a, b), np.count_nonzero(a != b)
    assert 0.01 < a.mean() < 0.99, a.mean()  # both outcomes are well represented
