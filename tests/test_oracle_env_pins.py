"""Pins the oracle's EnviormentLight (main.cpp:88-193) and Image::lookup (Image.h:46-64) on NON-uniform maps, without the
compiled reference: a numpy restatement written from the source text (CDF, the two binary searches, bilinear wrap-around
lookup, eval, sample_eval) must agree with the oracle value for value, and the closed forms that follow from the
formulas must hold: the pdf integrates to one over the sphere, texel (x, y) is drawn with probability lum(x, y) / total
(apart from the un-normalised cdf[0], main.cpp:116-118), eval(sample's direction) returns sample's own colour and pdf,
and env_offset turns the map about the vertical axis."""
import math

import numpy as np
import pytest

import scenes


def env_maps():
    """Named non-uniform maps (h, w, 3): a bright spot, a black row, colour gradients; 64x32 = 2048 texels is beyond the
    1024 the shading kernels stage in LDS."""
    rng = np.random.default_rng(7)
    small = (0.02 + 0.3 * rng.random((8, 16, 3))).astype(np.float32)
    small[2, 5] = (40.0, 30.0, 20.0)   # sun
    small[3, 6] = (6.0, 8.0, 10.0)
    small[5, :] = 0.0                  # a black row: a flat stretch of the CDF
    small[:, 0, 0] += 0.2              # something at the wrap-around seam
    yy, xx = np.mgrid[0:32, 0:64]
    big = np.stack([0.05 + 0.2 * (1 + np.sin(xx / 64 * 2 * np.pi)) * (yy < 16), 0.05 + 0.1 * yy / 32, 0.1 + 0.15 * (1 + np.cos(xx / 64 * 4 * np.pi))], -1)
    big = big.astype(np.float32)
    big[9, 40:43] = (25.0, 22.0, 18.0)
    big[20:22, 10] = 0.0
    return {"spot16x8": small, "sun64x32": big}


def _lum(c):
    return np.float32(0.212671) * c[..., 0] + np.float32(0.715160) * c[..., 1] + np.float32(0.072169) * c[..., 2]


def ref_cdf(rgb):  # main.cpp:97-120, fp32 running sum
    w = _lum(rgb.astype(np.float32)).reshape(-1)
    cdf = np.zeros_like(w)
    acc = np.float32(0)
    for i, x in enumerate(w):
        acc = np.float32(acc + x) if i else np.float32(x)
        cdf[i] = acc
    total = cdf[-1]
    cdf[1:] = cdf[1:] / total
    return cdf, total


def ref_search(cdf, w, h, val):  # main.cpp:121-143
    lo, hi = 0, h - 1
    while lo < hi:
        mid = (lo + hi) // 2
        if val < cdf[(mid + 1) * w - 1]:
            hi = mid
        else:
            lo = mid + 1
    y = lo
    lo, hi = 0, w - 1
    while lo < hi:
        mid = (lo + hi) // 2
        if val < cdf[y * w + mid]:
            hi = mid
        else:
            lo = mid + 1
    return lo, y


def ref_lookup(rgb, u, v):  # Image.h:46-64
    h, w, _ = rgb.shape
    u = u * w - 0.5
    v = v * h - 0.5
    ufi, vfi = int(u) % w, int(v) % h  # (int) truncates toward zero, modulo() is the positive remainder (Math.h:44-47)
    uci, vci = (ufi + 1) % w, (vfi + 1) % h
    uo = math.modf(u - ufi)[0]
    vo = math.modf(v - vfi)[0]
    g = lambda x, y: rgb[y, x].astype(np.float64)
    return g(ufi, vfi) * (1 - uo) * (1 - vo) + g(ufi, vci) * (1 - uo) * vo + g(uci, vfi) * uo * (1 - vo) + g(uci, vci) * uo * vo


def ref_eval(rgb, total, offset, d):  # main.cpp:144-164
    h, w, _ = rgb.shape
    theta = math.acos(d[1])
    phi = math.atan2(d[2], d[0])
    c = ref_lookup(rgb, (math.pi + phi) / math.pi * 0.5 + offset, theta / math.pi)
    st = math.sin(theta)
    pdf = 0.0 if st == 0 else float(_lum(c)) / total * (w * h / (2 * math.pi * math.pi * st))
    return c, pdf


def ref_sample(rgb, cdf, total, offset, rnd):  # main.cpp:165-184
    h, w, _ = rgb.shape
    x, y = ref_search(cdf, w, h, rnd)
    ux, uy = x / w, y / h
    c = ref_lookup(rgb, ux, uy)
    ux -= offset
    phi, theta = ux * 2 * math.pi, uy * math.pi
    d = np.array([-math.sin(theta) * math.cos(phi), math.cos(theta), -math.sin(theta) * math.sin(phi)])
    st = math.sin(theta)
    pdf = 0.0 if st == 0 else float(_lum(c)) / total * (w * h / (2 * math.pi * math.pi * st))
    return (x, y), d, c, pdf


def _oracle_scene(rta, orc, rgb, degrees=0.0):
    sc = scenes.quad_room(rta, env=None)
    sc.set_env(rgb, degrees)
    sc.build()
    return sc, orc.scene_from_host(sc)


@pytest.mark.parametrize("name,degrees", [("spot16x8", 0.0), ("spot16x8", 90.0), ("sun64x32", 0.0), ("sun64x32", -37.0)])
def test_env_matches_the_formulas_written_from_the_reference_text(rta, orc, name, degrees):
    rgb = env_maps()[name]
    h, w, _ = rgb.shape
    sc, osc = _oracle_scene(rta, orc, rgb, degrees)
    offset = degrees / 360.0
    cdf, total = ref_cdf(rgb)
    # the host's CDF (what the GPU gets), the oracle's and the restatement's: the same fp32 running sum
    assert np.array_equal(osc.env_cdf, cdf) and osc.c.env_total == total
    assert np.array_equal(sc.array(rta.ARR_ENV_CDF), cdf) and sc.info()["env_total"] == total
    assert osc.c.env_offset == pytest.approx(offset, abs=1e-7)
    rng = np.random.default_rng(3)
    rnd = np.concatenate([rng.random(3000), [0.0, 1.0 - 2 ** -24], cdf[5:40]]).astype(np.float32)  # incl. values equal to CDF entries
    dirs, col, pdf = orc.env_sample_batch(osc, rnd)
    for k in range(len(rnd)):
        (x, y), d, c, p = ref_sample(rgb, cdf, total, offset, rnd[k])
        assert np.allclose(dirs[k], d, atol=2e-6), (k, rnd[k])
        assert np.allclose(col[k], c, rtol=1e-5, atol=1e-7) and pdf[k] == pytest.approx(p, rel=2e-5, abs=1e-9)
    d = scenes.random_rays(2000, 5)["d"]
    col, pdf = orc.env_eval_batch(osc, d)
    for k in range(len(d)):
        c, p = ref_eval(rgb, total, offset, d[k].astype(np.float64))
        assert np.allclose(col[k], c, rtol=2e-4, atol=2e-6), k   # a direction's (u, v) in fp32 against float64: the bilinear weights move
        assert pdf[k] == pytest.approx(p, rel=3e-4, abs=1e-7)


@pytest.mark.parametrize("name", ["spot16x8", "sun64x32"])
def test_env_closed_forms(rta, orc, name):
    rgb = env_maps()[name]
    h, w, _ = rgb.shape
    sc, osc = _oracle_scene(rta, orc, rgb, 25.0)
    lum = _lum(rgb).astype(np.float64)
    # (1) the pdf integrates to one: pdf sin(theta) d(theta) d(phi) = mean of the bilinear luminance * w h / total, and
    # bilinear interpolation with wrap-around keeps the mean
    nt, nphi = 8 * h, 8 * w
    th = (np.arange(nt) + 0.5) / nt * math.pi
    ph = (np.arange(nphi) + 0.5) / nphi * 2 * math.pi - math.pi
    T, P = np.meshgrid(th, ph, indexing="ij")
    dirs = np.stack([np.sin(T) * np.cos(P), np.cos(T), np.sin(T) * np.sin(P)], -1).reshape(-1, 3).astype(np.float32)
    col, pdf = orc.env_eval_batch(osc, dirs)
    integral = float((pdf.astype(np.float64) * np.sin(T).reshape(-1)).sum() * (math.pi / nt) * (2 * math.pi / nphi))
    assert integral == pytest.approx(1.0, abs=5e-3)
    assert float(pdf.max()) > 20 * float(np.median(pdf))  # the sun is in it
    # (2) texel (x, y) is drawn with probability lum / total: sweep the random number over a fine regular grid
    n = 400_000
    rnd = ((np.arange(n) + 0.5) / n).astype(np.float32)
    dirs, col, pdf = orc.env_sample_batch(osc, rnd)
    theta = np.arccos(np.clip(dirs[:, 1].astype(np.float64), -1, 1))
    y = np.rint(theta / math.pi * h).astype(int)
    phi = np.arctan2(-dirs[:, 2].astype(np.float64), -dirs[:, 0].astype(np.float64))  # dir = (-sin cos phi, cos, -sin sin phi)
    x = np.rint(((phi / (2 * math.pi) + 25.0 / 360.0) % 1.0) * w).astype(int) % w
    ok = y > 0  # row 0 is the pole: every direction is +y there
    counts = np.bincount((y * w + x)[ok], minlength=w * h).reshape(h, w)
    want = lum / lum.sum() * n
    assert np.abs(counts[1:] - want[1:]).max() <= 3.0 + 2e-4 * n / (w * h)  # (fp32 CDF rounding: a few grid points per boundary)
    assert counts[lum == 0].sum() == 0  # black texels are never drawn
    # (3) eval(sample's direction) gives sample's own colour and pdf (the two mappings are inverses, offset included)
    pick = np.flatnonzero(ok)[:: max(1, ok.sum() // 3000)]
    col2, pdf2 = orc.env_eval_batch(osc, dirs[pick])
    close = np.isclose(pdf2, pdf[pick], rtol=5e-3, atol=1e-6)
    assert close.mean() > 0.97  # (texel corners sit on bilinear kinks next to the sun: a last-bit u moves a few)
    # (4) offset turns the map about the vertical axis: eval at offset o, direction turned by -o = eval at offset 0
    _, osc0 = _oracle_scene(rta, orc, rgb, 0.0)
    d0 = scenes.random_rays(500, 9)["d"].astype(np.float64)
    a = 2 * math.pi * 25.0 / 360.0
    # u = (pi + atan2(z, x)) / (2 pi) + offset: adding the offset = turning the direction by +a about y in the (x, z) plane
    d1 = np.stack([d0[:, 0] * math.cos(a) - d0[:, 2] * math.sin(a), d0[:, 1], d0[:, 0] * math.sin(a) + d0[:, 2] * math.cos(a)], -1)
    c_off, p_off = orc.env_eval_batch(osc, d0.astype(np.float32))
    c_rot, p_rot = orc.env_eval_batch(osc0, d1.astype(np.float32))
    # ... except within half a texel of the image's left edge: there u w - 0.5 < 0, (int) truncates towards zero and the
    # "bilinear" weights become (1.3, -0.3)-style extrapolation (Image.h:49-55); the offset moves that strip elsewhere
    u0 = (math.pi + np.arctan2(d1[:, 2], d1[:, 0])) / math.pi * 0.5
    inside = u0 * w > 0.5 + 1e-3
    assert 0.9 < inside.mean() < 1.0
    assert np.isclose(p_off, p_rot, rtol=5e-3, atol=1e-5)[inside].mean() > 0.995
