"""Pins the oracle's material functions (Material.cpp, Material.h) without the compiled reference, three ways:

  1. formula pins — every Eval is compared, at fixed direction pairs, with a float64 numpy evaluation written from the
     reference's source text line by line (a second, independent restatement: a slip in either shows as a mismatch);
  2. closed-form integrals — the solid-angle integral of each returned pdf by quadrature against its analytic value.
     Several of the reference's pdfs do NOT integrate to one, and the analytic value says by how much:
       Lambert / DisneyDiffuse / Sheen   cos/pi                          -> 1
       GGX, Beckmann at normal incidence D (n.h) / (4 v.h)               -> 1
       DisneyMetal                       D g1 / (4 V.z), with the reference's Smith term (Material.cpp:166-171: one
                                         "1 +" too many under the root)  -> g1_ref(v) (1 + 2 Lambda(v)), 0.8284 at normal incidence
       DisneyClearcoat                   D_c / (4 h.l): no (n.h) factor  -> -2 sqrt(b) atanh(sqrt(b)) / ln(1 - b), b = 1 - a_g^2 (> 1)
  3. sampler pins — Sample_Eval's direction is the reflection about the half vector the stated formula gives for the
     same random numbers; its (f, pdf) equal Eval's for that direction wherever the reference's own code paths agree;
     histograms of sampled directions follow the distribution the sampler is meant to draw from.

Beckmann has no reference arithmetic (BASELINE north star names it, the reference has none): pinned in closed form only
(normalisation of D, sampling inverse)."""
import math

import numpy as np
import pytest


N = np.array([0.2, 0.3, 0.93]) / np.linalg.norm([0.2, 0.3, 0.93])


def _norm(v):
    return v / np.linalg.norm(v)


def _onb(n):  # Utils.h:53-59
    up = np.array([1.0, 0, 0]) if abs(n[2]) > 0.999 else np.array([0, 0, 1.0])
    t = _norm(np.cross(up, n))
    return t, np.cross(n, t)


def _local(v, n):
    t, b = _onb(n)
    return np.array([v @ t, v @ b, v @ n])


def _lum(c):
    return 0.212671 * c[0] + 0.715160 * c[1] + 0.072169 * c[2]


def _tr_d(h, ax, ay):  # Material.cpp:160-164
    return 1.0 / (math.pi * ax * ay * ((h[0] / ax) ** 2 + (h[1] / ay) ** 2 + h[2] ** 2) ** 2)


def _smith(w, ax, ay):  # Material.cpp:166-171, as written
    inner = 1 + (w[0] * ax * w[0] * ax + w[1] * ay * w[1] * ay) / (w[2] * w[2])
    return 1.0 / (1.0 + (math.sqrt(1 + inner) - 1) / 2.0)


def _axay(rough, aniso):
    aspect = math.sqrt(1.0 - aniso * 0.9)
    return max(0.001, rough * rough / aspect), max(0.001, rough * rough * aspect)


def _fresnel3(ni, nt, eta):  # Material.cpp:343-349
    rs = (ni - eta * nt) / (ni + eta * nt)
    rp = (ni * eta - nt) / (ni * eta + nt)
    return (rs * rs + rp * rp) / 2


def _fresnel(ni, eta):  # Material.cpp:350-357
    nt2 = 1 - (1 - ni * ni) / (eta * eta)
    if nt2 < 0:
        return 1.0
    return _fresnel3(abs(ni), math.sqrt(nt2), eta)


def _diffuse_fresnel(ndw, f90):
    return 1 + (f90 - 1) * (1 - ndw) ** 5


# ---- float64 evaluations written from the reference text: (f[3], pdf) for in_dir (towards the surface), out_dir, normal n
def ref_matte(p, n, front, i, o):
    c = max(n @ o, 0.0)
    return p["c0"] * c / math.pi, c / math.pi


def ref_ggx(p, n, front, i, o):  # Material.cpp:4-13,31-66,87-98
    rough = p["s"][0]
    wo = -i
    h = _norm(wo + o)
    alpha = rough * rough

    def D():
        ndh = max(h @ n, 0.0)
        if ndh <= 0:
            return 0.0
        den = math.pi * ndh ** 4 * (alpha * alpha + math.tan(math.acos(min(ndh, 1.0))) ** 2) ** 2
        return alpha * alpha / den if den > 0 else 0.0

    def g1(ndv):
        k = rough * rough * 0.5
        den = ndv * (1 - k) + k
        return ndv / den if den != 0 else 0.0

    pdf = D() * max(h @ n, 0.0) / (4.0 * max(wo @ h, 0.0) + 1e-7)
    F = 0.8 + (1 - 0.8) * min(max(1 - max(wo @ h, 0.0), 0.0), 1.0) ** 5
    ndv, ndl = max(n @ wo, 0.0), max(n @ o, 0.0)
    f = ndl * (D() * F * g1(ndv) * g1(ndl)) / (4.0 * ndv * ndl + 1e-7)
    return np.full(3, f), pdf


def ref_disney_diffuse(p, n, front, i, o):  # Material.cpp:128-153
    albedo, rough, sub = p["c0"], p["s"][0], p["s"][1]
    h = _norm(-i + o)
    hdo, ndo, ndi = max(h @ o, 0.0), max(n @ o, 0.0), max(n @ -i, 0.0)
    fd90 = 0.5 + 2 * rough * hdo * hdo
    base = albedo / math.pi * _diffuse_fresnel(ndi, fd90) * _diffuse_fresnel(ndo, fd90) * ndo
    fss90 = rough * hdo * hdo
    ss = (1.25 * albedo) / math.pi * (_diffuse_fresnel(ndi, fss90) * _diffuse_fresnel(ndo, fss90) * (1 / (ndi + ndo) - 0.5) + 0.5) * ndo
    return (1 - sub) * base + sub * ss, ndo / math.pi


def ref_disney_metal(p, n, front, i, o):  # Material.cpp:201-212,234-248
    ax, ay = _axay(p["s"][0], p["s"][1])
    V, L = _local(-i, n), _local(o, n)
    H = _norm(V + L)
    F = p["c0"] + (1 - p["c0"]) * min(max(1 - max(V @ H, 0.0), 0.0), 1.0) ** 5
    D, g1, g2 = _tr_d(H, ax, ay), _smith(V, ax, ay), _smith(L, ax, ay)
    return F * D * g1 * g2 / (4 * V[2]), D * g1 / (4 * V[2])


def ref_clearcoat(p, n, front, i, o):  # Material.cpp:256-294,314-325
    gloss = p["s"][0]
    V, L = _local(-i, n), _local(o, n)
    H = _norm(V + L)
    r0 = (1.5 - 1) ** 2 / (1.5 + 1) ** 2
    ag = (1 - gloss) * 0.1 + gloss * 0.001
    F = r0 + (1 - r0) * (1 - max(H @ L, 0.0)) ** 5
    D = (ag * ag - 1) / (math.pi * math.log(ag * ag) * (1 + (ag * ag - 1) * H[2] * H[2]))
    G = _smith(V, 0.25, 0.25) * _smith(L, 0.25, 0.25)
    return np.full(3, F * D * G / (4 * V[2])), D / (4 * max(H @ L, 0.0) + 1e-8)


def _refraction(V, L, H, eta, ax, ay, base):  # Material.cpp:360-377
    D = _tr_d(H, ax, ay)
    hdi = H @ V
    F = _fresnel(hdi, 1 / eta)
    g1, g2 = _smith(V, ax, ay), _smith(L, ax, ay)
    hdo = abs(H @ L)
    sd = hdi + eta * hdo
    f = base * (1 - F) * D * g1 * g2 * eta * eta * abs(hdo * hdi) / (abs(V[2]) * sd * sd)
    return f, (1 - F) * D * g1 * abs(eta * eta * hdo / (sd * sd) * hdi / V[2])


def _reflection(V, L, H, eta, ax, ay, base):  # Material.cpp:379-392
    F = _fresnel(H @ V, eta)
    D, g1, g2 = _tr_d(H, ax, ay), _smith(V, ax, ay), _smith(L, ax, ay)
    return base * F * D * g1 * g2 / (4 * V[2]), F * D * g1 / (4 * V[2])


def ref_disney_glass(p, n, front, i, o):  # Material.cpp:432-456
    eta = 1 / p["s"][2] if front else p["s"][2]
    ax, ay = _axay(p["s"][0], p["s"][1])
    V, L = _local(-i, n), _local(o, n)
    if n @ o >= 0:
        return _reflection(V, L, _norm(V + L), eta, ax, ay, p["c0"])
    return _refraction(V, L, _norm(V + L * eta), eta, ax, ay, p["c0"])


def ref_rough_dielectric(p, n, front, i, o):  # Material.cpp:494-541 (the branch names are the reference's: `reflect` is n.out <= 0)
    eta_ie, rough = p["s"][0], p["s"][1]
    refl = (n @ o) <= 0
    eta = 1 / eta_ie if front else eta_ie
    half = _norm(-i + o) if refl else _norm(-i + o * eta)
    lin, lhalf, lout = _local(-i, n), _local(half, n), _local(o, n)
    if half[2] < 0:  # Material.cpp:512: the WORLD z axis
        half = -half
    hdi = half @ -i
    F = _fresnel(hdi, eta)
    a = rough * rough
    D, g1, g2 = _tr_d(lhalf, a, a), _smith(lin, a, a), _smith(lout, a, a)
    if refl:
        return p["c0"] * (F * D * g1 * g2) / (4 * abs(lin[2])), (F * D * g1) / (4 * abs(lin[2]))
    hdo = abs(half @ o)
    sd = hdi + eta * hdo
    f = p["c1"] * (1 - F) * D * g1 * g2 * eta * eta * abs(hdo * hdi) / (abs(lin[2]) * sd * sd)
    return f, (1 - F) * D * g1 * abs(eta * eta * hdo / (sd * sd) * hdi / lin[2])


def ref_sheen(p, n, front, i, o):  # Material.cpp:554-567
    base, tint_p = p["c0"], p["s"][0]
    half = _norm(-i + o)
    den = _lum(base)
    tint = base / (den if den >= 0 else 1)
    c = (1 - tint_p) + tint_p * tint
    return c * (1 - abs(half @ o)) ** 5 * (n @ o), max(n @ o, 0.0) / math.pi


def ref_uber(p, n, front, i, o):  # Material.cpp:571-609,678-722; s[] = Material.h:184-196 order
    base = p["c0"]
    st, metallic, sub, spec, rough, stint, aniso, _, _, clearcoat, _, ior = [float(x) for x in p["s"][:12]]
    V, L = _local(-i, n), _local(o, n)
    H = _norm(V + L)
    dw, mw, gw, cw = (1 - metallic) * (1 - st), 1 - st * (1 - metallic), (1 - metallic) * st, 0.25 * clearcoat
    if not front:
        dw = mw = cw = 0.0
    total = dw + mw + gw + cw
    ax, ay = _axay(rough, aniso)
    eta = 1 / ior if front else ior
    f, pdf = np.zeros(3), 0.0
    if dw / total > 0:
        hdo, ndo, ndi = max(H @ L, 0.0), max(L[2], 0.0), max(V[2], 0.0)
        fd90 = 0.5 + 2 * rough * hdo * hdo
        b = base / math.pi * _diffuse_fresnel(ndi, fd90) * _diffuse_fresnel(ndo, fd90) * ndo
        fss90 = rough * hdo * hdo
        ss = (1.25 * base) / math.pi * (_diffuse_fresnel(ndi, fss90) * _diffuse_fresnel(ndo, fss90) * (1 / (ndi + ndo) - 0.5) + 0.5) * ndo
        f = f + ((1 - sub) * b + sub * ss) * (1 - st) * (1 - metallic)
        pdf += max(L[2], 0.0) / math.pi * dw / total
    if mw / total > 0:
        ks = (1 - stint) + stint * base / _lum(base)
        r0 = (eta - 1) ** 2 / (eta + 1) ** 2
        c0 = spec * r0 * (1 - metallic) * ks + metallic * base
        F = c0 + (1 - c0) * (1 - max(L @ H, 0.0)) ** 5
        D, g1, g2 = _tr_d(H, ax, ay), _smith(V, ax, ay), _smith(L, ax, ay)
        f = f + F * D * g1 * g2 / (4 * V[2]) * (1 - st * (1 - metallic))
        pdf += D * g1 / (4 * V[2]) * mw / total
    return f, pdf


def _mats(rta):
    return {
        "matte": (rta.material(rta.MAT_MATTE, (0.2, 0.4, 0.8)), ref_matte),
        "ggx": (rta.material(rta.MAT_GGX, s=(0.6, 0.0)), ref_ggx),
        "disney_diffuse": (rta.material(rta.MAT_DISNEY_DIFFUSE, (0.725, 0.71, 0.68), s=(0.5, 0.4)), ref_disney_diffuse),
        "disney_metal": (rta.material(rta.MAT_DISNEY_METAL, (0.8, 0.2, 0.5), s=(0.6, 0.3)), ref_disney_metal),
        "clearcoat": (rta.material(rta.MAT_DISNEY_CLEARCOAT, s=(0.3,)), ref_clearcoat),
        "disney_glass": (rta.material(rta.MAT_DISNEY_GLASS, (1.0, 0.9, 0.8), s=(0.5, 0.2, 1.8)), ref_disney_glass),
        "rough_dielectric": (rta.material(rta.MAT_ROUGH_DIELECTRIC, (1, 0.5, 0.5), (0.9, 1, 0.8), s=(1.3, 0.6)), ref_rough_dielectric),
        "sheen": (rta.material(rta.MAT_DISNEY_SHEEN, (0.1, 0.3, 0.5), s=(0.3,)), ref_sheen),
        "uber": (rta.material(rta.MAT_DISNEY_UBER, (0.6, 0.25, 0.05), s=(0.2, 0.3, 0.5, 0.8, 0.5, 0.4, 0.3, 0, 0, 0.5, 0, 1.5)), ref_uber),
    }


def _pairs(seed, n):
    """Direction pairs around N: in_dir towards the surface from the front; out_dir anywhere on the sphere (the transmissive
    lobes are evaluated below the surface too)."""
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        i, o = _norm(rng.normal(size=3)), _norm(rng.normal(size=3))
        if -i @ N < 0.15 or abs(o @ N) < 0.1:
            continue
        out.append((i, o))
    return out


@pytest.mark.parametrize("name", ["matte", "ggx", "disney_diffuse", "disney_metal", "clearcoat", "disney_glass", "rough_dielectric",
                                  "sheen", "uber"])
def test_eval_matches_formulas_written_from_the_reference_text(rta, orc, name):
    mat, ref = _mats(rta)[name]
    p = {"c0": mat["c0"][0].astype(np.float64), "c1": mat["c1"][0].astype(np.float64), "s": mat["s"][0].astype(np.float64)}
    reflective_only = name in ("matte", "ggx", "disney_diffuse", "disney_metal", "clearcoat", "sheen", "uber")
    checked = 0
    for front in (1, 0) if name in ("disney_glass", "rough_dielectric", "uber") else (1,):
        for i, o in _pairs(11, 60):
            if reflective_only and o @ N < 0:
                continue
            f, pdf = orc.material_eval(mat, N, front, i, o)
            wf, wpdf = ref(p, N, bool(front), i, o)
            if not np.isfinite(wpdf) or not np.all(np.isfinite(wf)):
                continue
            assert pdf == pytest.approx(wpdf, rel=2e-4, abs=1e-7), (name, front, i, o)
            assert np.allclose(f, wf, rtol=2e-4, atol=1e-7), (name, front, i, o)
            checked += 1
    assert checked >= 25


def _sphere_grid(nt, nphi):
    ct = (np.arange(nt) + 0.5) / nt * 2 - 1
    ph = (np.arange(nphi) + 0.5) / nphi * 2 * math.pi
    c, p = np.meshgrid(ct, ph, indexing="ij")
    s = np.sqrt(1 - c * c)
    dirs = np.stack([s * np.cos(p), s * np.sin(p), c], -1).reshape(-1, 3)
    return dirs.astype(np.float32), (2.0 / nt) * (2 * math.pi / nphi)


def _integral_of_pdf(orc, mat, theta_in, nt=1024, nphi=1024):
    dirs, dw = _sphere_grid(nt, nphi)
    in_dir = np.array([math.sin(theta_in), 0, -math.cos(theta_in)], np.float32)
    _, pdf = orc.material_eval_batch(mat, (0, 0, 1), 1, in_dir, dirs)
    return float(pdf.astype(np.float64).sum() * dw)


def test_pdf_integrals_have_their_closed_form_values(rta, orc):
    # cosine lobes
    for mat in (rta.material(rta.MAT_MATTE, (0.2, 0.4, 0.8)), rta.material(rta.MAT_DISNEY_DIFFUSE, (0.7, 0.7, 0.7), s=(0.5, 0.5)),
                rta.material(rta.MAT_DISNEY_SHEEN, (0.1, 0.1, 0.5), s=(0.3,))):
        for th in (0.0, 1.1):
            assert _integral_of_pdf(orc, mat, th, 256, 256) == pytest.approx(1.0, abs=2e-3)
    # GGX / Beckmann: D (n.h) / (4 v.h) over all out directions = the normalisation of D over the half vectors the incident
    # direction can see; all of the upper hemisphere at normal incidence -> 1
    for kind, rough in ((rta.MAT_GGX, 0.7), (rta.MAT_GGX, 0.5), (rta.MAT_BECKMANN, 0.7), (rta.MAT_BECKMANN, 0.5)):
        assert _integral_of_pdf(orc, rta.material(kind, s=(rough, 0.0)), 0.0) == pytest.approx(1.0, abs=4e-3)
        assert _integral_of_pdf(orc, rta.material(kind, s=(rough, 0.0)), 1.0) <= 1.0 + 4e-3
    # DisneyMetal: g1_ref(v) (1 + 2 Lambda(v)) with alpha = roughness^2 along the incident azimuth (anisotropic = 0)
    for rough, th in ((0.7, 0.0), (0.7, 0.8), (0.9, 1.2)):
        a = rough * rough
        t2 = (a * math.tan(th)) ** 2
        lam = (math.sqrt(1 + t2) - 1) / 2
        g1_ref = 1 / (1 + (math.sqrt(2 + t2) - 1) / 2)
        want = g1_ref * (1 + 2 * lam)
        got = _integral_of_pdf(orc, rta.material(rta.MAT_DISNEY_METAL, (0.8, 0.8, 0.8), s=(rough, 0.0)), th)
        assert got == pytest.approx(want, rel=5e-3), (rough, th)
    assert 1 / (1 + (math.sqrt(2) - 1) / 2) == pytest.approx(0.828427, abs=1e-6)
    # DisneyClearcoat at normal incidence: the integral of D_c over the hemisphere of half vectors, no cosine
    for gloss in (0.0, 0.2):
        ag = (1 - gloss) * 0.1 + gloss * 0.001
        b = 1 - ag * ag
        want = -2 * math.sqrt(b) * math.atanh(math.sqrt(b)) / math.log(1 - b)
        got = _integral_of_pdf(orc, rta.material(rta.MAT_DISNEY_CLEARCOAT, s=(gloss,)), 0.0, 2048, 512)
        assert want > 1.0 and got == pytest.approx(want, rel=1e-2), gloss
    # the two-sided lobes are sub-normalised: what the Fresnel split leaves to each side never exceeds one in total
    for mat in (rta.material(rta.MAT_DISNEY_GLASS, (1, 1, 1), s=(0.6, 0.0, 1.5)), rta.material(rta.MAT_ROUGH_DIELECTRIC, (1, 1, 1), (1, 1, 1), s=(1.5, 0.7))):
        for th in (0.0, 0.9):
            assert 0.0 < _integral_of_pdf(orc, mat, th, 512, 512) <= 1.0 + 5e-3


def test_lambert_energy_and_furnace_of_f(rta, orc):
    """The integral of f (which carries the cosine) over the hemisphere is the albedo for Lambert."""
    dirs, dw = _sphere_grid(256, 256)
    f, _ = orc.material_eval_batch(rta.material(rta.MAT_MATTE, (0.2, 0.4, 0.8)), (0, 0, 1), 1, (0.6, 0, -0.8), dirs)
    assert np.allclose(f.astype(np.float64).sum(axis=0) * dw, (0.2, 0.4, 0.8), atol=2e-3)


def _sample(orc, mat, theta_in, n, seed):
    rng = np.random.default_rng(seed)
    rnd = rng.random((n, 4)).astype(np.float32)
    in_dir = np.array([math.sin(theta_in), 0, -math.cos(theta_in)], np.float32)
    o, d, f, pdf = orc.material_sample_batch(mat, (0, 0, 0), (0, 0, 1), 1, in_dir, rnd)
    return rnd, in_dir, o, d, f, pdf


def test_microfacet_samplers_invert_their_distributions(rta, orc):
    """GGXSample (Material.cpp:14-30): theta_m = atan(alpha sqrt(r1) / sqrt(1 - r1)), phi = 2 pi r2 — i.e. the inverse CDF of
    D cos(theta): P(theta_m <= t) = tan^2 t / (alpha^2 + tan^2 t).  Beckmann (new): tan^2 theta_m = -alpha^2 ln(1 - r1),
    P = 1 - exp(-tan^2 t / alpha^2).  The half vector is recovered from the sampled direction (reflection about it) and its
    polar angle is compared with the stated inverse for the SAME random number, then its histogram with the closed-form CDF."""
    for kind, rough in ((rta.MAT_GGX, 0.6), (rta.MAT_BECKMANN, 0.6)):
        alpha = rough * rough
        rnd, in_dir, o, d, f, pdf = _sample(orc, rta.material(kind, s=(rough, 0.0)), 0.3, 20000, 5)
        wo = -in_dir.astype(np.float64)
        h = d.astype(np.float64) + wo  # wo + wi is parallel to the half vector
        h /= np.linalg.norm(h, axis=1, keepdims=True)
        h *= np.sign(h[:, 2:3])
        tan_t = np.sqrt(1 - h[:, 2] ** 2) / h[:, 2]
        r1 = rnd[:, 0].astype(np.float64)
        want = alpha * np.sqrt(r1) / np.sqrt(1 - r1) if kind == rta.MAT_GGX else np.sqrt(-alpha * alpha * np.log(1 - r1))
        ok = (wo @ h.T > 0.05) & (want < 20)  # (a half vector facing away from the incident side reflects through -h: skip)
        assert ok.mean() > 0.8
        assert np.allclose(tan_t[ok], want[ok], rtol=2e-3, atol=2e-4)
        # azimuth: phi = 2 pi r2 in the ONB of n = +z (T = +y x ... : Utils.h:53-59 with up = +x)
        t2 = tan_t[ok] ** 2
        cdf = t2 / (alpha * alpha + t2) if kind == rta.MAT_GGX else 1 - np.exp(-t2 / (alpha * alpha))
        hist, _ = np.histogram(cdf, bins=10, range=(0, 1))
        assert np.all(np.abs(hist / hist.sum() - 0.1) < 0.02)  # the CDF of a sample of its own distribution is uniform
        assert o[0] == pytest.approx((0, 0, 1e-4), abs=1e-9)  # Material.cpp:76


def test_cosine_lobes_sample_what_they_evaluate(rta, orc):
    for mat in (rta.material(rta.MAT_MATTE, (0.2, 0.4, 0.8)), rta.material(rta.MAT_DISNEY_DIFFUSE, (0.7, 0.6, 0.5), s=(0.5, 0.3)),
                rta.material(rta.MAT_DISNEY_SHEEN, (0.1, 0.3, 0.5), s=(0.3,))):
        rnd, in_dir, o, d, f, pdf = _sample(orc, mat, 0.7, 4000, 9)
        f2, pdf2 = orc.material_eval_batch(mat, (0, 0, 1), 1, in_dir, d)
        assert np.array_equal(pdf, pdf2) and np.array_equal(f, f2)  # Sample_Eval returns Eval(...) (Material.h:78, Material.cpp:126,552)
        # random_cosine (Utils.h:70-83): cos(theta) = sqrt(1 - r2), so cos^2 is uniform
        assert np.allclose(d[:, 2].astype(np.float64) ** 2, 1 - rnd[:, 1].astype(np.float64), atol=2e-6)
        hist, _ = np.histogram(d[:, 2].astype(np.float64) ** 2, bins=8, range=(0, 1))
        assert np.all(np.abs(hist / hist.sum() - 0.125) < 0.03)
        assert np.allclose(pdf, d[:, 2] / math.pi, rtol=1e-5)


def test_vndf_lobes_sample_and_eval_agree(rta, orc):
    """DisneyMetal / DisneyClearcoat / RoughDielectric: Sample_Eval evaluates the same formulas as Eval for the direction it
    returns (Material.cpp:227,307,492), so (f, pdf) must agree for that direction — where the sampled direction stays on the
    side Eval's branch test expects (the reflection of a grazing half vector can dip below the surface)."""
    cases = [(rta.material(rta.MAT_DISNEY_METAL, (0.8, 0.2, 0.5), s=(0.5, 0.3)), 0.5, True),
             (rta.material(rta.MAT_DISNEY_CLEARCOAT, s=(0.2,)), 0.5, True),
             (rta.material(rta.MAT_ROUGH_DIELECTRIC, (1, 0.5, 0.5), (1, 1, 1), s=(1.3, 0.5)), 0.5, False)]
    for mat, th, reflective in cases:
        rnd, in_dir, o, d, f, pdf = _sample(orc, mat, th, 3000, 13)
        f2, pdf2 = orc.material_eval_batch(mat, (0, 0, 1), 1, in_dir, d)
        ok = np.isfinite(pdf) & np.isfinite(pdf2) & (pdf > 0) & (np.abs(d[:, 2]) > 1e-3)
        if reflective:
            # ClearcoatSample draws half vectors without regard to the incident side (Material.cpp:267-280): one facing away
            # reflects below the surface with pdf = D / 1e-8; Eval rebuilds the half vector as the opposite one
            ok &= d[:, 2] > 1e-3
        assert ok.mean() > 0.8
        assert np.allclose(pdf[ok], pdf2[ok], rtol=2e-3, atol=1e-6)
        assert np.allclose(f[ok], f2[ok], rtol=2e-3, atol=1e-6)


def _vndf(V, ax, ay, r1, r2):  # Material.cpp:178-199 as written (T1 is scaled by the length of Wi's, not Vh's, xy part)
    Vh = _norm(np.array([ax * V[0], ay * V[1], V[2]]))
    l2 = V[0] * V[0] + V[1] * V[1]
    T1 = np.array([-Vh[1], Vh[0], 0.0]) / math.sqrt(l2) if l2 > 0 else np.array([1.0, 0, 0])
    T2 = np.cross(T1, Vh)
    r = math.sqrt(r1)
    phi = 2.0 * math.pi * r2
    t1, t2 = r * math.cos(phi), r * math.sin(phi)
    sv = 0.5 * (1 + Vh[2])
    t2 = (1 - sv) * math.sqrt(1 - t1 * t1) + sv * t2
    Nh = t1 * T1 + t2 * T2 + math.sqrt(max(1 - t1 * t1 - t2 * t2, 0.0)) * Vh
    return _norm(np.array([ax * Nh[0], ay * Nh[1], max(0.0, Nh[2])]))


def _reflect(v, n):  # Utils.h:85-88
    return v - 2 * (v @ n) * n


def _refract(uv, n, eta):  # Utils.h:89-95
    theta = min(-uv @ n, 1.0)
    perp = eta * (uv + theta * n)
    return perp - math.sqrt(abs(1.0 - perp @ perp)) * n


def _world(L, n):
    t, b = _onb(n)
    return L[0] * t + L[1] * b + L[2] * n


def test_sampled_directions_match_the_samplers_written_from_the_reference_text(rta, orc):
    """Sample_Eval's outgoing ray for given random numbers against float64 evaluations of sampleGGXVNDF (Material.cpp:178-199),
    ClearcoatSample (267-280), the Fresnel coin of DisneyGlass (410-429) and RoughDielectric (471-490), GlassMaterial
    (Material.h:248-266) and DisneyUber's lobe pick (Material.cpp:618-664); draw order = the reference's call order."""
    rng = np.random.default_rng(23)
    n = N
    checked = {k: 0 for k in ("metal", "clearcoat", "glass_d", "rough", "glass", "uber")}
    for _ in range(400):
        i = _norm(rng.normal(size=3))
        if -i @ n < 0.2:
            continue
        rnd = rng.random(4).astype(np.float32)
        r = rnd.astype(np.float64)
        V = _local(-i, n)
        # DisneyMetal
        m = rta.material(rta.MAT_DISNEY_METAL, (0.8, 0.2, 0.5), s=(0.6, 0.4))
        ax, ay = _axay(0.6, 0.4)
        o, d, f, pdf, used = orc.material_sample(m, (1, 2, 3), n, 1, i, rnd)
        want = _world(_reflect(-V, _vndf(V, ax, ay, r[0], r[1])), n)
        assert used == 2 and np.allclose(d, want, atol=3e-5) and np.allclose(o, np.array([1, 2, 3]) + n * 1e-5, atol=1e-6)
        checked["metal"] += 1
        # DisneyClearcoat
        gloss = 0.4
        m = rta.material(rta.MAT_DISNEY_CLEARCOAT, s=(gloss,))
        ag2 = ((1 - gloss) * 0.1 + gloss * 0.001) ** 2
        ct = math.sqrt((1 - ag2 ** (1 - r[0])) / (1 - ag2))
        st = min(max(math.sqrt(max(1 - ct * ct, 0.0)), 0.0), 1.0)
        H = np.array([st * math.cos(2 * math.pi * r[1]), st * math.sin(2 * math.pi * r[1]), ct])
        o, d, f, pdf, used = orc.material_sample(m, (0, 0, 0), n, 1, i, rnd)
        assert used == 2 and np.allclose(d, _world(_reflect(-V, H), n), atol=3e-5)
        checked["clearcoat"] += 1
        # DisneyGlass: F = DielectricFresnel(h.v, 1 / eta) against the third random number
        for front in (1, 0):
            eta_ie = 1.6
            m = rta.material(rta.MAT_DISNEY_GLASS, (1, 1, 1), s=(0.5, 0.2, eta_ie))
            eta = 1 / eta_ie if front else eta_ie
            ax, ay = _axay(0.5, 0.2)
            H = _vndf(V, ax, ay, r[0], r[1])
            F = _fresnel(H @ V, 1 / eta)
            o, d, f, pdf, used = orc.material_sample(m, (0, 0, 0), n, front, i, rnd)
            if abs(F - r[2]) > 1e-4:
                L = _norm(_reflect(-V, H)) if F > r[2] else _norm(_refract(-V, H, eta))
                if np.all(np.isfinite(L)):
                    assert used == 3 and np.allclose(d, _world(L, n), atol=5e-5), (front, F, r)
                    assert np.allclose(o, n * (1e-5 if F > r[2] else -1e-5), atol=1e-9)
                    checked["glass_d"] += 1
            # RoughDielectric: the same sampler with alpha = roughness^2 both ways, F with eta (not 1 / eta), coin r <= F
            m = rta.material(rta.MAT_ROUGH_DIELECTRIC, (1, 1, 1), (1, 1, 1), s=(eta_ie, 0.5))
            Hw = _world(_vndf(V, 0.25, 0.25, r[0], r[1]), n)
            F = _fresnel(Hw @ -i, eta)
            o, d, f, pdf, used = orc.material_sample(m, (0, 0, 0), n, front, i, rnd)
            if abs(F - r[2]) > 1e-4:
                L = _norm(_reflect(i, Hw)) if r[2] <= F else _norm(_refract(i, Hw, eta))
                if np.all(np.isfinite(L)):
                    assert used == 3 and np.allclose(d, L, atol=5e-5), (front, F, r)
                    checked["rough"] += 1
            # GlassMaterial: total internal reflection or Schlick against the first random number
            m = rta.material(rta.MAT_GLASS, s=(1.5,))
            ratio = 1 / 1.5 if front else 1.5
            c = min(-i @ n, 1.0)
            sn = math.sqrt(1 - c * c)
            r0 = ((1 - ratio) / (1 + ratio)) ** 2
            schlick = r0 + (1 - r0) * (1 - c) ** 5
            o, d, f, pdf, used = orc.material_sample(m, (0, 0, 0), n, front, i, rnd)
            tir = ratio * sn > 1.0
            if tir or abs(schlick - r[0]) > 1e-4:
                refl = tir or schlick > r[0]
                L = _norm(_reflect(i, n) if refl else _refract(i, n, ratio))
                assert np.allclose(d, L, atol=3e-5) and pdf == 1.0 and np.all(f == 1.0)
                assert np.allclose(o, n * (1e-3 if refl else -1e-3), atol=1e-9)
                checked["glass"] += 1
        # DisneyUber, front face: lobe = r0 * sum; diffuse below diffuse_weight, metal below + metal_weight, else nothing
        st_, metallic, clearcoat = 0.3, 0.4, 0.5
        m = rta.material(rta.MAT_DISNEY_UBER, (0.6, 0.25, 0.05), s=(st_, metallic, 0.5, 0.8, 0.5, 0.4, 0.3, 0, 0, clearcoat, 0, 1.5))
        dw, mw, gw, cw = (1 - metallic) * (1 - st_), 1 - st_ * (1 - metallic), (1 - metallic) * st_, 0.25 * clearcoat
        total = dw + mw + gw + cw
        lobe = r[0] * total
        o, d, f, pdf, used = orc.material_sample(m, (0, 0, 0), n, 1, i, rnd)
        if abs(lobe - dw) > 1e-4 and abs(lobe - dw - mw) > 1e-4:
            if lobe < dw:
                phi, ct = 2 * math.pi * r[1], math.sqrt(1 - r[2])
                L = np.array([math.cos(phi) * math.sqrt(r[2]), math.sin(phi) * math.sqrt(r[2]), ct])
                assert used == 3 and np.allclose(d, _world(L, n), atol=3e-5)
                assert pdf == pytest.approx(ct / math.pi * dw / total, rel=1e-4)
            elif lobe < dw + mw:
                ax, ay = _axay(0.5, 0.3)
                L = _reflect(-V, _vndf(V, ax, ay, r[1], r[2]))
                assert used == 3 and np.allclose(d, _world(L, n), atol=5e-5)
            else:  # glass and clearcoat draws: unwritten branches, f = 0 and pdf = 0 end the path (Material.cpp:665-676)
                assert pdf == 0.0 and not f.any()
            checked["uber"] += 1
    assert min(checked.values()) > 100, checked
