#!/usr/bin/env python3
"""Writes tests/golden/cornell_vectors.npz, material_vectors.npz and two_level_vectors.npz from the CPU oracle (oracle/, test
infrastructure).  The reference ships no tests, fixtures or golden vectors and cannot be built in this image (DESIGN.md
§2), so these vectors pin the ORACLE AS IT IS TODAY: a later change to the oracle, the host BVH builder, the flattener
or the HIP kernels that alters any hit index, t, u, v, occlusion flag, BSDF value or image shows up as a diff against
them.  Run from the repository root after a deliberate change:  python3 tools/make_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_bindings as orc
import raytracing_amd as rta
import scenes

OUT = os.path.join(ROOT, "tests", "golden")


def secondary(rays, hits, seed):
    ok = hits["tri"] >= 0
    p = rays["o"][ok] + rays["d"][ok] * hits["t"][ok, None]
    rng = np.random.default_rng(seed)
    d = rng.normal(size=p.shape)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    out = np.zeros(len(p), orc.RAY)
    out["o"] = (p + 1e-3 * d).astype(np.float32)
    out["d"] = d.astype(np.float32)
    return out


def cornell():
    sc = scenes.cornellbox(rta)
    sc.build()
    osc = orc.scene_from_host(sc)
    cam = scenes.cornell_camera(rta, sc, 64, 64)
    prim = orc.camera_rays(cam, 64, 64, 1, 4)                     # 4096 primary rays
    h1, _ = orc.closest_hit(osc, prim)
    sec = secondary(prim, h1, 21)[:4096]                          # incoherent bounce rays
    h2, _ = orc.closest_hit(osc, sec)
    rnd = scenes.random_rays(4096, 22, lo=(-1.7, 0, -1.7), hi=(1.7, 3.4, 1.7))
    h3, _ = orc.closest_hit(osc, rnd)
    tmax = (np.random.default_rng(23).random(len(rnd)) * 3.0).astype(np.float32)
    tmax[::5] = np.inf
    occ = orc.any_hit(osc, rnd, tmax)
    img, stats = orc.render(osc, scenes.cornell_camera(rta, sc, 48, 27), 48, 27, 8, 8, seed=5)
    np.savez_compressed(os.path.join(OUT, "cornell_vectors.npz"), primary=prim, primary_hits=h1, secondary=sec, secondary_hits=h2,
                        random=rnd, random_hits=h3, random_tmax=tmax, random_occluded=occ, image_48x27_8spp_d8_seed5=img,
                        image_rays=np.array([stats["rays_closest"], stats["rays_shadow"], stats["paths"]], np.int64))


def materials():
    rng = np.random.default_rng(31)
    n = 256
    out = {}
    table = {
        "matte": rta.material(rta.MAT_MATTE, (0.7, 0.6, 0.5)),
        "ggx": rta.material(rta.MAT_GGX, s=(0.5, 0.0)),
        "beckmann": rta.material(rta.MAT_BECKMANN, s=(0.5, 0.0)),
        "disney_diffuse": rta.material(rta.MAT_DISNEY_DIFFUSE, (0.6, 0.5, 0.4), s=(0.4, 0.3)),
        "disney_metal": rta.material(rta.MAT_DISNEY_METAL, (0.9, 0.7, 0.3), s=(0.3, 0.4)),
        "disney_clearcoat": rta.material(rta.MAT_DISNEY_CLEARCOAT, s=(0.6,)),
        "disney_glass": rta.material(rta.MAT_DISNEY_GLASS, (0.9, 0.9, 0.95), s=(0.2, 0.0, 1.5)),
        "rough_dielectric": rta.material(rta.MAT_ROUGH_DIELECTRIC, (1, 1, 1), (0.8, 0.85, 1.0), s=(1.5, 0.3)),
        "disney_sheen": rta.material(rta.MAT_DISNEY_SHEEN, (0.5, 0.2, 0.2), s=(0.5,)),
        "glass": rta.material(rta.MAT_GLASS, s=(1.5,)),
    }
    normal = np.array((0.0, 1.0, 0.0), np.float32)
    for name, m in table.items():
        ev = np.zeros((n, 10), np.float32)  # in_dir(3) out_dir(3) f(3) pdf
        sm = np.zeros((n, 13), np.float32)  # in_dir(3) rnd(3) dir(3) f(3) pdf
        for i in range(n):
            a = rng.normal(size=3)
            a[1] = -abs(a[1]) - 0.05                      # arriving from above the surface
            a /= np.linalg.norm(a)
            b = rng.normal(size=3)
            b[1] = abs(b[1]) + 0.05
            b /= np.linalg.norm(b)
            f, pdf = orc.material_eval(m, normal, 1, a.astype(np.float32), b.astype(np.float32))
            ev[i] = (*a, *b, *f, pdf)
            r3 = rng.random(3).astype(np.float32)
            _, sd, sf, spdf, _ = orc.material_sample(m, np.zeros(3, np.float32), normal, 1, a.astype(np.float32), r3)
            sm[i] = (*a, *r3, *sd, *sf, spdf)
        out[name + "_eval"] = ev
        out[name + "_sample"] = sm
    np.savez_compressed(os.path.join(OUT, "material_vectors.npz"), **out)


def two_level():
    """The legacy two-level design: scenes.legacy_two_level (six geometry kinds, one mesh placed three times, a light)."""
    sc = scenes.legacy_two_level(rta, extra_spheres=20)
    sc.build()
    osc = orc.two_level_scene_from_host(sc)
    cam = scenes.legacy_camera(rta, 64, 64)
    prim = orc.camera_rays(cam, 64, 64, 1, 4)
    h1, i1, n1, f1 = orc.closest_hit2(osc, prim)
    sec = secondary(prim, h1, 41)[:4096]
    h2, i2, n2, f2 = orc.closest_hit2(osc, sec)
    tmax = (np.random.default_rng(43).random(len(sec)) * 1.5).astype(np.float32)
    tmax[::5] = np.inf
    occ = orc.any_hit2(osc, sec, tmax)
    img, stats = orc.render(osc, scenes.legacy_camera(rta, 48, 32), 48, 32, 8, 5, seed=6)
    np.savez_compressed(os.path.join(OUT, "two_level_vectors.npz"), primary=prim, primary_hits=h1, primary_index=i1, primary_normal=n1,
                        primary_front=f1, secondary=sec, secondary_hits=h2, secondary_index=i2, secondary_normal=n2, secondary_front=f2,
                        secondary_tmax=tmax, secondary_occluded=occ, image_48x32_8spp_d5_seed6=img,
                        image_rays=np.array([stats["rays_closest"], stats["rays_shadow"], stats["paths"]], np.int64))


if __name__ == "__main__":
    if "--only-two-level" not in sys.argv:
        cornell()
        materials()
    two_level()
    for f in ("cornell_vectors.npz", "material_vectors.npz", "two_level_vectors.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)), "bytes")
