#!/usr/bin/env python3
"""Measures <inner-node visits> and <leaf visits> per ray on the bench workload by instrumenting the CPU ORACLE's
traversal, and writes the algorithmic bytes per ray that bench.py's roofline uses:
    B_ray = 64 * <inner> + 52 * <leaf> + 48
(64 B child-pair record, 52 B = 4 B index + 48 B triangle, 48 B = 32 B ray read + 16 B hit write).
The full 1080p view is sampled on a 480x270 raster (same camera, same materials, depth 8): the visit statistics are a
property of the ray distribution, not of the resolution.  Test/measurement infrastructure: uses oracle/."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_bindings as orc  # noqa: E402
import raytracing_amd as rta  # noqa: E402
import scenes  # noqa: E402

W, H, SPP, DEPTH = 480, 270, 8, 8
sc = scenes.cornellbox(rta)
sc.build()
osc = orc.scene_from_host(sc)
cam = scenes.cornell_camera(rta, sc, W, H)
_, c = orc.render(osc, cam, W, H, SPP, DEPTH, seed=1, spp_total=1024)


def entry(inner, leaf, rays):
    i, l = inner / rays, leaf / rays
    return {"inner_per_ray": i, "leaf_per_ray": l, "bytes_per_ray": 64 * i + 52 * l + 48, "rays": rays}


out = {
    "workload": f"cornellbox.glb GGX+Beckmann, depth {DEPTH}, camera of the file, sampled at {W}x{H} x {SPP} spp by the oracle",
    "formula": "bytes_per_ray = 64*inner + 52*leaf + 48",
    "closest": entry(c["inner_visits"], c["leaf_visits"], c["rays_closest"]),
    "shadow": entry(c["shadow_inner_visits"], c["shadow_leaf_visits"], c["rays_shadow"]),
    "rays_per_sample": (c["rays_closest"] + c["rays_shadow"]) / c["paths"],
}
path = os.path.join(ROOT, "profiles", "visits_cornellbox_1080p_d8.json")
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
