#!/usr/bin/env python3
"""Measures <inner-node visits> and <leaf visits> per ray on the bench workload by instrumenting the CPU ORACLE's
traversal, and writes the algorithmic bytes per ray that bench.py's roofline uses:
    B_ray = 64 * <inner> + 52 * <leaf> + 48
(64 B child-pair record, 52 B = 4 B index + 48 B triangle per triangle tested — <leaf> counts triangles, which is leaves
when a leaf holds one —, 48 B = 32 B ray read + 16 B hit write).
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

WORKLOAD = sys.argv[1] if len(sys.argv) > 1 else "cornellbox"
per_leaf = 2 if WORKLOAD.endswith("_leaf2") else 1  # <workload>_leaf2: BVH::build(bounds, 2, BVH_SAH) — the tree bench.py walks by default
if per_leaf == 2:
    WORKLOAD = WORKLOAD[: -len("_leaf2")]
if WORKLOAD == "cornellbox":
    W, H, SPP, DEPTH = 480, 270, 8, 8
    sc = scenes.cornellbox(rta)
    sc.build(max_per_leaf=per_leaf)
    cam = scenes.cornell_camera(rta, sc, W, H)
    label = f"cornellbox.glb GGX+Beckmann, depth {DEPTH}, camera of the file, BVH::build(bounds, {per_leaf}, BVH_SAH)"
    name = "visits_cornellbox_1080p_d8.json"
elif WORKLOAD == "spheres10m":  # BASELINE config 4 geometry: 9.6 M triangles (the SAH build takes about a minute on 8 cores)
    W, H, SPP, DEPTH = 480, 270, 1, 8
    import numpy as np
    sc = scenes.sphere_field(rta)
    sc.build(max_per_leaf=per_leaf)
    sc.set_env(np.full((8, 16, 3), 0.3, np.float32))
    cam = scenes.random_triangles_camera(rta, W, H)
    label = "10 000 tessellated spheres (9.6 M triangles, seed 1234), rectangle light + uniform environment, depth 8"
    name = "visits_spheres10m_4k_d8.json"
elif WORKLOAD == "instanced5m":  # BASELINE config 5 as a two-level scene (an instance test counts as 3 inner visits: its 192-byte record)
    W, H, SPP, DEPTH = 480, 270, 2, 8
    sc = scenes.legacy_instanced(rta, max_per_leaf=per_leaf)  # TriangleMesh(mesh, per_leaf)
    sc.build()
    cam = scenes.legacy_instanced_camera(rta, W, H)
    label = "room of Rectangle instances + one 99 856-triangle mesh x 50 Instance transforms (two-level), depth 8"
    name = "visits_instanced5m_1080p_d8.json"
elif WORKLOAD == "instanced5m_flat":  # the same surfaces flattened into one 5 M-triangle mesh (the SAH build takes about a minute on 8 cores)
    W, H, SPP, DEPTH = 480, 270, 2, 8
    sc = scenes.legacy_instanced_flat(rta)
    sc.build(max_per_leaf=per_leaf)
    cam = scenes.legacy_instanced_camera(rta, W, H)
    label = "room + one 99 856-triangle mesh x 50 flattened into one mesh, depth 8"
    name = "visits_instanced5m_flat_1080p_d8.json"
else:  # BASELINE config 3: 1 000 000 random triangles, primary + shadow rays only (depth 1); the SAH build takes ~40 s
    W, H, SPP, DEPTH = 480, 270, 2, 1
    sc = scenes.random_triangles(rta, 1_000_000)
    sc.build(max_per_leaf=per_leaf)
    cam = scenes.random_triangles_camera(rta, W, H)
    label = "1 000 000 random triangles in the unit box (seed 1234), one rectangle light, depth 1"
    name = "visits_random1m_1080p_d1.json"
if per_leaf == 2:
    name = name[:-5] + "_leaf2.json"
    if WORKLOAD != "cornellbox":
        label += ", BVH::build(bounds, 2, BVH_SAH)"
osc = orc.two_level_scene_from_host(sc) if WORKLOAD == "instanced5m" else orc.scene_from_host(sc)
orc.lib().orc_count_product_occlusion_walk(1)  # shadow-ray visits of the walk the kernels run (boxes beyond tmax skipped)
_, c = orc.render(osc, cam, W, H, SPP, DEPTH, seed=1, spp_total=1024)
orc.lib().orc_count_product_occlusion_walk(0)


def entry(inner, leaf, rays):
    i, l = inner / rays, leaf / rays
    return {"inner_per_ray": i, "leaf_per_ray": l, "bytes_per_ray": 64 * i + 52 * l + 48, "rays": rays}


out = {
    "workload": f"{label}, sampled at {W}x{H} x {SPP} spp by the oracle",
    "formula": "bytes_per_ray = 64*inner + 52*leaf + 48",
    "closest": entry(c["inner_visits"], c["leaf_visits"], c["rays_closest"]),
    "shadow": entry(c["shadow_inner_visits"], c["shadow_leaf_visits"], c["rays_shadow"]),
    "rays_per_sample": (c["rays_closest"] + c["rays_shadow"]) / c["paths"],
}
path = os.path.join(ROOT, "profiles", name)
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
