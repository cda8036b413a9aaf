#!/usr/bin/env python3
"""Lane occupancy of the TWO-LEVEL walks on the instanced5m workload (needs the -DRT_STATS build copied over librt_hip.so:
LS_SCRIPT=tools/lane_stats2.py bash tools/lane_stats.sh).  Prints wave-level step counts, lanes active in them, and the lanes
per kind of leaf-turn work (mesh exits, mesh leaves, TLAS leaves, finished rays)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import raytracing_amd as rta
import scenes

lib = C.CDLL(os.path.join(ROOT, "raytracing_amd", "librt_hip.so"), mode=C.RTLD_GLOBAL)
sc = scenes.legacy_instanced(rta, max_per_leaf=2)
sc.build()
sc.upload(0)
cam = scenes.legacy_instanced_camera(rta, 1920, 1080)
SPP = int(os.environ.get("LS_SPP", "4"))
a, b = (C.c_ulonglong * 16)(), (C.c_ulonglong * 16)()
sc.render(cam, 1920, 1080, spp=SPP, max_depth=8, seed=1, spp_begin=0, spp_total=1024)
lib.rt_debug_lane_stats(a)
lib.rt_debug_lane_stats2(b)
_, st = sc.render(cam, 1920, 1080, spp=SPP, max_depth=8, seed=1, spp_begin=SPP, spp_total=1024)
lib.rt_debug_lane_stats(a)
lib.rt_debug_lane_stats2(b)
v, k = list(a), list(b)
for name, base, rays in (("closest2", 0, st["rays_closest"]), ("shadow2", 8, st["rays_shadow"])):
    inner_w, inner_l, leaf_w, leaf_l, _, top_l, rounds, held = v[base:base + 8]
    print(f"{name}: rays {rays}  rounds {rounds} (lanes holding a ray {held / max(rounds, 1):.1f})")
    print(f"  inner steps: {inner_w} wave-steps, {inner_l / max(inner_w, 1):.1f} lanes each, {inner_l / rays:.2f} per ray; from the LDS top {100.0 * top_l / max(inner_l, 1):.1f} %")
    print(f"  leaf turns : {leaf_w} wave-turns, {leaf_l / max(leaf_w, 1):.1f} lanes each, {leaf_l / rays:.2f} per ray")
    for i, kind in enumerate(("mesh exits", "mesh leaves", "TLAS leaves", "finished rays")):
        w, l = k[base + 2 * i], k[base + 2 * i + 1]
        print(f"    {kind:14s}: in {w} turns, {l / max(w, 1):.1f} lanes each, {l / rays:.2f} per ray")
