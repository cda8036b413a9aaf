#!/usr/bin/env python3
"""Lane occupancy of the traversal loops on the bench workload (needs a -DRT_STATS build copied over librt_hip.so:
tools/lane_stats.sh).  Prints, per walk, wave-level step counts and the mean number of lanes active in them."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import raytracing_amd as rta
import scenes

lib = C.CDLL(os.path.join(ROOT, "raytracing_amd", "librt_hip.so"), mode=C.RTLD_GLOBAL)
buf = (C.c_ulonglong * 16)()
import json

PER_LEAF = int(sys.argv[1]) if len(sys.argv) > 1 else 2  # the tree bench.py walks: BVH::build(bounds, 2, BVH_SAH)
sc = scenes.cornellbox(rta)
sc.build(max_per_leaf=PER_LEAF)
sc.upload(0)
summary = {"workload": "cornellbox", "max_per_leaf": PER_LEAF, "sample": f"1920x1080, {os.environ.get('LS_SPP', '4')} spp per call, depth 8, -DRT_STATS build of the walks"}
cam = scenes.cornell_camera(rta, sc, 1920, 1080)
SPP = int(os.environ.get("LS_SPP", "4"))
sc.render(cam, 1920, 1080, spp=SPP, max_depth=8, seed=1, spp_begin=0, spp_total=1024)
lib.rt_debug_lane_stats(buf)  # warm-up discarded
tbuf = (C.c_ulonglong * 16)()
if hasattr(lib, "rt_debug_walk_times"):
    lib.rt_debug_walk_times(tbuf)
_, st = sc.render(cam, 1920, 1080, spp=SPP, max_depth=8, seed=1, spp_begin=SPP, spp_total=1024)
assert lib.rt_debug_lane_stats(buf) == 0
v = list(buf)
for name, base, rays in (("closest", 0, st["rays_closest"]), ("occlusion", 8, st["rays_shadow"])):
    inner_w, inner_l, leaf_w, leaf_l, _, top_l, rounds, held = v[base:base + 8]
    print(f"{name}: rays {rays}  rounds {rounds} (lanes holding a ray {held / max(rounds, 1):.1f})")
    print(f"  inner steps: {inner_w} wave-steps, {inner_l / max(inner_w, 1):.1f} lanes each, {inner_l / rays:.2f} per ray")
    print(f"  inner steps read from the LDS tree top: {100.0 * top_l / max(inner_l, 1):.1f} %")
    print(f"  leaf turns : {leaf_w} wave-turns, {leaf_l / max(leaf_w, 1):.1f} lanes each, {leaf_l / rays:.2f} per ray")
    summary["closest" if base == 0 else "shadow"] = {
        "rays": int(rays), "inner_steps_per_ray": inner_l / rays, "lanes_per_inner_step": inner_l / max(inner_w, 1),
        "inner_from_lds_top_frac": top_l / max(inner_l, 1), "leaf_turns_per_ray": leaf_l / rays, "lanes_per_leaf_turn": leaf_l / max(leaf_w, 1),
        "lanes_holding_a_ray_per_round": held / max(rounds, 1)}
if hasattr(lib, "rt_debug_walk_times"):
    lib.rt_debug_walk_times(tbuf)
    t = list(tbuf)
    parts = ["refill", "record wait", "visit arithmetic + pop", "leaf turns", "round tests"]
    for name, base in (("closest", 0), ("occlusion", 8)):
        rounds = v[base + 6] or 1
        mark = t[base + 5] / rounds  # cycles of one mark (s_memtime + its wait)
        n_marks = [rounds, v[base], v[base], v[base + 2], rounds]  # marks taken per section
        net = [max(t[base + i] - mark * n_marks[i], 0.0) for i in range(5)]
        tot = sum(net) or 1.0
        # caveat: the -DRT_STATS build keeps its counters in registers (> 64 VGPRs: half the resident waves of the shipped
        # walks) and every mark is an s_memtime read whose wait joins the section it closes: shares are indicative, the
        # lane counts above are exact
        print(f"{name}: wave cycles by section (indicative: instrumented build at half the shipped occupancy): " + ", ".join(f"{n} {100.0 * net[i] / tot:.1f} %" for i, n in enumerate(parts)))
        print(f"  per round {tot / rounds:.0f} cycles; per visit: {net[1] / max(v[base], 1):.0f} until the record is there, {net[2] / max(v[base], 1):.0f} after; "
              f"per leaf turn {net[3] / max(v[base + 2], 1):.0f}; round tests {net[4] / rounds:.0f}; refill {net[0] / rounds:.0f} per round")
        summary["closest" if base == 0 else "shadow"]["wave_cycle_share"] = {n: net[i] / tot for i, n in enumerate(parts)}
        summary["closest" if base == 0 else "shadow"]["cycles_per_round"] = tot / rounds
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(summary, open(os.path.join(ROOT, "gpurun_out", "lane_stats_cornellbox.json"), "w"), indent=1)

sbuf = (C.c_ulonglong * 32)()
if hasattr(lib, "rt_debug_shade_stats"):
    lib.rt_debug_shade_stats(sbuf)
    sc.render(cam, 1920, 1080, spp=SPP, max_depth=8, seed=1, spp_begin=2 * SPP, spp_total=1024)
    lib.rt_debug_shade_stats(sbuf)
    names = ["loads", "miss/env lookup", "surface+emission", "Sample_Eval", "env NEE", "light NEE", "state+next ray", "(miss lanes waiting)"]
    tot = sum(sbuf[2 * i] for i in range(7))
    print("k_shade sections (wave cycles share, wave visits are not counted: lanes = lanes at the section's end / 1e6)")
    for i, n in enumerate(names):
        print(f"  {n:22s} {100.0 * sbuf[2 * i] / tot:5.1f} %   lanes {sbuf[2 * i + 1] / 1e6:8.2f} M")
