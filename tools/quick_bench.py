#!/usr/bin/env python3
"""Kernel-tuning loop: cornellbox at 1080p, 8 bounces, 32-spp steps.  Prints whole-step throughput with the default four
passes in flight, the stand-alone per-kernel times of a serial step (RT_RENDER_SERIAL + RT_RENDER_TIMING), and a parity
spot check of the traversal kernels against the oracle (200 k rays, bit for bit) so that a fast wrong kernel is seen at once.
usage: tools/quick_bench.py [steps] [workload: cornellbox|spheres10m|instanced5m]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_bindings as orc  # noqa: E402
import raytracing_amd as rta  # noqa: E402
import scenes  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
workload = sys.argv[2] if len(sys.argv) > 2 else "cornellbox"
W, H, DEPTH, SPP = 1920, 1080, 8, int(os.environ.get("QB_SPP", "32"))
two_level = False
if workload == "cornellbox":
    sc = scenes.cornellbox(rta)
    sc.build(max_per_leaf=2)
    cam = scenes.cornell_camera(rta, sc, W, H)
    lo, hi = (-1.7, 0, -1.7), (1.7, 3.4, 1.7)
elif workload == "instanced5m":
    sc = scenes.legacy_instanced(rta, max_per_leaf=2)
    sc.build()
    cam = scenes.legacy_instanced_camera(rta, W, H)
    lo, hi = (-1.7, 0, -1.7), (1.7, 3.4, 1.7)
    two_level = True
else:
    W, H = 3840, 2160
    SPP = 8
    sc = scenes.sphere_field(rta)
    sc.set_env(np.full((8, 16, 3), 0.3, np.float32))
    sc.build(max_per_leaf=2)
    cam = scenes.random_triangles_camera(rta, W, H)
    lo, hi = (-0.1, -0.1, -0.1), (1.1, 1.1, 1.1)
sc.upload(0)

# parity spot check
rays = scenes.random_rays(200_000, 5, lo=lo, hi=hi)
tmax = (np.random.default_rng(6).random(len(rays)) * 3).astype(np.float32)
if two_level:
    osc = orc.two_level_scene_from_host(sc)
    want, windex, _, _ = orc.closest_hit2(osc, rays)
    got, index = sc.trace_closest_indexed(rays)
    ok = np.array_equal(got["tri"], want["tri"]) and np.array_equal(index, windex) and np.array_equal(got["t"][want["tri"] >= 0], want["t"][want["tri"] >= 0])
    ok_any = np.array_equal(sc.trace_any(rays, tmax), orc.any_hit2(osc, rays, tmax))
else:
    osc = orc.scene_from_host(sc)
    want, _ = orc.closest_hit(osc, rays)
    got = sc.trace_closest(rays)
    ok = got.tobytes() == want.tobytes()
    ok_any = np.array_equal(sc.trace_any(rays, tmax), orc.any_hit(osc, rays, tmax))
print(f"parity: closest {'OK' if ok else 'MISMATCH'}, any {'OK' if ok_any else 'MISMATCH'}", flush=True)

import torch  # noqa: E402

frame = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()


def step(i, **kw):
    _, st = sc.render(cam, W, H, spp=SPP, max_depth=DEPTH, seed=1, spp_begin=i * SPP, spp_total=1024, device_out=frame.data_ptr(), **kw)
    return st


for i in range(2):
    step(i)
t0 = time.perf_counter()
rays_n = 0
for i in range(steps):
    st = step(i)
    rays_n += st["rays_closest"] + st["rays_shadow"]
dt = time.perf_counter() - t0
print(f"in flight : {rays_n / dt / 1e9:.3f} Grays/s, {dt / steps * 1e3:.2f} ms per {SPP}-spp step, frame mean {float(frame.mean()) / ((steps + 2) * SPP):.6f}", flush=True)
tot = None
for i in range(3):
    st = step(i, timing=True, serial=True)
    if i:
        tot = st if tot is None else {k: tot[k] + st[k] for k in st}
n = 2.0
print(f"stand-alone per step: closest {tot['ms_closest'] / n:.2f} ms, shadow {tot['ms_shadow'] / n:.2f} ms, shade {tot['ms_shade'] / n:.2f} ms, "
      f"generate {tot['ms_generate'] / n:.2f}, accumulate {tot['ms_accumulate'] / n:.2f}, total {tot['ms_total'] / n:.2f} ms "
      f"({(tot['rays_closest'] + tot['rays_shadow']) / tot['ms_total'] / 1e6:.3f} Grays/s serial)", flush=True)
