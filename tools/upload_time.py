import sys, time, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import raytracing_amd as rta, scenes
for name in ("spheres10m", "instanced5m"):
    t0 = time.perf_counter()
    sc = scenes.sphere_field(rta) if name == "spheres10m" else scenes.legacy_instanced(rta, max_per_leaf=2)
    t1 = time.perf_counter()
    sc.build(max_per_leaf=2) if name == "spheres10m" else sc.build()
    t2 = time.perf_counter()
    sc.upload(0)
    t3 = time.perf_counter()
    print(f"{name}: make {t1-t0:.2f} s, build {t2-t1:.2f} s, upload (flatten + rt_scene_create) {t3-t2:.2f} s", flush=True)
    sc.close()
