"""python -m raytracing_amd scene.glb -o out.bmp  — the reference's main() (main.cpp:1022-1050) with real flags
instead of the constants of main.cpp:21-34: load a glTF scene, set the environment, render on one MI355X through
Renderer::render_image, write a BMP."""
import argparse
import sys
import time

import numpy as np

import raytracing_amd as rta


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m raytracing_amd", description=__doc__)
    ap.add_argument("scene", help=".glb / .gltf file")
    ap.add_argument("-o", "--out", default="output.bmp", help="BMP to write (main.cpp:29 OUTPUT_NAME)")
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--spp", type=int, default=64, help="samples per pixel (main.cpp:22 SAMPLES_PER_PIXEL)")
    ap.add_argument("--depth", type=int, default=4, help="bounces (main.cpp:33 MAX_DEPTH)")
    ap.add_argument("--materials", choices=["uber", "matte", "ggx"], default="ggx", help="how glTF materials map to reference materials")
    ap.add_argument("--emissive-scale", type=float, default=15.0)
    ap.add_argument("--env-hdr", help="Radiance .hdr environment (main.cpp:1028)")
    ap.add_argument("--env", type=float, default=0.02, help="uniform environment radiance when no --env-hdr is given; negative: none")
    ap.add_argument("--env-rotate", type=float, default=0.0, help="degrees")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)

    if rta.device_count() < 1:
        print("no HIP device: the renderer has no CPU fallback", file=sys.stderr)
        return 2
    sc = rta.Scene()
    mode = {"uber": rta.GLTF_UBER, "matte": rta.GLTF_MATTE, "ggx": rta.GLTF_GGX}[a.materials]
    try:
        sc.load_gltf(a.scene, mode, a.emissive_scale)
    except Exception as e:  # the loader has already said why
        print(f"could not load {a.scene}: {e}", file=sys.stderr)
        return 1
    if a.env_hdr:
        if not sc.load_env_hdr(a.env_hdr, a.env_rotate):
            return 1
    elif a.env >= 0:
        sc.set_env(np.full((8, 16, 3), a.env, np.float32), a.env_rotate)
    t0 = time.time()
    sc.build()
    sc.upload(a.device)
    info = sc.info()
    t1 = time.time()
    _, _, st = sc.render_image(info["camera_view"], info["camera_fov"], a.width, a.height, a.spp, a.depth, seed=a.seed, bmp_path=a.out)
    dt = time.time() - t1
    rays = st["rays_closest"] + st["rays_shadow"]
    print(f"{a.scene}: {len(sc.array(rta.ARR_TRIS))} triangles, set-up {t1 - t0:.2f} s; {a.width}x{a.height} x {a.spp} spp, depth {a.depth}: "
          f"{rays} rays in {dt:.2f} s ({rays / dt / 1e6:.0f} Mrays/s) -> {a.out}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
