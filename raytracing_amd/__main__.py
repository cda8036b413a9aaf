"""python -m raytracing_amd scene.glb -o out.bmp  — the reference's main() (main.cpp:1022-1050) with real flags
instead of the constants of main.cpp:21-34: load a glTF scene, set the environment, render on one MI355X through
Renderer::render_image, write a BMP."""
import argparse
import sys
import time

import numpy as np

import raytracing_amd as rta


def _translate(x, y, z):
    m = np.eye(4, dtype=np.float32)
    m[:3, 3] = (x, y, z)
    return m.T.reshape(-1)  # column-major


def ball_test_scene(sc):
    """The instances, materials and camera of the reference's BallTestScene (main.cpp:725-781): a 0.7 x 3 Rectangle and
    five Spheres of radius 0.1 with DisneyUber / DisneyGlass / DisneyDiffuse / DisneyMetal materials, lit by the
    environment.  Returns (view matrix, vertical fov)."""
    uber = lambda base, rough, metallic: rta.material(  # DisneyUber defaults (Material.h:183-196: zeros, ior 1) with the script's overrides
        rta.MAT_DISNEY_UBER, base, s=(0.0, metallic, 0.0, 1.0, rough, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0))
    floor = sc.add_material(rta.material(rta.MAT_DISNEY_DIFFUSE, (0.7, 0.7, 0.7), s=(0.4, 0.5)))
    sc.add_primitive_instance(rta.GEO_RECTANGLE, (0, 1, 0, 0.7, 3.0), floor, _translate(0, 0, 0))
    balls = [
        ((0, 0.1, 0.0), uber((0.6, 0.05, 0.05), 0.1, 0.0)),
        ((0, 0.1, 0.25), rta.material(rta.MAT_DISNEY_GLASS, (1, 1, 1), s=(0.01, 0.0, 1.5))),
        ((0, 0.1, -0.25), uber((0.7, 0.7, 0.7), 0.6, 0.8)),
        ((0, 0.1, -0.5), rta.material(rta.MAT_DISNEY_DIFFUSE, (0.7, 0.7, 0.7), s=(0.5, 0.5))),
        ((0, 0.1, 0.5), rta.material(rta.MAT_DISNEY_METAL, (0.7, 0.7, 0.7), s=(0.1, 0.0))),
    ]
    for pos, mat in balls:
        sc.add_primitive_instance(rta.GEO_SPHERE, (0.1,), sc.add_material(mat), _translate(*pos))
    return rta.look_at((-1.4, 0.7, 0.0), (0.0, 0.2, 0.0), (0, 1, 0)), 45.0


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m raytracing_amd", description=__doc__)
    ap.add_argument("scene", help=".glb / .gltf file, or `balltest`: the reference's BallTestScene (main.cpp:725-781), a legacy two-level "
                                  "scene of Rectangle and Sphere instances under a TLAS")
    ap.add_argument("-o", "--out", default="output.bmp", help="image to write (main.cpp:29 OUTPUT_NAME): .png by extension, else BMP")
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
    view = fov = None
    if a.scene == "balltest":
        view, fov = ball_test_scene(sc)
    else:
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
    if view is None:
        view, fov = info["camera_view"], info["camera_fov"]
    _, _, st = sc.render_image(view, fov, a.width, a.height, a.spp, a.depth, seed=a.seed, bmp_path=a.out)
    dt = time.time() - t1
    rays = st["rays_closest"] + st["rays_shadow"]
    n_inst = len(sc.array(rta.ARR_INSTANCES))
    what = f"{n_inst} instances" if n_inst else f"{len(sc.array(rta.ARR_TRIS))} triangles"
    print(f"{a.scene}: {what}, set-up {t1 - t0:.2f} s; {a.width}x{a.height} x {a.spp} spp, depth {a.depth}: "
          f"{rays} rays in {dt:.2f} s ({rays / dt / 1e6:.0f} Mrays/s) -> {a.out}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
