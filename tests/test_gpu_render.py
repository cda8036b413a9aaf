"""-m gpu: the wavefront integrator (generate / closest / shade+NEE / shadow / accumulate) against the oracle's
get_ray_color restatement, same counter-based random stream, through rt_render.
Tolerance (tests/compare.py): transcendental functions differ by ULPs between ocml (GPU, fp32) and libm (oracle, double
narrowed), so per-pixel sums are compared at rel 2e-3 / abs 2e-4 with at most 0.5 % of pixels allowed to take a different
discrete decision (lobe choice, edge shadow ray), each bounded by the image's brightest value; relative RMSE over ALL
pixels < 1e-3 (the north star's image bar) at 32 samples per pixel."""
import numpy as np
import pytest

import scenes
from compare import compare_full

pytestmark = pytest.mark.gpu


def _compare(img, ref, frac_bad=0.005, rmse_bar=1e-3):
    return compare_full(img, ref, frac_bad, rmse_bar)


def _render_both(rta, orc, sc, cam, w, h, spp, depth, seed, **kw):
    osc = orc.scene_from_host(sc)
    img, st = sc.render(cam, w, h, spp=spp, max_depth=depth, seed=seed, **kw)
    ref, oc = orc.render(osc, cam, w, h, spp, depth, seed=seed)
    return img, st, ref, oc


@pytest.mark.parametrize("box", ["ggx", "beckmann", "matte", "disney_diffuse", "disney_metal", "uber", "glass",
                                 "disney_glass", "rough_dielectric", "clearcoat", "sheen"])
def test_room_materials(rta, orc, gpu, box):
    mats = {
        "ggx": rta.material(rta.MAT_GGX, s=(0.4, 0.0)),
        "beckmann": rta.material(rta.MAT_BECKMANN, s=(0.35, 0.0)),
        "matte": rta.material(rta.MAT_MATTE, (0.2, 0.4, 0.8)),
        "disney_diffuse": rta.material(rta.MAT_DISNEY_DIFFUSE, (0.725, 0.71, 0.68), s=(0.5, 0.5)),
        "disney_metal": rta.material(rta.MAT_DISNEY_METAL, (0.8, 0.2, 0.5), s=(0.4, 0.3)),
        "uber": rta.material(rta.MAT_DISNEY_UBER, (0.6, 0.05, 0.05), s=(0.0, 0.2, 0.5, 1.0, 0.3, 0.0, 0.0, 0, 0, 0, 0, 1.5)),
        "glass": rta.material(rta.MAT_GLASS, s=(1.5,)),
        "disney_glass": rta.material(rta.MAT_DISNEY_GLASS, (1.0, 0.9, 0.8), s=(0.2, 0.1, 1.8)),
        "rough_dielectric": rta.material(rta.MAT_ROUGH_DIELECTRIC, (1, 0.5, 0.5), (1, 1, 1), s=(1.3, 0.3)),
        "clearcoat": rta.material(rta.MAT_DISNEY_CLEARCOAT, s=(0.7,)),
        "sheen": rta.material(rta.MAT_DISNEY_SHEEN, (0.1, 0.1, 0.5), s=(0.3,)),
    }
    sc = scenes.quad_room(rta, box_material=mats[box])
    sc.build()
    sc.upload(gpu)
    cam = rta.camera(rta.look_at((0.0, 1.0, 3.2), (0.0, 1.0, 0.0), (0, 1, 0)), 45.0, 64, 64)
    img, st, ref, oc = _render_both(rta, orc, sc, cam, 64, 64, 32, 4, seed=3)
    print(box, "relative rmse over all pixels:", _compare(img, ref))
    assert st["paths"] == oc["paths"] == 64 * 64 * 32
    # ray counters: same definition on both sides (main.cpp:338,250,281); a handful of paths may end differently
    assert abs(int(st["rays_closest"]) - int(oc["rays_closest"])) <= 0.002 * oc["rays_closest"]
    assert abs(int(st["rays_shadow"]) - int(oc["rays_shadow"])) <= 0.002 * oc["rays_shadow"]


def test_cornellbox_depth8(rta, orc, gpu):
    sc = scenes.cornellbox(rta)
    sc.build()
    sc.upload(gpu)
    cam = scenes.cornell_camera(rta, sc, 96, 54)
    img, st, ref, oc = _render_both(rta, orc, sc, cam, 96, 54, 32, 8, seed=11)
    print("cornellbox depth 8, relative rmse over all pixels:", _compare(img, ref))


def test_window_split_and_tiles_equal_one_shot(rta, orc, gpu):
    """Sample windows (multi-GPU sample-pass sharding) and tiles (image-tile sharding) must reproduce the one-shot
    image: windows bit-exactly when accumulated in order, tiles bit-exactly."""
    sc = scenes.quad_room(rta)
    sc.build()
    sc.upload(gpu)
    w, h, spp = 48, 40, 8
    cam = rta.camera(rta.look_at((0.0, 1.0, 3.2), (0.0, 1.0, 0.0), (0, 1, 0)), 45.0, w, h)
    one, _ = sc.render(cam, w, h, spp=spp, max_depth=4, seed=5)
    acc = np.zeros((h, w, 3), np.float32)
    for b in range(0, spp, 2):
        sc.render(cam, w, h, spp=2, max_depth=4, seed=5, spp_begin=b, spp_total=spp, out=acc)
    assert np.array_equal(acc, one)
    tiled = np.zeros((h, w, 3), np.float32)
    for tile in [(0, 0, 20, 40), (20, 0, 48, 17), (20, 17, 48, 40)]:
        sc.render(cam, w, h, spp=spp, max_depth=4, seed=5, tile=tile, out=tiled)
    assert np.array_equal(tiled, one)
    # small passes inside one call as well
    small, _ = sc.render(cam, w, h, spp=spp, max_depth=4, seed=5, spp_per_pass=3)
    assert np.array_equal(small, one)


def test_resolve_and_render_image(rta, orc, gpu, tmp_path):
    sc = scenes.quad_room(rta)
    sc.build()
    sc.upload(gpu)
    view = rta.look_at((0.0, 1.0, 3.2), (0.0, 1.0, 0.0), (0, 1, 0))
    bmp = tmp_path / "out.bmp"
    rgb, rad, st = sc.render_image(view, 45.0, 40, 30, 4, 4, seed=2, bmp_path=str(bmp))
    assert bmp.stat().st_size == 54 + 40 * 3 * 30
    want = orc.resolve(rad, 4)
    assert np.abs(rgb.astype(int) - want.astype(int)).max() <= 1  # powf vs pow at a u8 rounding edge
    assert np.array_equal(sc.resolve(rad, 4), rgb)


def test_no_env_no_light(rta, orc, gpu):
    sc = scenes.quad_room(rta, light=False, env=None)
    sc.build()
    sc.upload(gpu)
    cam = rta.camera(rta.look_at((0.0, 1.0, 3.2), (0.0, 1.0, 0.0), (0, 1, 0)), 45.0, 16, 16)
    img, st = sc.render(cam, 16, 16, spp=2, max_depth=3)
    assert not img.any() and st["rays_shadow"] == 0


def test_device_out_equals_host_out(rta, orc, gpu):
    """RT_RENDER_DEVICE_OUT (the caller's HBM buffer, synchronised before the call as include/rt_hip.h asks) accumulates
    the same sums as the host-buffer form, over two calls into one buffer."""
    import torch

    sc = scenes.quad_room(rta)
    sc.build()
    sc.upload(gpu)
    w, h = 40, 24
    cam = rta.camera(rta.look_at((0.0, 1.0, 3.2), (0.0, 1.0, 0.0), (0, 1, 0)), 45.0, w, h)
    host, _ = sc.render(cam, w, h, spp=6, max_depth=4, seed=8, spp_total=12)
    sc.render(cam, w, h, spp=6, max_depth=4, seed=8, spp_begin=6, spp_total=12, out=host)
    dev = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    sc.render(cam, w, h, spp=6, max_depth=4, seed=8, spp_total=12, device_out=dev.data_ptr())
    sc.render(cam, w, h, spp=6, max_depth=4, seed=8, spp_begin=6, spp_total=12, device_out=dev.data_ptr())
    assert np.array_equal(dev.cpu().numpy(), host)
    sc.render(cam, w, h, spp=6, max_depth=4, seed=8, spp_total=12, device_out=dev.data_ptr(), clear=True)
    first, _ = sc.render(cam, w, h, spp=6, max_depth=4, seed=8, spp_total=12)
    assert np.array_equal(dev.cpu().numpy(), first)
