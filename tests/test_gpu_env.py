"""-m gpu: the importance-sampled environment light (EnviormentLight, main.cpp:88-193; Image::lookup, Image.h:46-64) and the
emissive-mesh light on NON-uniform data, HIP path against the oracle through rt_render: a 16x8 map with a sun and a black
row (tables staged in LDS), a 64x32 map (2048 texels: the shading kernels' GlobalTables instantiation), env_offset != 0,
a map written as Radiance .hdr and read back through load_hdr, and a mesh light of more than 2048 emissive triangles
(GlobalTables again).  The oracle's side of these is pinned on the CPU by tests/test_oracle_env_pins.py.

Bar: RMSE over ALL pixels (no pixel is excluded) relative to the image's RMS < 1e-3 at 64 spp per pixel with the same
random streams; pixel-wise, at most 0.5 % of the pixels may differ beyond rel 2e-3 / abs 2e-4 (a path whose discrete
decision flips on a last-bit difference between ocml and libm) and none by more than the brightest value in the image."""
import numpy as np
import pytest

import scenes
from compare import compare_full
from test_oracle_env_pins import env_maps

pytestmark = pytest.mark.gpu


def _room(rta, rgb, degrees=0.0, light=True):
    sc = scenes.quad_room(rta, env=None, light=light)
    sc.set_env(rgb, degrees)
    return sc


def _both(rta, orc, gpu, sc, w=48, h=48, spp=64, depth=4, seed=21):
    sc.build()
    sc.upload(gpu)
    cam = rta.camera(rta.look_at((0.0, 1.0, 3.2), (0.0, 1.0, 0.0), (0, 1, 0)), 50.0, w, h)
    img, st = sc.render(cam, w, h, spp=spp, max_depth=depth, seed=seed)
    ref, oc = orc.render(orc.scene_from_host(sc), cam, w, h, spp, depth, seed=seed)
    assert st["paths"] == oc["paths"]
    assert abs(int(st["rays_shadow"]) - int(oc["rays_shadow"])) <= 0.002 * oc["rays_shadow"]
    return img, ref


@pytest.mark.parametrize("name,degrees,light", [("spot16x8", 0.0, True), ("spot16x8", 90.0, False), ("sun64x32", 0.0, True),
                                                ("sun64x32", -37.0, False)])
def test_non_uniform_environment(rta, orc, gpu, name, degrees, light):
    rgb = env_maps()[name]
    img, ref = _both(rta, orc, gpu, _room(rta, rgb, degrees, light))
    compare_full(img, ref)
    # the map matters: the same room under the map's mean colour is a different image
    flat = np.broadcast_to(rgb.mean(axis=(0, 1)), rgb.shape).copy()
    img_flat, _ = _both(rta, orc, gpu, _room(rta, flat, degrees, light), spp=16)
    assert np.sqrt(np.mean((img / 64 - img_flat / 16) ** 2)) > 0.02 * np.sqrt(np.mean((img / 64) ** 2))


def test_offset_turns_the_lighting(rta, orc, gpu):
    """env_offset = 0.25 against 0: different images (the sun moves a quarter turn), each equal to the oracle's."""
    rgb = env_maps()["spot16x8"]
    a, ra = _both(rta, orc, gpu, _room(rta, rgb, 0.0, False), spp=32)
    b, rb = _both(rta, orc, gpu, _room(rta, rgb, 90.0, False), spp=32)
    compare_full(a, ra)
    compare_full(b, rb)
    assert np.sqrt(np.mean((a - b) ** 2)) > 0.05 * np.sqrt(np.mean(a ** 2))


def _write_hdr_rle(path, rgb):
    """Radiance .hdr, new-style RLE scanlines need width >= 8; written flat (uncompressed RGBE) for any size."""
    h, w, _ = rgb.shape
    m = rgb.max(axis=2)
    e = np.where(m > 1e-32, np.floor(np.log2(np.maximum(m, 1e-38))) + 1, 0)
    scale = np.where(m > 1e-32, 256.0 / np.exp2(e), 0.0)
    rgbe = np.zeros((h, w, 4), np.uint8)
    rgbe[..., :3] = np.clip(rgb * scale[..., None], 0, 255).astype(np.uint8)
    rgbe[..., 3] = np.where(m > 1e-32, e + 128, 0).astype(np.uint8)
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n")
        f.write(f"-Y {h} +X {w}\n".encode())
        f.write(rgbe.tobytes())


def test_environment_loaded_from_hdr_file(rta, orc, gpu, tmp_path):
    """f-3: load_hdr (Image.cpp:6-30) -> EnviormentLight -> render.  The file holds RGBE-quantised texels; the oracle is
    given the image the loader produced, and the loader's output must be the decoder's m * 2^(e - 136)."""
    rgb = env_maps()["sun64x32"]
    path = tmp_path / "sky.hdr"
    _write_hdr_rle(str(path), rgb)
    sc = scenes.quad_room(rta, env=None)
    assert sc.load_env_hdr(str(path), 30.0)
    got = sc.env()
    # a shared exponent per texel: every channel within 1/128 of the texel's largest
    assert got.shape == rgb.shape and (np.abs(got - rgb) <= rgb.max(axis=2, keepdims=True) / 128 + 1e-7).all() and not np.array_equal(got, rgb)
    assert sc.info()["env_offset"] == pytest.approx(30.0 / 360.0)
    img, ref = _both(rta, orc, gpu, sc)
    compare_full(img, ref)


def test_mesh_light_with_more_than_2048_emissive_triangles(rta, orc, gpu):
    """A tessellated emissive sphere (2 x 48 x 23 = 2208 triangles > RT_LDS_CDF) as the only light: the area-CDF pick and
    the triangle reads come from global memory (GlobalTables); plus a non-uniform 16x8 map so both table kinds are live."""
    sc = scenes.quad_room(rta, env=None, light=False)
    emit = sc.add_material(rta.material(rta.MAT_EMISSIVE, (9, 7, 5)))
    v, t = scenes.uv_sphere(rta, 24, 48, material=emit)
    v["pos"] = v["pos"] * np.float32(0.22) + np.array([0.0, 1.55, 0.1], np.float32)
    sc.add_mesh(v, t)
    sc.set_env(env_maps()["spot16x8"], 10.0)
    sc.build()
    assert len(sc.array(rta.ARR_EMISSIVE)) == 2208 and len(sc.array(rta.ARR_LIGHTS)) == 1
    img, ref = _both(rta, orc, gpu, sc)
    compare_full(img, ref)
    assert img.mean() > 0.05 * 64
