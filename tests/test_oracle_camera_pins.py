"""CPU: the oracle's ray generation and resolve pinned by float64 restatements written from the reference text:
look_at (Math.h:595-610), the Camera constructor's raster_to_world = view * screen_to_camera * raster_to_screen
(Scene.h:14-44), hammersley_2d (Utils.h:138-151), the sample loop's coords = hammersley + (x, y) (main.cpp:976-980),
Camera::get_ray with GaussianFilter(0.5) / BoxFilter (Scene.h:61-75, Filter.h:12-42), and Renderer::add_to_buffer
(main.cpp:865-877) with the row flip of main.cpp:990.  The oracle's rays (fp32) must agree with the float64 evaluation to
fp32 rounding: origin exactly, direction within 2e-6 — except the one sample whose Gaussian radius comes from the clamp
max(dx, 1e-8) (sample 0: dx = 0), which is checked against its closed form, r = 0.5 * sqrt(-2 ln 1e-8) = 3.035 pixels."""
import math

import numpy as np
import pytest


def look_at(eye, at, up):  # Math.h:595-610 -> column-major 16
    eye, at, up = (np.asarray(v, np.float64) for v in (eye, at, up))
    front = (at - eye) / np.linalg.norm(at - eye)
    z = -front
    x = np.cross(up, z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = x, y, z, eye  # transpose(mat4(x.x, y.x, z.x, p.x, ...)): columns x, y, z, p
    return m


def raster_to_world(view, fov, w, h):  # Scene.h:22-40 (mat4's sixteen-float constructor fills COLUMNS: Math.h:352-360)
    r2s = np.array([[2.0 / w, 0, 0, -1], [0, 2.0 / h, 0, -1], [0, 0, 1, 1], [0, 0, 0, 1]])
    t = math.tan(math.radians(fov / 2.0))
    s2c = np.diag([t * (w / h), t, -1.0, 1.0])
    return view @ s2c @ r2s


def radical_inverse(i):  # Utils.h:138-146
    return int(f"{i:032b}"[::-1], 2) * 2.3283064365386963e-10


def get_ray(view, r2w, rx, ry, box_width=None):  # Scene.h:61-75
    dx, dy = rx - math.floor(rx), ry - math.floor(ry)
    if box_width is None:  # GaussianFilter(0.5), Filter.h:34-38
        r = 0.5 * math.sqrt(-2.0 * math.log(max(dx, 1e-8)))
        ox, oy = r * math.cos(2 * math.pi * dy), r * math.sin(2 * math.pi * dy)
    else:  # BoxFilter, Filter.h:18-23
        ox, oy = (2 * dx - 1) * box_width * 0.5, (2 * dy - 1) * box_width * 0.5
    px, py = math.floor(rx) + 0.5 + ox, math.floor(ry) + 0.5 + oy
    pos = (r2w @ np.array([px, py, 0.0, 1.0]))[:3]
    eye = view[:3, 3]
    d = pos - eye
    return eye, d / np.linalg.norm(d)


@pytest.mark.parametrize("w,h,fov", [(64, 48, 45.0), (96, 54, 39.6), (31, 17, 75.0)])
def test_primary_rays_match_the_camera_written_from_the_reference_text(orc, w, h, fov):
    eye, at, up = (1.5729, 2.3375, 0.6281), (0.6, 2.15, 0.47), (0.0, 1.0, 0.0)
    view = look_at(eye, at, up)
    oview = orc.look_at(eye, at, up)
    assert np.allclose(oview.reshape(4, 4).T, view, atol=2e-6)
    cam = orc.camera(oview, fov, w, h)
    r2w = raster_to_world(view, fov, w, h)
    n = 16
    worst = 0.0
    for i in (1, 2, 7, 15):
        rays = orc.camera_rays(cam, w, h, i, n)
        hx, hy = np.float32(i) / np.float32(n), np.float32(radical_inverse(i))
        for y in range(0, h, 5):
            for x in range(0, w, 7):
                # coords = hammersley_2d(i, n) + vec2(x, y) in fp32 (main.cpp:978-979): the fractional part the filter sees is fp32's
                rx, ry = float(np.float32(hx + np.float32(x))), float(np.float32(hy + np.float32(y)))
                o, d = get_ray(view, r2w, rx, ry)
                got = rays[y * w + x]
                assert np.allclose(got["o"], o, atol=1e-6)
                err = float(np.abs(got["d"] - d).max())
                worst = max(worst, err)
                assert err < 2e-6, (i, x, y, got["d"], d)
    # sample 0: dx = 0 exactly -> max(dx, 1e-8): the ray leaves 3.035 pixels to the right of the pixel centre (cos 0 = 1)
    rays = orc.camera_rays(cam, w, h, 0, n)
    for x, y in ((3, 4), (w - 2, h - 3)):
        o, d = get_ray(view, r2w, float(x), float(y))
        assert np.abs(rays[y * w + x]["d"] - d).max() < 2e-6
        centre = (r2w @ np.array([x + 0.5 + 0.5 * math.sqrt(-2 * math.log(1e-8)), y + 0.5, 0, 1.0]))[:3] - view[:3, 3]
        assert np.abs(d - centre / np.linalg.norm(centre)).max() < 1e-12 and abs(0.5 * math.sqrt(-2 * math.log(1e-8)) - 3.0349) < 1e-3
    print(f"camera pins {w}x{h}: worst direction error {worst:.2e}")


def test_resolve_matches_add_to_buffer(orc):
    """Renderer::add_to_buffer (main.cpp:865-877): scale by 1/spp, pow(c, 1/2.2), clamp to [0, 1], * 255 truncated, row H-y-1."""
    rng = np.random.default_rng(3)
    h, w, spp = 9, 13, 7
    sums = (rng.random((h, w, 3)) * 9.0).astype(np.float32)
    sums[0, 0] = (0.0, 7.0, 70.0)  # black, exactly one, far above one
    got = orc.resolve(sums, spp, 2.2)
    c = np.power(sums.astype(np.float64) * (1.0 / spp), 1 / 2.2)
    want = (np.clip(c, 0, 1) * 255).astype(np.uint8)[::-1]
    diff = np.abs(got.astype(int) - want.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 0.02  # (a value a hair under an integer may truncate the other way in fp32)
    assert tuple(got[h - 1, 0]) == (0, 255, 255)
