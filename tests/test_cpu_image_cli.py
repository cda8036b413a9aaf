"""Host-side: load_hdr (Radiance RGBE, Image.cpp:6-30 of the reference) and the command line.  The reference's own HDR
asset (noon_grass.hdr) is git-ignored and absent, so the files are written here: flat and run-length-encoded scanlines
must decode to mantissa * 2^(e - 136), the conversion of the decoder the reference links."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rgbe(rng, h, w):
    px = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    px[..., 3] = rng.integers(120, 136, size=(h, w), dtype=np.uint8)
    px[0, 0] = (10, 20, 30, 0)  # exponent 0 -> black
    return px


def _decode(px):
    f = np.ldexp(np.float32(1.0), px[..., 3].astype(np.int32) - 136).astype(np.float32)
    out = px[..., :3].astype(np.float32) * f[..., None]
    out[px[..., 3] == 0] = 0
    return out


def _write_hdr(path, px, rle):
    h, w, _ = px.shape
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\n# made by a test\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n")
        f.write(f"-Y {h} +X {w}\n".encode())
        for y in range(h):
            if not rle:
                f.write(px[y].tobytes())
                continue
            f.write(bytes([2, 2, w >> 8, w & 255]))
            for c in range(4):
                row, x = px[y, :, c], 0
                while x < w:
                    run = 1
                    while x + run < w and run < 127 and row[x + run] == row[x]:
                        run += 1
                    if run >= 3:
                        f.write(bytes([128 + run, int(row[x])]))
                        x += run
                    else:
                        n = min(w - x, 5)
                        f.write(bytes([n]) + row[x:x + n].tobytes())
                        x += n


@pytest.mark.parametrize("rle,w,h", [(False, 5, 3), (False, 16, 8), (True, 16, 8), (True, 200, 7)])
def test_load_hdr_decodes_rgbe(rta, tmp_path, rle, w, h):
    rng = np.random.default_rng(7 + w)
    px = _rgbe(rng, h, w)
    if rle:
        px[2, 3:60 if w > 60 else 9] = px[2, 3]  # a long run
    path = tmp_path / "env.hdr"
    _write_hdr(path, px, rle)
    sc = rta.Scene()
    assert sc.load_env_hdr(str(path))
    got = sc.env()
    assert got.shape == (h, w, 3) and np.array_equal(got, _decode(px))


def test_load_hdr_rejects_bad_files(rta, tmp_path, capfd):
    sc = rta.Scene()
    assert not sc.load_env_hdr(str(tmp_path / "missing.hdr"))
    (tmp_path / "junk.hdr").write_bytes(b"P6\n1 1\n255\n\0\0\0")
    assert not sc.load_env_hdr(str(tmp_path / "junk.hdr"))
    px = _rgbe(np.random.default_rng(1), 4, 16)
    _write_hdr(tmp_path / "cut.hdr", px, True)
    data = (tmp_path / "cut.hdr").read_bytes()
    (tmp_path / "cut.hdr").write_bytes(data[:-20])
    assert not sc.load_env_hdr(str(tmp_path / "cut.hdr"))
    assert sc.env() is None
    assert "Couldn't load image" in capfd.readouterr().out


def test_cli_without_gpu_fails_loudly():
    """No CPU fallback: without a HIP device the command line says so and exits non-zero."""
    import raytracing_amd as rta
    if rta.device_count() > 0:
        pytest.skip("a GPU is present")
    out = subprocess.run([sys.executable, "-m", "raytracing_amd", os.path.join(ROOT, "tests", "golden", "cornellbox.glb")], cwd=ROOT,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 2 and "no HIP device" in out.stderr


@pytest.mark.gpu
def test_cli_renders_a_bmp(gpu, tmp_path):
    px = _rgbe(np.random.default_rng(3), 8, 16)
    px[..., 3] = 127
    _write_hdr(tmp_path / "sky.hdr", px, True)
    bmp = tmp_path / "out.bmp"
    out = subprocess.run([sys.executable, "-m", "raytracing_amd", os.path.join(ROOT, "tests", "golden", "cornellbox.glb"), "-o", str(bmp),
                          "--width", "64", "--height", "36", "--spp", "4", "--depth", "4", "--env-hdr", str(tmp_path / "sky.hdr")],
                         cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    data = bmp.read_bytes()
    assert data[:2] == b"BM" and len(data) >= 54 + 64 * 36 * 3 and "Mrays/s" in out.stdout
    assert np.frombuffer(data[54:], np.uint8).mean() > 2  # not black
