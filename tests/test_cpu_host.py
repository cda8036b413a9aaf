"""CPU suite (-m "not gpu"): host logic against the oracle, the C-ABI surface, and loud failure without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol(rta):
    header = open(os.path.join(ROOT, "include", "rt_hip.h")).read()
    declared = sorted(set(re.findall(r"^\s*(?:const\s+char\s*\*|int64_t|int|void)\s+(rt_[a-z_]+)\s*\(", header, re.M)))
    assert declared == sorted(rta.HIP_SYMBOLS)
    lib = ctypes.CDLL(os.path.join(ROOT, "raytracing_amd", "librt_hip.so"))
    for name in declared:
        assert hasattr(lib, name), name


def test_struct_sizes_match_header(rta):
    # sizes the header's comments promise (and the kernels index by)
    assert rta.PAIR.itemsize == 64 and rta.FLAT_TRI.itemsize == 48 and rta.FLAT_NORMALS.itemsize == 48
    assert rta.MATERIAL.itemsize == 80 and rta.NODE.itemsize == 32
    assert ctypes.sizeof(rta.RtCamera) == 84 and ctypes.sizeof(rta.RtOptions) == 52


def _random_bounds(n, seed, degenerate=False):
    rng = np.random.default_rng(seed)
    c = rng.random((n, 3), dtype=np.float32)
    e = rng.random((n, 3), dtype=np.float32) * 0.05
    if degenerate:
        c[: n // 2] = c[0]  # coincident primitives: the reference's uninitialised-split case
        e[: n // 2] = e[0]
    return np.concatenate([c - e, c + e], axis=1).astype(np.float32)


@pytest.mark.parametrize("n,leaf,strategy,degenerate", [(1, 1, 1, False), (2, 1, 1, False), (257, 1, 1, False), (300, 4, 1, False),
                                                        (500, 1, 0, False), (300, 2, 0, False), (64, 1, 1, True), (65, 3, 0, True)])
def test_bvh_build_matches_oracle_bit_for_bit(rta, orc, n, leaf, strategy, degenerate):
    b = _random_bounds(n, 7 * n + leaf, degenerate)
    hn, hi = rta.bvh_build(b, leaf, strategy)
    on, oi = orc.bvh_build(b, leaf, strategy)
    assert hn.tobytes() == on.tobytes()
    assert np.array_equal(hi, oi)
    # structural invariants of BVH.cpp:58-128
    assert sorted(hi.tolist()) == list(range(n))
    if leaf == 1:
        assert len(hn) == 2 * n - 1
    branches = hn[hn["count"] == -1]
    assert np.all(branches["left_node"] % 2 == 1)  # children allocated as adjacent pairs after the root
    leaves = hn[hn["count"] != -1]
    assert leaves["count"].sum() == n and leaves["count"].max() <= leaf


def test_threaded_sah_build_is_bit_identical(rta, orc, monkeypatch):
    """Above 32 768 primitives the SAH build spreads candidate planes and small subtrees over threads; node numbering
    is fixed in advance (2n - 1 nodes per n-primitive subtree), so nodes and indices equal the oracle's one-thread
    recursion bit for bit — including coincident primitives (median fallback) inside the threaded part."""
    n = 40_000
    b = _random_bounds(n, 99, False)
    b[1000:1400] = b[1000]  # 400 coincident boxes: no plane separates them
    monkeypatch.setenv("RT_BVH_THREADS", "5")
    hn, hi = rta.bvh_build(b, 1, 1)
    on, oi = orc.bvh_build(b, 1, 1)
    assert len(hn) == 2 * n - 1 and hn.tobytes() == on.tobytes() and np.array_equal(hi, oi)
    monkeypatch.setenv("RT_BVH_THREADS", "1")
    sn, si = rta.bvh_build(b, 1, 1)
    assert sn.tobytes() == hn.tobytes() and np.array_equal(si, hi)


@pytest.mark.parametrize("leaf,strat", [(2, 1), (4, 1), (3, 0), (1, 0)])
def test_threaded_build_other_leaf_sizes_is_bit_identical(rta, orc, monkeypatch, leaf, strat):
    """Larger leaves and BVH_MIDDLE have no subtree size known in advance: subtrees are built into private arrays and
    renumbered in the recursion's order afterwards.  Same bytes as the oracle's recursion and as the one-thread build."""
    n = 40_000
    b = _random_bounds(n, 7 + leaf, False)
    b[2000:2300] = b[2000]
    monkeypatch.setenv("RT_BVH_THREADS", "5")
    hn, hi = rta.bvh_build(b, leaf, strat)
    on, oi = orc.bvh_build(b, leaf, strat)
    assert hn.tobytes() == on.tobytes() and np.array_equal(hi, oi)
    monkeypatch.setenv("RT_BVH_THREADS", "1")
    sn, si = rta.bvh_build(b, leaf, strat)
    assert sn.tobytes() == hn.tobytes() and np.array_equal(si, hi)


@pytest.mark.parametrize("leaf,flat", [(1, False), (2, True)])
def test_binned_sah_costs_equal_the_sweep_bit_for_bit(rta, orc, monkeypatch, leaf, flat):
    """The host builder gets all 300 candidate costs of a node from one binning pass (unions are min / max, counts integers: exact),
    with several threads binning one large range; the oracle sweeps every candidate over every primitive as the reference does
    (BVH.cpp:30-46).  140 000 boxes — enough for two binning threads at the root — with coincident boxes, signed zeros, many equal
    centres on plane values, and optionally one flat axis (lo == hi: no candidates on it): same nodes, same indices."""
    n = 140_000
    b = _random_bounds(n, 1234 + leaf, False)
    b[5000:5600] = b[5000]
    grid = (np.arange(3000) % 101).astype(np.float32) / 100.0  # centres that sit exactly on candidate planes of the root
    b[10_000:13_000, 0] = grid - 0.01
    b[10_000:13_000, 3] = grid + 0.01
    b[20_000:20_100, 1] = -0.0
    b[20_100:20_200, 1] = 0.0
    if flat:
        b[:, 2] = 0.25
        b[:, 5] = 0.25
    monkeypatch.setenv("RT_BVH_THREADS", "5")
    hn, hi = rta.bvh_build(b, leaf, 1)
    on, oi = orc.bvh_build(b, leaf, 1)
    assert hn.tobytes() == on.tobytes() and np.array_equal(hi, oi)
    monkeypatch.setenv("RT_BVH_THREADS", "1")
    sn, si = rta.bvh_build(b, leaf, 1)
    assert sn.tobytes() == hn.tobytes() and np.array_equal(si, hi)


def test_large_ranges_gather_in_parallel_with_the_same_result(rta, orc, monkeypatch):
    """Ranges of 262 144 primitives and more evaluate a split's predicate and a node's box on all threads and then apply
    std::partition's own permutation: 300 000 boxes, BVH_MIDDLE against the oracle's recursion (partition + box on every level),
    BVH_SAH several threads against one thread (which the other tests hold against the oracle)."""
    n = 300_000
    b = _random_bounds(n, 4321, False)
    b[7000:7400] = b[7000]
    monkeypatch.setenv("RT_BVH_THREADS", "5")
    hn, hi = rta.bvh_build(b, 1, 0)
    on, oi = orc.bvh_build(b, 1, 0)
    assert hn.tobytes() == on.tobytes() and np.array_equal(hi, oi)
    tn, ti = rta.bvh_build(b, 2, 1)
    monkeypatch.setenv("RT_BVH_THREADS", "1")
    sn, si = rta.bvh_build(b, 2, 1)
    assert sn.tobytes() == tn.tobytes() and np.array_equal(si, ti)


def test_tri_bounds_and_mesh_build(rta, orc):
    sc = scenes.quad_room(rta)
    sc.build()
    verts, tris = sc.array(rta.ARR_VERTS), sc.array(rta.ARR_TRIS)
    nodes, idx = sc.array(rta.ARR_NODES), sc.array(rta.ARR_INDICES)
    on, oi = orc.bvh_build(orc.tri_bounds(verts, tris), 1, 1)  # Mesh::build == BVH::build(tri bounds, 1, BVH_SAH)
    assert nodes.tobytes() == on.tobytes() and np.array_equal(idx, oi)


def _walk_flat(rta, sc):
    """Checks the flattened pair/triangle arrays describe exactly the host BVH."""
    nodes, idx = sc.array(rta.ARR_NODES), sc.array(rta.ARR_INDICES)
    pairs, ftris = sc.array(rta.ARR_FLAT_PAIRS), sc.array(rta.ARR_FLAT_TRIS)
    verts, tris = sc.array(rta.ARR_VERTS), sc.array(rta.ARR_TRIS)
    assert len(pairs) == np.count_nonzero(nodes["count"] == -1)
    assert np.array_equal(ftris["tri_id"] & 0x7FFFFFFF, idx.astype(np.uint32))
    p = verts["pos"][tris["idx"][idx]]
    assert np.array_equal(ftris["v0"], p[:, 0])
    assert np.array_equal(ftris["e1"], p[:, 1] - p[:, 0]) and np.array_equal(ftris["e2"], p[:, 2] - p[:, 0])
    assert np.array_equal(ftris["material"], tris["material"][idx].astype(np.int32))
    stack = [(0, sc.info()["root_ref"])]
    seen = 0
    while stack:
        node, ref = stack.pop()
        nd = nodes[node]
        if nd["count"] != -1:
            assert ref == ~int(nd["left_node"])
            last = nd["left_node"] + nd["count"] - 1
            flags = ftris["tri_id"][nd["left_node"]: last + 1] >> 31
            assert flags[-1] == 1 and not flags[:-1].any()
            seen += int(nd["count"])
            continue
        pr = pairs[ref]
        l, r = nodes[nd["left_node"]], nodes[nd["left_node"] + 1]
        assert np.array_equal(pr["lmin"], l["bmin"]) and np.array_equal(pr["lmax"], l["bmax"])
        assert np.array_equal(pr["rmin"], r["bmin"]) and np.array_equal(pr["rmax"], r["bmax"])
        stack.append((int(nd["left_node"]), int(pr["lref"])))
        stack.append((int(nd["left_node"]) + 1, int(pr["rref"])))
    assert seen == len(tris)


def test_flatten_describes_the_bvh(rta):
    sc = scenes.quad_room(rta)
    sc.build()
    _walk_flat(rta, sc)
    sc2 = scenes.quad_room(rta)
    sc2.build(max_per_leaf=4, strategy=0)
    _walk_flat(rta, sc2)


def test_light_tables(rta):
    sc = scenes.quad_room(rta)
    sc.build()
    lights = sc.array(rta.ARR_LIGHTS)
    assert len(lights) == 1 and lights["kind"][0] == 0
    assert lights["area"][0] == pytest.approx(0.6 * 0.5)  # Rectangle::area, Object.h:121-123
    tl = sc.array(rta.ARR_TRI_LIGHT)
    assert np.count_nonzero(tl == 0) == 2 and np.count_nonzero(tl >= 0) == 2
    mats = sc.array(rta.ARR_MATERIALS)
    assert mats["type"][lights["material"][0]] == rta.MAT_EMISSIVE


def test_gltf_loader_cornellbox(rta):
    sc = scenes.cornellbox(rta)
    verts, tris = sc.array(rta.ARR_VERTS), sc.array(rta.ARR_TRIS)
    assert len(tris) == 13170 and len(verts) == 8515
    counts = np.bincount(tris["material"], minlength=6)
    assert counts.tolist() == [18, 12, 2, 2, 12168, 968]
    p = verts["pos"]
    assert np.allclose(p.min(0), (-1.72733, -0.0256, -1.77316), atol=1e-4)
    assert np.allclose(p.max(0), (1.72733, 3.40802, 1.77316), atol=1e-4)
    assert np.allclose(np.linalg.norm(verts["normal"], axis=1), 1.0, atol=1e-4)
    info = sc.info()
    assert info["has_camera"]
    view = info["camera_view"].reshape(4, 4)  # rows = columns of the matrix
    assert np.allclose(view[3, :3], (1.5729, 2.3375, 0.6281), atol=1e-4)
    assert np.allclose(-view[2, :3], (-0.972, -0.181, -0.152), atol=1e-3)  # forward, SURVEY §8(d)
    sc.build()
    lights = sc.array(rta.ARR_LIGHTS)
    assert len(lights) == 1 and lights["kind"][0] == 1 and lights["count"][0] == 968  # Suzanne, the only emitter
    em = sc.array(rta.ARR_EMISSIVE)
    assert np.all(np.diff(em["cdf"]) >= 0) and em["cdf"][-1] == 1.0
    mats = sc.array(rta.ARR_MATERIALS)
    assert mats["type"].tolist() == [rta.MAT_GGX, rta.MAT_DISNEY_METAL, rta.MAT_GGX, rta.MAT_BECKMANN,
                                     rta.MAT_ROUGH_DIELECTRIC, rta.MAT_EMISSIVE]


def test_gltf_loader_rejects_garbage(rta, tmp_path):
    bad = tmp_path / "bad.glb"
    bad.write_bytes(b"glTF" + b"\x00" * 40)
    sc = rta.Scene()
    with pytest.raises(rta.RtError):
        sc.load_gltf(str(bad))
    with pytest.raises(rta.RtError):
        rta.Scene().load_gltf(str(tmp_path / "missing.glb"))


def _tiny_gltf(tmp_path, name, indices, n_normals=3, material=0, index_type=5123):
    """A one-triangle .gltf with an embedded base64 buffer: 3 positions, n_normals normals, the given indices."""
    import base64
    import json
    import struct

    pos = struct.pack("<9f", 0, 0, 0, 1, 0, 0, 0, 1, 0)
    nrm = struct.pack(f"<{3 * n_normals}f", *([0, 0, 1] * n_normals))
    idx = struct.pack({5123: f"<{len(indices)}H", 5125: f"<{len(indices)}I", 5126: f"<{len(indices)}f"}[index_type], *indices)
    idx += b"\0" * (-len(idx) % 4)
    blob = pos + nrm + idx
    doc = {
        "asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}], "nodes": [{"mesh": 0}],
        "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 1}, "indices": 2, "material": material}]}],
        "materials": [{"pbrMetallicRoughness": {"baseColorFactor": [0.5, 0.5, 0.5, 1]}}],
        "buffers": [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode()}],
        "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 36}, {"buffer": 0, "byteOffset": 36, "byteLength": len(nrm)},
                        {"buffer": 0, "byteOffset": 36 + len(nrm), "byteLength": len(idx)}],
        "accessors": [{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3"},
                      {"bufferView": 1, "componentType": 5126, "count": n_normals, "type": "VEC3"},
                      {"bufferView": 2, "componentType": index_type, "count": len(indices), "type": "SCALAR"}],
    }
    path = tmp_path / name
    path.write_text(json.dumps(doc))
    return str(path)


def test_gltf_loader_checks_indices_and_accessor_counts(rta, tmp_path):
    """A well-formed one-triangle file loads; an index beyond the vertex count, an index count that is not a multiple of
    three, a NORMAL accessor shorter than POSITION, a negative material and a float index accessor all fail the load
    (the reference asserts on malformed files, Scene.cpp:100,168) instead of reading past a buffer."""
    good = rta.Scene()
    good.load_gltf(_tiny_gltf(tmp_path, "good.gltf", [0, 1, 2]))
    assert len(good.array(rta.ARR_TRIS)) == 1 and len(good.array(rta.ARR_VERTS)) == 3
    for name, kw in (("oob.gltf", dict(indices=[0, 1, 7])), ("ragged.gltf", dict(indices=[0, 1, 2, 0])),
                     ("short_normals.gltf", dict(indices=[0, 1, 2], n_normals=2)), ("neg_mat.gltf", dict(indices=[0, 1, 2], material=-3)),
                     ("float_idx.gltf", dict(indices=[0, 1, 2], index_type=5126))):
        with pytest.raises(rta.RtError):
            rta.Scene().load_gltf(_tiny_gltf(tmp_path, name, **kw))


def test_host_camera_matches_oracle(rta, orc):
    view = rta.look_at((-1.0, 0.5, 0.8), (0.0, 0.2, 0.0), (0, 1, 0))  # main.cpp:719
    assert np.array_equal(view, orc.look_at((-1.0, 0.5, 0.8), (0.0, 0.2, 0.0), (0, 1, 0)))
    cam = rta.camera(view, 45.0, 64, 48)
    ocam = orc.camera(view, 45.0, 64, 48)
    assert np.array_equal(np.frombuffer(cam, np.float32, 19), np.frombuffer(ocam, np.float32, 19))
    # Camera::get_ray(raster) on the host == oracle's hammersley + get_ray for the same raster position
    n = 16
    for (x, y, i) in [(0, 0, 0), (5, 7, 3), (63, 47, 15), (31, 2, 8)]:
        bits = int("{:032b}".format(i)[::-1], 2)
        rx = np.float32(np.float32(i) / np.float32(n)) + np.float32(x)
        ry = np.float32(np.float32(bits) * np.float32(2.3283064365386963e-10)) + np.float32(y)
        want = orc.camera_rays(ocam, 64, 48, i, n, tile=(x, y, x + 1, y + 1))[0]
        got = rta.camera_ray(view, 45.0, 64, 48, float(rx), float(ry))
        assert np.array_equal(got["o"], want["o"]) and np.array_equal(got["d"], want["d"])


def test_gpu_entry_points_fail_loudly_without_a_gpu(rta):
    if rta.device_count() > 0:
        pytest.skip("a GPU is visible")
    sc = scenes.quad_room(rta)
    sc.build()
    with pytest.raises(rta.RtError, match="no HIP device"):
        sc.upload(0)
    with pytest.raises(rta.RtError):
        sc.trace_closest(np.zeros(1, rta.RAY))  # no silent CPU path


def test_sharding_helpers():
    from raytracing_amd import sharding

    wins = [sharding.sample_window(r, 4, 256) for r in range(4)]
    assert wins == [(0, 256, 1024), (256, 256, 1024), (512, 256, 1024), (768, 256, 1024)]
    w, h = 200, 130
    cover = np.zeros((h, w), int)
    for r in range(3):
        for (x0, y0, x1, y1) in sharding.tiles_for_rank(w, h, r, 3, extent=64):
            cover[y0:y1, x0:x1] += 1
    assert (cover == 1).all()
    with pytest.raises(ValueError):
        sharding.sample_window(4, 4, 1)


def test_host_partitions_refuse_nonsense(rta):
    """rth_shard_tiles / rth_shard_sample_window with world 0, a rank outside the world or a zero tile edge return nothing
    (world 0 used to loop for ever, tile 0 to divide by zero)."""
    assert len(rta.shard_tiles(0, 0, 64, 64, 16)) == 0
    assert len(rta.shard_tiles(2, 2, 64, 64, 16)) == 0
    assert len(rta.shard_tiles(-1, 2, 64, 64, 16)) == 0
    assert len(rta.shard_tiles(0, 1, 64, 64, 0)) == 0
    assert len(rta.shard_tiles(0, 1, 64, 64, 16)) == 16
    assert rta.shard_sample_window(0, 0, 64) == (0, 0)
    assert rta.shard_sample_window(3, 2, 64) == (0, 0)
    assert rta.shard_sample_window(1, 2, 65) == (33, 32)


def test_add_instance_flattens_with_the_reference_transform(rta):
    """Scene::add_instance = Instance(TriangleMesh, material, transform) flattened: positions through the matrix,
    normals through the normalised inverse transpose (Object.h:388-391); identity = add_mesh."""
    v, t = scenes.uv_sphere(rta, 6, 8)
    a, b = rta.Scene(), rta.Scene()
    for s in (a, b):
        s.add_material(rta.material(rta.MAT_MATTE, (0.5, 0.5, 0.5)))
    a.add_mesh(v, t)
    b.add_instance(v, t, np.eye(4, dtype=np.float32).reshape(-1))
    av, bv = a.array(rta.ARR_VERTS), b.array(rta.ARR_VERTS)
    assert np.array_equal(av["pos"], bv["pos"]) and np.allclose(av["normal"], bv["normal"], atol=1e-6)  # (normals are re-normalised)
    assert np.array_equal(a.array(rta.ARR_TRIS), b.array(rta.ARR_TRIS))
    m = np.array([[0.0, -2.0, 0.0, 1.0], [0.5, 0.0, 0.0, 2.0], [0.0, 0.0, 3.0, -1.0], [0, 0, 0, 1]], np.float64)  # rotation x non-uniform scale + translation
    c = rta.Scene()
    mid = c.add_material(rta.material(rta.MAT_MATTE, (0.5, 0.5, 0.5)))
    c.add_material(rta.material(rta.MAT_GGX, s=(0.3, 0.0)))
    c.add_mesh(v, t)
    c.add_instance(v, t, m.T.reshape(-1).astype(np.float32), material=1)
    cv, ct = c.array(rta.ARR_VERTS), c.array(rta.ARR_TRIS)
    assert len(cv) == 2 * len(v) and len(ct) == 2 * len(t)
    want = (m[:3, :3] @ v["pos"].astype(np.float64).T).T + m[:3, 3]
    assert np.allclose(cv["pos"][len(v):], want, rtol=1e-6, atol=1e-6)
    nn = (np.linalg.inv(m[:3, :3]).T @ v["normal"].astype(np.float64).T).T
    nn /= np.linalg.norm(nn, axis=1, keepdims=True)
    assert np.allclose(cv["normal"][len(v):], nn, atol=1e-5)
    assert np.array_equal(ct["idx"][len(t):], t["idx"] + len(v)) and (ct["material"][len(t):] == 1).all() and (ct["material"][:len(t)] == mid).all()
    c.build()
    assert c.info()["root_ref"] >= 0


def test_image_writers_round_trip(rta, tmp_path):
    """Renderer output (main.cpp:1015 writes a BMP; PNG is this build's addition): both files decode back to the buffer,
    rows top-down in the buffer.  The PNG is checked chunk by chunk (CRC-32, zlib stream with Adler-32, filter 0)."""
    import struct
    import zlib
    rng = np.random.default_rng(5)
    for w, h in ((1, 1), (37, 23), (300, 250)):  # 300x250x3 > 65535 bytes: several stored blocks
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        png, bmp = tmp_path / f"o{w}.PNG", tmp_path / f"o{w}.bmp"
        rta.write_image(str(png), img)
        rta.write_image(str(bmp), img)
        data = png.read_bytes()
        assert data[:8] == b"\x89PNG\r\n\x1a\n"
        at, chunks = 8, []
        while at < len(data):
            n, typ = struct.unpack(">I4s", data[at:at + 8])
            body = data[at + 8:at + 8 + n]
            assert struct.unpack(">I", data[at + 8 + n:at + 12 + n])[0] == zlib.crc32(typ + body)
            chunks.append((typ, body))
            at += 12 + n
        assert [c[0] for c in chunks] == [b"IHDR", b"IDAT", b"IEND"]
        assert struct.unpack(">IIBBBBB", chunks[0][1]) == (w, h, 8, 2, 0, 0, 0)
        raw = np.frombuffer(zlib.decompress(chunks[1][1]), np.uint8).reshape(h, 1 + 3 * w)
        assert np.all(raw[:, 0] == 0) and np.array_equal(raw[:, 1:].reshape(h, w, 3), img)
        b = bmp.read_bytes()
        assert b[:2] == b"BM" and struct.unpack("<ii", b[18:26]) == (w, h)
        row = (3 * w + 3) & ~3
        px = np.frombuffer(b[54:], np.uint8).reshape(h, row)[::-1, :3 * w].reshape(h, w, 3)[:, :, ::-1]
        assert np.array_equal(px, img)
    with pytest.raises(OSError):
        rta.write_image(str(tmp_path / "no_such_dir" / "x.png"), img)


def test_inverse_is_the_references_fp32_adjugate(rta):
    """inverse(mat4) (Math.h:454-542): fp32 cofactors over an fp32 determinant, terms summed left to right in the
    reference's order; det == 0 returns the matrix unchanged.  Known answers + the first cofactor recomputed by hand."""
    f = np.float32
    # translation and scale: exact
    m = np.eye(4, dtype=np.float32)
    m[:3, 3] = (1.5, -2.0, 0.25)
    m[0, 0], m[1, 1], m[2, 2] = 2.0, 4.0, 0.5
    inv, det = rta.mat4_inverse(m.T.reshape(-1))
    assert det == 4.0
    want = np.eye(4)
    want[0, 0], want[1, 1], want[2, 2] = 0.5, 0.25, 2.0
    want[:3, 3] = (-0.75, 0.5, -0.5)
    assert np.array_equal(inv.reshape(4, 4).T, want.astype(np.float32))
    # a general affine matrix: close to the float64 inverse, and NOT what a double-precision elimination rounds to everywhere
    rng = np.random.default_rng(4)
    a = np.eye(4, dtype=np.float32)
    a[:3, :] = rng.normal(size=(3, 4)).astype(np.float32)
    c = a.T.reshape(-1).copy()  # column-major: c[4 * col + row]
    inv, det = rta.mat4_inverse(c)
    exact = np.linalg.inv(a.astype(np.float64))
    assert np.allclose(inv.reshape(4, 4).T, exact, rtol=2e-5, atol=2e-6) and det == pytest.approx(np.linalg.det(a.astype(np.float64)), rel=1e-5)
    # element 0 by hand, in the reference's order (Math.h:461-465) with fp32 rounding after every operation
    t = lambda i, j, k: f(f(c[i] * c[j]) * c[k])
    e0 = f(f(f(f(f(t(9, 14, 7) - t(13, 10, 7)) + t(13, 6, 11)) - t(5, 14, 11)) - t(9, 6, 15)) + t(5, 10, 15))
    assert inv[0] == f(f(f(1.0) / f(det)) * e0)
    # singular: unchanged (Math.h:457-458)
    s = np.zeros(16, np.float32)
    s[0] = s[5] = 1.0
    out, det = rta.mat4_inverse(s)
    assert det == 0.0 and np.array_equal(out, s)


def test_gltf_loader_survives_corrupted_files(rta, tmp_path):
    """80 corrupted copies of cornellbox.glb (digits of the JSON chunk, header and chunk lengths, bytes of the binary chunk,
    truncation): load_gltf refuses or loads, and what it loads builds — it never reads outside a buffer (1 200 such files ran
    clean under ASan + UBSan, tools/sanitize.sh; here the run must simply end with every file answered)."""
    import struct

    src = open(os.path.join(os.path.dirname(__file__), "golden", "cornellbox.glb"), "rb").read()
    json_len = struct.unpack_from("<I", src, 12)[0]
    rng = np.random.default_rng(5)
    answered = 0
    for it in range(80):
        b = bytearray(src)
        mode = it % 4
        if mode == 0:
            for _ in range(int(rng.integers(1, 4))):
                b[20 + int(rng.integers(json_len))] = int(rng.integers(48, 58))
        elif mode == 1:
            b[int(rng.integers(0, 20))] = int(rng.integers(256))
        elif mode == 2:
            for _ in range(int(rng.integers(1, 20))):
                b[20 + json_len + 8 + int(rng.integers(len(b) - 28 - json_len))] = int(rng.integers(256))
        else:
            b = b[: int(rng.integers(12, len(b)))]
        path = tmp_path / "fuzz.glb"
        path.write_bytes(bytes(b))
        sc = rta.Scene()
        try:
            sc.load_gltf(str(path))
            sc.build()
        except rta.RtError:
            pass
        answered += 1
    assert answered == 80
