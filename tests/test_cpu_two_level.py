"""CPU (-m "not gpu"): the legacy two-level design — Scene::instances under a TLAS, Instance over the
Geometry classes.  Host classes against the oracle's restatement (bounds, TLAS, flattening), and the oracle itself
against closed forms and against its own flat walk on the same geometry moved to world space."""
import numpy as np

import scenes


def _ray(o, d):
    import oracle_bindings as orc

    r = np.zeros(1, orc.RAY)
    r["o"], r["d"] = o, d
    return r


def test_instance_bounds_and_tlas_match_oracle(rta, orc):
    sc = scenes.legacy_two_level(rta, extra_spheres=40)
    sc.build()
    inst = sc.array(rta.ARR_INSTANCES)
    assert len(inst) == 10 + 3 + 40
    osc = orc.two_level_scene_from_host(sc, own_bvh=True)  # oracle's own bounds, mesh BVH and TLAS
    for i in range(len(inst)):
        if inst["geometry"][i] == rta.GEO_MESH:
            root = osc.meshes[osc.instances["mesh"][i]][2][0]
            obj = np.concatenate([root["bmin"], root["bmax"]])
        else:
            obj = orc.object_bounds(osc.instances[i])
        assert np.array_equal(sc.instance_bounds(i), orc.world_bounds(obj, inst["m"][i])), i
    assert sc.array(rta.ARR_TLAS_NODES).tobytes() == osc.tlas_nodes.tobytes()
    assert np.array_equal(sc.array(rta.ARR_TLAS_INDICES), osc.tlas_indices)
    assert sc.mesh_array(0, 2).tobytes() == osc.meshes[0][2].tobytes()
    assert np.array_equal(sc.mesh_array(0, 3), osc.meshes[0][3])


def test_flattened_two_level_tables(rta):
    sc = scenes.legacy_two_level(rta)
    sc.build()
    inst, pairs, tl = sc.array(rta.ARR_INSTANCES), sc.array(rta.ARR_FLAT_PAIRS), sc.array(rta.ARR_TLAS_PAIRS)
    tris = sc.array(rta.ARR_FLAT_TRIS)
    nodes, idx = sc.array(rta.ARR_TLAS_NODES), sc.array(rta.ARR_TLAS_INDICES)
    assert len(tl) == len(inst) - 1 == (len(nodes) - 1) // 2  # one instance per TLAS leaf
    leaves = sorted(~r for r in np.concatenate([tl["lref"], tl["rref"]]) if r < 0)
    assert leaves == list(range(len(inst)))
    mesh_inst = inst[inst["geometry"] == rta.GEO_MESH]
    assert len(mesh_inst) == 3 and len(set(mesh_inst["blas_root"])) == 1  # one BVH, placed three times
    assert len(tris) == len(sc.mesh_array(0, 1)) and len(pairs) == len(tris) - 1
    assert (tris["tri_id"][-1] & 0x80000000) != 0
    lights = sc.array(rta.ARR_LIGHTS)
    assert len(lights) == 1 and lights["kind"][0] == 0 and np.isclose(lights["area"][0], 0.3 * 1.3)
    rect = inst[inst["geometry"] == rta.GEO_RECTANGLE]
    assert np.isclose(rect["area"], [9.0, 0.39]).all() and (inst["area"][inst["geometry"] != rta.GEO_RECTANGLE] == 0).all()
    assert np.array_equal(idx[np.argsort(idx)], np.arange(len(inst)))


def test_analytic_primitives_known_answers(rta, orc):
    sc = rta.Scene()
    m = sc.add_material(rta.material(rta.MAT_MATTE, (0.5, 0.5, 0.5)))
    sc.add_primitive_instance(rta.GEO_SPHERE, (0.5,), m, scenes._trs((0, 0, -3)))
    sc.add_primitive_instance(rta.GEO_RECTANGLE, (0, 0, 1, 2.0, 1.0), m, scenes._trs((5, 0, -2)))
    sc.add_primitive_instance(rta.GEO_BOX, (1.0, 1.0, 1.0), m, scenes._trs((-5, 0, -4)))
    sc.add_primitive_instance(rta.GEO_CYLINDER, (0.5, 2.0), m, scenes._trs((0, 5, -3)))
    sc.add_primitive_instance(rta.GEO_DISK, (0, 0, 1, 1.0, 0.25), m, scenes._trs((0, -5, -2)))
    sc.build()
    osc = orc.two_level_scene_from_host(sc)
    cases = [  # origin, direction, expected (index, t, normal, front)
        ((0, 0, 0), (0, 0, -1), (0, 2.5, (0, 0, 1), 1)),      # sphere, front
        ((0, 0, -3), (0, 0, -1), (0, 0.5, (0, 0, 1), 0)),     # from the centre: second root, back face, normal flipped
        ((5.5, 0.25, 0), (0, 0, -1), (1, 2.0, (0, 0, 1), 1)),  # rectangle
        ((6.5, 0, 0), (0, 0, -1), (-1, np.inf, (0, 0, 0), 0)),  # beside it (u half length is 1)
        ((-5, 0.2, 0), (0, 0, -1), (2, 3.5, (0, 0, 1), 1)),    # box, +z face
        ((0, 5, 0), (0, 0, -1), (3, 2.5, (0, 0, 1), 1)),       # cylinder side
        ((0, 6.5, 0), (0, 0, -1), (-1, np.inf, (0, 0, 0), 0)),  # above its caps (open cylinder)
        ((0.5, -5, 0), (0, 0, -1), (4, 2.0, (0, 0, 1), 1)),    # disk annulus
        ((0.1, -5, 0), (0, 0, -1), (-1, np.inf, (0, 0, 0), 0)),  # through its hole
    ]
    for o, d, (index, t, n, front) in cases:
        hits, idx, normal, ff = orc.closest_hit2(osc, _ray(o, d))
        assert idx[0] == index, (o, idx[0])
        assert hits["t"][0] == np.float32(t) or np.isclose(hits["t"][0], t, rtol=1e-6), (o, hits["t"][0])
        assert np.allclose(normal[0], n, atol=1e-6) and ff[0] == front, (o, normal[0], ff[0])
    # Rectangle u, v (Object.h:116-117): u_dist / u with the half lengths
    hits, idx, _, _ = orc.closest_hit2(osc, _ray((5.5, 0.25, 0), (0, 0, -1)))
    assert np.isclose(abs(hits["u"][0]), 0.5) and np.isclose(abs(hits["v"][0]), 0.5)
    # trace_scene as a boolean: segment ends before / after the sphere
    occ = orc.any_hit2(osc, np.concatenate([_ray((0, 0, 0), (0, 0, -1))] * 2), [2.4, 2.6])
    assert list(occ) == [0, 1]


def test_mesh_instances_equal_flattened_world_mesh(rta, orc):
    """The same TriangleMesh placed by the two-level walk (ray moved to model space) and flattened into world space
    (Scene::add_instance) must see the same triangle at the same distance, up to the rounding of the transform."""
    v, t = scenes.uv_sphere(rta, 10, 20)
    v["pos"][:, 1] *= 1.6
    mats = scenes.instance_transforms(6, seed=3, lo=(-0.5, -0.5, -0.5), hi=(0.5, 0.5, 0.5), scale=(0.3, 0.5))
    two, flat = rta.Scene(), rta.Scene()
    m2, mf = two.add_material(rta.material(rta.MAT_MATTE, (0.5, 0.5, 0.5))), flat.add_material(rta.material(rta.MAT_MATTE, (0.5, 0.5, 0.5)))
    mesh = two.new_triangle_mesh(v, t)
    for m in mats:
        two.add_mesh_instance(mesh, m2, m)
        flat.add_instance(v, t, m, material=mf)
    two.build()
    flat.build()
    o2, of = orc.two_level_scene_from_host(two), orc.scene_from_host(flat)
    rays = scenes.random_rays(4000, 11, lo=(-1.2, -1.2, -1.2), hi=(1.2, 1.2, 1.2))
    h2, idx, _, _ = orc.closest_hit2(o2, rays)
    hf, _ = orc.closest_hit(of, rays)
    hit = idx >= 0
    assert 0.1 < hit.mean() < 0.95, hit.mean()
    agree = (hit == (hf["tri"] >= 0)) & (~hit | ((hf["tri"] == idx * len(t) + h2["tri"]) & np.isclose(h2["t"], hf["t"], rtol=1e-4, atol=1e-5)))
    assert agree.mean() > 0.995, agree.mean()  # silhouette rays may fall either side after the transform's rounding
    # occlusion (linear loop over instances) <=> closest hit within tmax
    tmax = np.where(hit, h2["t"] * np.where(np.arange(len(rays)) % 2 == 0, 1.01, 0.99), 1.0).astype(np.float32)
    occ = orc.any_hit2(o2, rays, tmax)
    expect = hit & (h2["t"] <= tmax)
    assert (occ.astype(bool) == expect).mean() > 0.999


def test_two_level_render_runs_on_oracle(rta, orc):
    sc = scenes.legacy_two_level(rta)
    sc.build()
    osc = orc.two_level_scene_from_host(sc)
    cam = scenes.legacy_camera(rta, 32, 24)
    img, c = orc.render(osc, cam, 32, 24, 4, 4, seed=2)
    assert np.isfinite(img).all() and img.mean() > 0.05
    assert c["paths"] == 32 * 24 * 4 and c["rays_shadow"] > 0.5 * c["rays_closest"]  # env + one light per shaded hit


def _build_example(tmp_path):
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "legacy_scene")
    pkg = os.path.join(root, "raytracing_amd")
    subprocess.run(["g++", "-std=c++17", "-O1", os.path.join(root, "examples", "legacy_scene.cpp"), "-I" + os.path.join(pkg, "host"), "-L" + pkg,
                    "-lrt_host", "-lrt_hip", "-Wl,-rpath," + pkg, "-o", exe], check=True)
    return exe


def test_cpp_scene_script_compiles_against_the_host_mirror(tmp_path):
    """a scene script in the reference's own shape (Instance(new Sphere(...), material, offset), ...)
    compiles and links against librt_host.so / librt_hip.so."""
    assert _build_example(tmp_path)


def test_set_material_after_instances_were_placed(rta):
    """Scene.set_material on a slot that Instances already reference (and a Rectangle light in both lists): the instances
    follow the new material; flatten finds it at the same index (no dangling pointer, no duplicate record)."""
    sc = rta.Scene()
    a = sc.add_material(rta.material(rta.MAT_MATTE, (0.1, 0.2, 0.3)))
    e = sc.add_material(rta.material(rta.MAT_EMISSIVE, (5, 5, 5)))
    eye = np.eye(4, dtype=np.float32).reshape(-1)
    sc.add_primitive_instance(rta.GEO_SPHERE, (0.5,), a, eye)
    sc.add_primitive_instance(rta.GEO_BOX, (1, 1, 1), a, eye)
    up = np.eye(4, dtype=np.float32)
    up[1, 3] = 3.0
    sc.add_primitive_instance(rta.GEO_RECTANGLE, (0, -1, 0, 0.5, 0.5), e, up.T.reshape(-1), light=True)
    sc.set_material(a, rta.material(rta.MAT_GGX, s=(0.3, 0.0)))
    sc.set_material(e, rta.material(rta.MAT_EMISSIVE, (9, 8, 7)))
    sc.build()
    mats, inst, lights = sc.array(rta.ARR_MATERIALS), sc.array(rta.ARR_INSTANCES), sc.array(rta.ARR_LIGHTS)
    assert len(mats) == 2 and mats["type"].tolist() == [rta.MAT_GGX, rta.MAT_EMISSIVE]
    assert inst["material"].tolist() == [a, a, e] and mats["s"][a][0] == np.float32(0.3)
    assert len(lights) == 1 and lights["material"][0] == e and mats["c0"][e].tolist() == [9, 8, 7]
    sc.set_material(a, rta.material(rta.MAT_MATTE, (0.5, 0.5, 0.5)))  # and again, after a flatten
    sc.build()
    assert sc.array(rta.ARR_MATERIALS)["type"].tolist() == [rta.MAT_MATTE, rta.MAT_EMISSIVE]
    sc.close()
