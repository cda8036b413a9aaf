"""CPU: the compressed four-wide occlusion records (rt_collapse_quads: what rt_scene_create builds under RT_TUNE_QUADS=1) —
their structure, and the claim the quad walk rests on, checked WITHOUT a GPU by emulating both walks in numpy float32:
  * every child box of every quad record, decoded as the kernel decodes it (fma(q, step, origin) in fp32), contains the
    exact box of that child in the pair tree; every leaf record carries the exact box its parent pair record holds;
  * every triangle slot is the target of exactly one leaf ref; the worst-case stack is what a depth-first walk can reach;
  * for random segments, "leaves whose EXACT box passes the slab test" is the same set whether it is found by the binary
    walk over the pair records (the reference's procedure, Object.h:322-343) or by the walk over the conservative quad
    records followed by the exact leaf-box test — the superset property and the monotonicity argument of rt_device.h."""
import numpy as np
import pytest

import scenes

F = np.float32


def _flat(rta, sc):
    return sc.array(rta.ARR_FLAT_PAIRS), sc.array(rta.ARR_FLAT_TRIS), sc.info()["root_ref"]


def _decode_quad(rec):
    """-> origin[3], step[3], lo[3][4], hi[3][4] (decoded planes, float32), refs[4] (int32)"""
    org = rec[0:3].view(F)
    e = [(int(rec[3]) >> (8 * a)) & 0xFF for a in range(3)]
    step = np.array([np.uint32(x << 23) for x in e], np.uint32).view(F)
    lo = np.zeros((3, 4), F)
    hi = np.zeros((3, 4), F)
    for a in range(3):
        for k in range(4):
            ql, qh = (int(rec[4 + a]) >> (8 * k)) & 0xFF, (int(rec[7 + a]) >> (8 * k)) & 0xFF
            # fma with ONE rounding: exact in float64 (8-bit x power of two + fp32), then rounded to fp32 once
            lo[a, k] = F(np.float64(ql) * np.float64(step[a]) + np.float64(org[a]))
            hi[a, k] = F(np.float64(qh) * np.float64(step[a]) + np.float64(org[a]))
    return org, step, lo, hi, rec[10:14].view(np.int32)


def _slab(bmin, bmax, o, inv, tcull):
    """Bounds::intersect (Math.h:755-770) in float32, finite rays: -> passes (incl. dist < tcull)"""
    t0 = ((bmin - o).astype(F) * inv).astype(F)
    t1 = ((bmax - o).astype(F) * inv).astype(F)
    tn = np.max(np.minimum(t0, t1))
    tf = np.min(np.maximum(t0, t1))
    return (not (tn > tf or tf < 0)) and tn < tcull


REF_NONE = -(2 ** 31)


def _collapse(rta, sc):
    pairs, tris, root = _flat(rta, sc)
    words, nq, qroot, need = rta.collapse_quads(pairs, root, tris)
    return pairs, tris, root, words, nq, qroot, need


@pytest.mark.parametrize("which", ["cornellbox_leaf1", "cornellbox_leaf2", "room_leaf4", "random20k"])
def test_quad_records_contain_the_exact_boxes_and_cover_every_triangle(rta, which):
    sc = {"cornellbox_leaf1": lambda: scenes.cornellbox(rta), "cornellbox_leaf2": lambda: scenes.cornellbox(rta),
          "room_leaf4": lambda: scenes.quad_room(rta), "random20k": lambda: scenes.random_triangles(rta, 20_000)}[which]()
    sc.build(**{"cornellbox_leaf2": dict(max_per_leaf=2), "room_leaf4": dict(max_per_leaf=4, strategy=0)}.get(which, {}))
    pairs, tris, root, words, nq, qroot, need = _collapse(rta, sc)
    assert 0 < nq < len(pairs) and qroot == 0 and words.shape == (nq + len(tris), 16)
    leaf_base = nq * 64
    # walk the pair tree and the quad tree together: each quad is a pair record with up to two of its children opened
    seen_slots = np.zeros(len(tris), int)
    todo = [(0, root)]  # (quad index, pair record it starts from)
    checked = 0
    deepest = {}
    order = []
    while todo:
        q, pr = todo.pop()
        order.append(q)
        org, step, lo, hi, refs = _decode_quad(words[q])
        # the exact children of this quad: the descendants of `pr` that the refs name, found by opening inner children
        exact = []  # (bmin, bmax, ref)
        P = pairs[pr]
        frontier = [(P["lmin"], P["lmax"], int(P["lref"])), (P["rmin"], P["rmax"], int(P["rref"]))]
        n_children = int(np.count_nonzero(refs != REF_NONE))
        assert 2 <= n_children <= 4 and (refs[:n_children] != REF_NONE).all()
        while len(frontier) < n_children:
            def area(mn, mx):
                e = (mx - mn).astype(np.float64)
                return e[0] * e[1] + e[1] * e[2] + e[2] * e[0]

            areas = [-1.0 if r < 0 else area(mn, mx) for mn, mx, r in frontier]  # (the builder opens the first of equal maxima)
            k = int(np.argmax(areas))
            assert areas[k] >= 0
            C = pairs[frontier[k][2]]
            frontier[k] = (C["lmin"], C["lmax"], int(C["lref"]))
            frontier.append((C["rmin"], C["rmax"], int(C["rref"])))
        assert len(frontier) == n_children  # a quad with fewer than four children has only leaves left to open
        if n_children < 4:
            assert all(r < 0 for _, _, r in frontier)
        for k, (mn, mx, r) in enumerate(frontier):
            for a in range(3):
                assert lo[a, k] <= mn[a] and hi[a, k] >= mx[a], (q, k, a)  # conservative, as decoded
                # ... and tight: within one grid step (plus fp32 rounding of the plane) of the exact box
                slack = np.float64(step[a]) * 1.001 + (abs(np.float64(mn[a])) + abs(np.float64(mx[a]))) * 2.4e-7
                assert np.float64(mn[a]) - np.float64(lo[a, k]) <= slack and np.float64(hi[a, k]) - np.float64(mx[a]) <= slack, (q, k, a)
            checked += 1
            ref = int(refs[k])
            if r < 0:
                slot = ~r
                assert ref == (leaf_base + slot * 64) | 1
                rec = words[nq + slot]
                assert np.array_equal(rec[0:3].view(F), mn) and np.array_equal(rec[3:6].view(F), mx)  # the leaf's exact box
                s = slot
                while True:  # the leaf's triangle records, as the walk steps through them
                    seen_slots[s] += 1
                    t = tris[s]
                    assert np.array_equal(words[nq + s][6:9].view(F), t["v0"]) and np.array_equal(words[nq + s][9:12].view(F), t["e1"])
                    assert np.array_equal(words[nq + s][12:15].view(F), t["e2"])
                    last = bool(t["tri_id"] & 0x80000000)
                    assert (words[nq + s][15] & 1) == int(last)
                    if last:
                        break
                    s += 1
            else:
                assert ref >= 0 and ref % 64 == 0 and ref // 64 > q  # children follow their parent (top of the tree first)
                todo.append((ref // 64, r))
                deepest.setdefault(q, []).append(ref // 64)
    assert (seen_slots == 1).all() and len(order) == nq and checked > nq
    # worst-case stack of a depth-first walk that always has (children - 1) siblings waiting
    depth_need = {}
    for q in sorted(order, reverse=True):
        n_children = int(np.count_nonzero(words[q][10:14].view(np.int32) != REF_NONE))
        depth_need[q] = n_children - 1 + max([depth_need[c] for c in deepest.get(q, [])], default=0)
    assert need == depth_need[0] <= 96


def _reached_binary(pairs, root, o, inv, tcull):
    """slots of the leaves the reference's walk enters (any order; every box test as Object.h:325-326 has it)"""
    out, stack = set(), [root]
    while stack:
        r = stack.pop()
        if r < 0:
            out.add(~r)
            continue
        P = pairs[r]
        if _slab(P["lmin"], P["lmax"], o, inv, tcull):
            stack.append(int(P["lref"]))
        if _slab(P["rmin"], P["rmax"], o, inv, tcull):
            stack.append(int(P["rref"]))
    return out


def _reached_quads(words, nq, o, inv, tcull, decoded):
    leaf_base = nq * 64
    cand, out, stack = 0, set(), [0]
    while stack:
        ref = stack.pop()
        if ref & 1:
            slot = ((ref & ~3) - leaf_base) // 64
            cand += 1
            rec = words[nq + slot]
            if _slab(rec[0:3].view(F), rec[3:6].view(F), o, inv, tcull):  # the exact leaf box decides
                out.add(slot)
            continue
        org, step, lo, hi, refs = decoded[ref // 64]
        for k in range(4):
            if refs[k] != REF_NONE and _slab(lo[:, k], hi[:, k], o, inv, tcull):
                stack.append(int(refs[k]))
    return out, cand


@pytest.mark.parametrize("per_leaf", [1, 2])
def test_quad_walk_reaches_the_reference_walks_leaves(rta, per_leaf):
    sc = scenes.cornellbox(rta)
    sc.build(max_per_leaf=per_leaf)
    pairs, tris, root, words, nq, qroot, need = _collapse(rta, sc)
    decoded = [_decode_quad(words[q]) for q in range(nq)]
    rays = scenes.random_rays(600, 17 + per_leaf, lo=(-1.7, 0.0, -1.7), hi=(1.7, 3.4, 1.7))
    rng = np.random.default_rng(5)
    tmax = (rng.random(len(rays)) * 3).astype(F)
    tmax[::5] = np.inf
    extra = total = 0
    for r, tm in zip(rays, tmax):
        o, d = r["o"].astype(F), r["d"].astype(F)
        inv = (F(1.0) / d).astype(F)
        tcull = F(tm) * F(1.0001) + F(0.000001) if np.isfinite(tm) else F(np.inf)  # the walks' cull value (rt_device.h)
        want = _reached_binary(pairs, root, o, inv, tcull)
        got, cand = _reached_quads(words, nq, o, inv, tcull, decoded)
        assert got == want
        extra += cand - len(got)
        total += len(want)
    assert total > 500 and extra < 0.25 * total  # the conservative boxes let few extra leaves through to their exact test


def test_collapse_refuses_what_the_quad_walk_cannot_prove(rta):
    """A caller's own records need not nest; NaN / infinite boxes, one-leaf trees and bad references are refused too —
    rt_scene_create then keeps such a scene on the pair walk (and says so under RT_TUNE_VERBOSE)."""
    sc = scenes.quad_room(rta)
    sc.build()
    pairs, tris, root = _flat(rta, sc)
    words, nq, _, _ = rta.collapse_quads(pairs, root, tris)
    assert nq > 0
    inner = [i for i in range(len(pairs)) if pairs[i]["lref"] >= 0][0]  # a record whose left child is a record itself
    child = int(pairs[inner]["lref"])
    bad = pairs.copy()
    bad[child]["lmax"][0] = bad[inner]["lmax"][0] + 1.0  # the grandchild now sticks out of its parent's box
    with pytest.raises(rta.RtError, match="pair walk"):
        rta.collapse_quads(bad, root, tris)
    for poison in (np.nan, np.inf):
        bad = pairs.copy()
        bad[child]["rmin"][1] = poison
        with pytest.raises(rta.RtError, match="pair walk"):
            rta.collapse_quads(bad, root, tris)
    bad = pairs.copy()
    bad[child]["lmin"][2], bad[child]["lmax"][2] = bad[child]["lmax"][2] + 1.0, bad[child]["lmin"][2]  # min > max
    with pytest.raises(rta.RtError, match="pair walk"):
        rta.collapse_quads(bad, root, tris)
    one = pairs[:1].copy()  # a tree whose root is a leaf has nothing to collapse
    one[0]["lref"], one[0]["rref"] = ~0, ~1
    with pytest.raises(rta.RtError, match="pair walk"):
        rta.collapse_quads(one, -1, tris)
    bad = pairs.copy()
    bad[0]["rref"] = len(pairs)
    with pytest.raises(rta.RtError, match="out of range"):
        rta.collapse_quads(bad, root, tris)
    bad = pairs.copy()
    bad[child]["lref"] = 0  # a cycle
    with pytest.raises(rta.RtError, match="do not form a tree"):
        rta.collapse_quads(bad, root, tris)


def test_collapse_of_a_degenerate_chain_reports_its_stack(rta):
    """A tree that is one long chain (every record: one leaf + one record) collapses to quads of three leaves + one quad:
    the worst-case stack grows by up to three per quad level and is reported; beyond 96 entries the tree is refused."""
    n = 60  # 60 chained records: within the pair walk's 64-level stack
    tris = np.zeros(n + 1, rta.FLAT_TRI)
    tris["tri_id"] = np.arange(n + 1, dtype=np.uint32) | np.uint32(0x80000000)
    tris["e1"][:, 0] = 1
    tris["e2"][:, 1] = 1
    pairs = np.zeros(n, rta.PAIR)
    for i in range(n):
        lo = np.float32(i)
        pairs[i]["lmin"], pairs[i]["lmax"] = (lo, 0, 0), (lo + 1, 1, 1)          # leaf i
        pairs[i]["rmin"], pairs[i]["rmax"] = (lo + 1, 0, 0), (np.float32(n + 1), 1, 1)  # the rest of the chain
        pairs[i]["lref"] = ~i
        pairs[i]["rref"] = i + 1 if i + 1 < n else ~n
    words, nq, qroot, need = rta.collapse_quads(pairs, 0, tris)
    assert nq == (n + 2) // 3 and need <= 96 and need >= 2 * nq - 2
    # every slot reachable once
    refs = words[:nq, 10:14].view(np.int32)
    leaf_refs = refs[(refs != REF_NONE) & ((refs & 1) == 1)]
    assert sorted(((leaf_refs & ~3) - nq * 64) // 64) == list(range(n + 1))
