"""Host containers vs golden vectors generated from the reference's own utils/structs.py
(tests/golden/make_structs_golden.py).  This part of the oracle chain IS pinned."""
import json
import os

from image_stitcher_amd.utils.structs import ConstantLengthList, DynamicConnectivity, LazyList

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "structs_golden.json")))


def test_union_find_matches_reference():
    for c in G["union_find"]:
        dc = DynamicConnectivity(c["n"])
        for p, q in c["unions"]:
            dc.union(p, q)
        comps = dc.get_connected_components()
        assert comps == c["components"]
        assert sorted(comps, key=len, reverse=True) == c["components_sorted"]
        assert [dc.connected(p, q) for p, q in c["probes"]] == c["connected"]
        assert dc.parent == c["parent"] and dc.size == c["size"]


def test_ring_buffer_matches_reference():
    for c in G["ring"]:
        r = ConstantLengthList(c["max_length"])
        trace = iter(c["trace"])
        for step, op in enumerate(c["ops"]):
            t = next(trace)
            assert t[0] == step
            if op[0] == "append":
                r.append(op[1])
            elif op[0] == "pop":
                r.pop()
            elif op[0] == "set_last":
                r[-1] = op[1]
            else:
                assert list(iter(r)) == t[1]
                continue
            assert r.list == t[1] and list(reversed(r[:-1])) == t[2] and len(r) == t[3]


def test_lazy_list():
    calls = []
    ll = LazyList(lambda i: calls.append(i) or i * i, 4)
    assert len(ll) == 4 and ll[3] == 9 and calls == [3]
    try:
        ll[4]
        assert False
    except IndexError:
        pass
    ll.length -= 1
    assert len(ll) == 3


def test_components_from_counts_replays_the_reference_loop():
    """pairs.components_from_counts == the reference's double loop over ordered pairs (stitcher.py:285-300),
    including the union-find's internal state, on asymmetric random counts."""
    import numpy as np
    from image_stitcher_amd import pairs as prs
    from image_stitcher_amd.utils.structs import DynamicConnectivity
    rng = np.random.default_rng(0)
    for n, p_edge, cr in ((40, 0.03, None), (64, 0.01, None), (50, 0.2, 3)):
        M = (rng.random((n, n)) < p_edge).astype(np.int32) * 9        # M[i, j] = len(good) of the ordered pair i -> j
        ref = DynamicConnectivity(n + 2)
        for i in range(n):
            for j in range(n):
                if j == i or (cr is not None and abs(i - j) > cr):
                    continue
                if M[i, j] >= 5:
                    ref.union(i, j)
        exp = sorted(ref.get_connected_components(), key=len, reverse=True)
        pr = prs.unordered_pairs(n, cr)
        comps, dc = prs.components_from_counts(n, pr, M[pr[:, 0], pr[:, 1]], M[pr[:, 1], pr[:, 0]], 5, n_union_find=n + 2)
        assert comps == exp and dc.parent == ref.parent and dc.size == ref.size
