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
