"""Generates tests/golden/structs_golden.json by importing the REFERENCE's own
image_stitcher/utils/structs.py (no cv2 dependency) from /root/reference.  Run in the build container
only; the JSON holds inputs and expected outputs (data), never reference source."""
import importlib.util
import json
import os
import random

REF = "/root/reference/image_stitcher/utils/structs.py"
spec = importlib.util.spec_from_file_location("ref_structs", REF)
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

random.seed(20250222)
cases = []
for n, m in [(1, 0), (2, 1), (5, 3), (8, 20), (16, 10), (40, 25), (64, 200), (100, 60), (512, 700)]:
    unions = [(random.randrange(n), random.randrange(n)) for _ in range(m)]
    dc = ref.DynamicConnectivity(n)
    for p, q in unions:
        dc.union(p, q)
    comps = dc.get_connected_components()
    comps_sorted = sorted(comps, key=len, reverse=True)          # stitcher.py:300
    probes = [(random.randrange(n), random.randrange(n)) for _ in range(10)]
    conn = [bool(dc.connected(p, q)) for p, q in probes]
    cases.append({"n": n, "unions": unions, "components": comps, "components_sorted": comps_sorted,
                  "probes": probes, "connected": conn, "parent": dc.parent, "size": dc.size})

ring_cases = []
for max_len in (1, 2, 3, 5):
    ops, trace = [], []
    r = ref.ConstantLengthList(max_len)
    for step in range(40):
        c = random.random()
        if c < 0.55 or len(r) == 0:
            v = random.randrange(1000)
            r.append(v)
            ops.append(["append", v])
        elif c < 0.7:
            r.pop()
            ops.append(["pop"])
        elif c < 0.85:
            v = random.randrange(1000)
            r[-1] = v
            ops.append(["set_last", v])
        else:
            ops.append(["iter"])
            trace.append([step, list(iter(r))])
            continue
        trace.append([step, list(r.list), list(reversed(r[:-1])), len(r)])
    ring_cases.append({"max_length": max_len, "ops": ops, "trace": trace})

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "structs_golden.json")
with open(out, "w") as f:
    json.dump({"union_find": cases, "ring": ring_cases}, f)
print("wrote", out, os.path.getsize(out), "bytes")
