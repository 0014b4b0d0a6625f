"""Developer probe: extend time for different numberings of the node-pair records (uvrt_set_record_perm).
Records 2j and 2j+1 share one 128-byte L2/L1 line, the first 127 records of the numbering are served from
LDS.  Layouts: bfs (default), hot (hottest 127 first, rest BFS), dfs (top 127 BFS, rest depth-first
preorder: a left child follows its parent), pair (top 127 hottest, rest greedily packed as parent + its
most visited inner child per line)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

pkg = g.load_package(); orc = g.load_oracle()
glb = os.path.join(ROOT, "tests/golden/testroomopt.glb")
route = orc.load_route(os.path.join(ROOT, "tests/golden/lange_route.xml"))
s = orc.Scene(glb)
comp = orc.Computation(s, route["lamps"], 1 << 16, route["lightHeight"], route["lightLength"], route["lightIntensity"])
lp = comp.lamp_world_pos(route["lamps"][int(os.environ.get("LAMP", 0))])
n = 2073600
sample = int(os.environ.get("SAMPLE", 262144))
rays, _ = orc.generate(0, sample, lp, route["lightLength"], 12345)
hist = orc.extend_visit_hist(s.tris, rays, s.nodes, s.triIdx)
nodes = s.nodes
q = [0] if nodes[0]["triCount"] == 0 else []
pos = {0: 0}
kids = []
i = 0
while i < len(q):
    l = int(nodes[q[i]]["leftFirst"]); i += 1
    ks = []
    for k in (0, 1):
        if nodes[l + k]["triCount"] == 0:
            pos[l + k] = len(q); ks.append(len(q)); q.append(l + k)
    kids.append(ks)
P = len(q)
visits = hist[np.array(q)].astype(np.int64)
TOP = 127


def perm_from_order(order):
    perm = np.empty(P, dtype=np.uint32)
    perm[np.asarray(order, dtype=np.int64)] = np.arange(P, dtype=np.uint32)
    return perm


def layout_hot():
    return perm_from_order(np.argsort(-visits, kind="stable"))


def layout_dfs(top):
    top = list(top)
    placed = np.zeros(P, bool); placed[top] = True
    order = top[:]
    stack = [0]
    while stack:
        x = stack.pop()
        if not placed[x]:
            placed[x] = True; order.append(x)
        for k in reversed(sorted(kids[x], key=lambda c: -visits[c])):    # hotter child first
            stack.append(k)
    return perm_from_order(order)


def layout_pair(top):
    top = list(top)
    if len(top) % 2 == 0:
        pass
    placed = np.zeros(P, bool); placed[top] = True
    order = top[:]
    if len(order) % 2:            # keep line alignment: lines start at even indices
        rest_first = None
    # hottest unplaced node first, together with its hottest unplaced inner child
    for x in np.argsort(-visits, kind="stable"):
        if placed[x]:
            continue
        if len(order) % 2 == 1:
            # fill the odd slot with the hottest unplaced child of the previous record, else with x
            prev = order[-1]
            c = [k for k in kids[prev] if not placed[k]]
            if c:
                k = max(c, key=lambda k: visits[k]); placed[k] = True; order.append(k)
                if placed[x]:
                    continue
            else:
                placed[x] = True; order.append(x); continue
        placed[x] = True; order.append(x)
        c = [k for k in kids[x] if not placed[k]]
        if c:
            k = max(c, key=lambda k: visits[k]); placed[k] = True; order.append(k)
    return perm_from_order(order)


bfs_top = list(range(min(TOP, P)))
hot_top = list(np.argsort(-visits, kind="stable")[:TOP])
layouts = [("bfs", None), ("hot", layout_hot()), ("dfs", layout_dfs(bfs_top)), ("hot+dfs", layout_dfs(hot_top)),
           ("hot+pair", layout_pair(hot_top))]
for name, pm in layouts:
    if pm is not None:
        assert np.array_equal(np.sort(pm), np.arange(P, dtype=np.uint32)), name
        inv = np.empty(P, dtype=np.int64); inv[pm] = np.arange(P)
        same_line = sum(visits[k] for x in range(P) for k in kids[x] if pm[x] >= TOP and pm[k] >= TOP and (pm[x] >> 1) == (pm[k] >> 1))
        print("%-9s: LDS coverage %.3f, descents into the parent's own 128-B line %.3f of all visits"
              % (name, visits[inv[:TOP]].sum() / visits.sum(), same_line / visits.sum()))

c = pkg.capi.Ctx(0)
c.set_scene(s.tris, s.nodes, s.triIdx)
c.resize_rays(n)
if os.environ.get("PIPELINE", "0") == "0":
    c.set_pipeline(False)
full, _ = orc.generate(0, n, lp, route["lightLength"], 0)
ref = np.zeros(s.T, dtype=np.int32)
orc.extend(ref, s.tris, full, s.nodes, s.triIdx)
best = {}
for rnd in range(4):
    for name, pm in layouts:
        c.set_record_perm(pm)
        c.set_timing(True); c.reset(False); c.seed = 0
        c.generate(lp, route["lightLength"], 0, n); c.extend(n); c.sync()
        ok = np.array_equal(c.read_counts(), ref)
        c.extend_time_ms()
        for _ in range(6):
            c.seed = 0
            c.generate(lp, route["lightLength"], 0, n); c.extend(n); c.accumulate(60.0)
        c.sync()
        ms, k = c.extend_time_ms()
        best[name] = min(best.get(name, 1e9), ms / k)
        if rnd == 0:
            print(name, "counts", "OK" if ok else "MISMATCH", flush=True)
for name, ms in best.items():
    print("%-9s: extend %.3f ms (%.1f Mray/s)" % (name, ms, n / ms / 1e3))
