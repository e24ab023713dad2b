"""Offline model of the path kernel's wave vote (design exploration, not product code).

Takes the per-path event strings of the oracle (ora_trace_path_events: S box test, L sphere
test, H scatter, E end of path) for the 64 pixels of some 8x8 tiles and replays them through
scheduling policies, counting issue cost (wave-steps x instructions per step) against the
useful lane-steps.  Uses oracle/, hence it lives under tests/ (the only tree allowed to); never imported by the package."""
import ctypes as C
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402
import path_tracer_amd as pt  # noqa: E402
import bench  # noqa: E402

COST = {"S": 25, "L": 45, "H": 260, "E": 120, "vote": 14}


def events(workload="final", tiles=24, n_samples=8, seed=1, keep_x=False):
    kind, w, h, S, _ = bench.WORKLOADS[workload]
    rt = bench.build_raytracer(pt, kind, w, h, 16)
    osc = oracle_lib.OracleScene(rt.packed)
    L = osc.L
    L.ora_trace_path_events.restype = C.c_int32
    L.ora_trace_path_events.argtypes = [C.c_void_p, C.POINTER(oracle_lib.OraCamera), C.c_int32, C.c_uint64, C.c_int32,
                                        C.c_int32, C.c_int32, C.c_char_p, C.c_int32]
    cam = oracle_lib.OraCamera.from_buffer_copy(bytes(rt.camera.pod))
    rnd = random.Random(seed)
    buf = C.create_string_buffer(1 << 16)
    out = []
    for _ in range(tiles):
        tx, ty = rnd.randrange(w // 8), rnd.randrange(h // 8)
        s0 = rnd.randrange(0, 256 - n_samples)
        lanes = []
        for k in range(64):
            px, py = tx * 8 + k % 8, ty * 8 + k // 8
            seq = []
            for s in range(s0, s0 + n_samples):
                n = L.ora_trace_path_events(osc.handle, C.byref(cam), 16, bench.RENDER_SEED, px, py, s, buf, 1 << 16)
                seq.append(buf.raw[:n].decode())
            if not keep_x:
                seq = [x.replace("X", "L") for x in seq]
            lanes.append("".join(seq))
        out.append(lanes)
    return out


def sim_vote(lanes, w=(1, 2, 1, 1), keep8=4, cost=COST):
    """The kernel's policy: plurality vote with weights, SPLIT runs while >= keep/8 remain."""
    pos = [0] * 64
    n = [len(x) for x in lanes]
    issue = 0
    useful = 0
    steps = {"S": 0, "L": 0, "H": 0, "E": 0}
    act = {"S": 0, "L": 0, "H": 0, "E": 0}
    while True:
        cnt = {"S": 0, "L": 0, "H": 0, "E": 0}
        for i in range(64):
            if pos[i] < n[i]:
                cnt[lanes[i][pos[i]]] += 1
        top, pick = cnt["S"] * w[0], "S"
        for k, ww in (("L", w[1]), ("H", w[2]), ("E", w[3])):
            if cnt[k] * ww > top:
                top, pick = cnt[k] * ww, k
        if top == 0:
            break
        issue += cost["vote"]
        if pick == "S":
            keep = (cnt["S"] * keep8 + 7) >> 3
            while True:
                c = 0
                for i in range(64):
                    if pos[i] < n[i] and lanes[i][pos[i]] == "S":
                        pos[i] += 1
                        c += 1
                issue += cost["S"] + 4
                useful += c * cost["S"]
                steps["S"] += 1
                act["S"] += c
                c = sum(1 for i in range(64) if pos[i] < n[i] and lanes[i][pos[i]] == "S")
                if not (c >= keep and keep > 0):
                    break
        else:
            c = 0
            for i in range(64):
                if pos[i] < n[i] and lanes[i][pos[i]] == pick:
                    pos[i] += 1
                    c += 1
            issue += cost[pick]
            useful += c * cost[pick]
            steps[pick] += 1
            act[pick] += c
    return issue, useful, steps, act


def sim_pend(lanes, wS=1.0, wP=0.5, wB=2.0, wH=1.0, wE=1.0, keep8=4, cost=COST, pass_cost=4):
    """Deferred leaves: a lane that reaches a leaf parks it (one slot) and walks on with the
    stale closest hit; the LEAF step tests the parked leaf and rewinds the lane to just after
    it when the test hits.  Blocked: second leaf, or end of walk, with the slot full."""
    pos = [0] * 64
    n = [len(x) for x in lanes]
    pend = [None] * 64
    issue = useful = 0
    steps = {"S": 0, "L": 0, "H": 0, "E": 0}
    act = dict(steps)
    redo = 0

    def cur(i):
        return lanes[i][pos[i]] if pos[i] < n[i] else None

    def s_ok(i):
        c = cur(i)
        return c == "S" or (c in ("L", "X") and pend[i] is None)

    while True:
        cS = sum(1 for i in range(64) if s_ok(i))
        cP = sum(1 for i in range(64) if pend[i] is not None)
        cB = sum(1 for i in range(64) if pend[i] is not None and not s_ok(i))
        cH = sum(1 for i in range(64) if pend[i] is None and cur(i) == "H")
        cE = sum(1 for i in range(64) if pend[i] is None and cur(i) == "E")
        opts = [(cS * wS, "S"), ((cP - cB) * wP + cB * wB if cP else 0, "L"), (cH * wH, "H"), (cE * wE, "E")]
        top, pick = max(opts, key=lambda t: t[0])
        if top == 0:
            break
        issue += cost["vote"]
        if pick == "S":
            keep = (cS * keep8 + 7) >> 3
            while True:
                c = 0
                for i in range(64):
                    if s_ok(i):
                        if cur(i) != "S":
                            pend[i] = pos[i]
                        pos[i] += 1
                        c += 1
                issue += cost["S"] + 4 + pass_cost
                steps["S"] += 1
                act["S"] += c
                c = sum(1 for i in range(64) if s_ok(i))
                if not (c >= keep and keep > 0):
                    break
        elif pick == "L":
            for i in range(64):
                if pend[i] is not None:
                    if lanes[i][pend[i]] == "X":
                        redo += pos[i] - pend[i] - 1
                        pos[i] = pend[i] + 1
                    pend[i] = None
            issue += cost["L"] + 4
            steps["L"] += 1
            act["L"] += cP
        else:
            c = 0
            for i in range(64):
                if pend[i] is None and cur(i) == pick:
                    pos[i] += 1
                    c += 1
            issue += cost[pick]
            steps[pick] += 1
            act[pick] += c
    useful = sum(cost[{"X": "L"}.get(ch, ch)] for x in lanes for ch in x)
    return issue, useful, steps, act


def report(name, tiles, paths=None, sim=None, **kw):
    sim_fn = sim or sim_vote
    issue = useful = 0
    steps = {"S": 0, "L": 0, "H": 0, "E": 0}
    act = dict(steps)
    for lanes in tiles:
        i, u, s, a = sim_fn(lanes, **kw)
        issue += i
        useful += u
        for k in steps:
            steps[k] += s[k]
            act[k] += a[k]
    paths = paths or sum(sum(x.count("E") for x in lanes) for lanes in tiles)
    print("%-28s issue/path %7.0f  util %.3f  lanes/step %s" % (
        name, issue / paths, useful / (issue * 64),
        " ".join("%s %.1f" % (k, act[k] / max(1, steps[k])) for k in "SLHE")))
    return issue / paths


if __name__ == "__main__":
    evx = events(tiles=12, n_samples=32, keep_x=True)
    ev = [[x.replace("X", "L") for x in lanes] for lanes in evx]
    paths = sum(sum(x.count("E") for x in lanes) for lanes in ev)
    tot = {k: sum(sum(x.count(k) for x in lanes) for lanes in ev) / paths for k in "SLHE"}
    print("events per path:", tot)
    report("kernel (w 1,2,1,1 keep 4)", ev)
    report("w 1,1,1,1 keep 4", ev, w=(1, 1, 1, 1))
    report("keep 8", ev, keep8=8)
    report("keep 2", ev, keep8=2)
    for kw in (dict(), dict(wP=0.25), dict(wP=0.75), dict(wP=1.0), dict(wB=1.0), dict(wB=4.0, wP=0.25), dict(keep8=2),
               dict(keep8=6), dict(wP=0.5, wB=1.0, wH=0.5)):
        report("pend " + " ".join("%s=%s" % t for t in kw.items()), evx, sim=sim_pend, **kw)
    # traversal-only (H/E free): bound on what separating shading would give for the walk
    free = dict(COST, H=0, E=0)
    trav = [[x.replace("H", "").replace("E", "") for x in lanes] for lanes in ev]
    report("walk only, keep 4", trav, paths=paths, cost=free)
    report("walk only, w 1,1 keep 4", trav, paths=paths, cost=free, w=(1, 1, 1, 1))


def sim_shade(lanes, w=(1, 2, 1), keep8=4, cost=COST):
    """H and E voted as ONE class; the step runs both bodies (each under its own mask)."""
    pos = [0] * 64
    n = [len(x) for x in lanes]
    issue = useful = 0
    steps = {"S": 0, "L": 0, "H": 0, "E": 0}
    act = dict(steps)
    while True:
        cnt = {"S": 0, "L": 0, "H": 0, "E": 0}
        for i in range(64):
            if pos[i] < n[i]:
                cnt[lanes[i][pos[i]]] += 1
        opts = [(cnt["S"] * w[0], "S"), (cnt["L"] * w[1], "L"), ((cnt["H"] + cnt["E"]) * w[2], "Z")]
        top, pick = max(opts, key=lambda t: t[0])
        if top == 0:
            break
        issue += cost["vote"] - 3
        if pick == "S":
            keep = (cnt["S"] * keep8 + 7) >> 3
            while True:
                c = 0
                for i in range(64):
                    if pos[i] < n[i] and lanes[i][pos[i]] == "S":
                        pos[i] += 1
                        c += 1
                issue += cost["S"] + 4
                useful += c * cost["S"]
                steps["S"] += 1
                act["S"] += c
                c = sum(1 for i in range(64) if pos[i] < n[i] and lanes[i][pos[i]] == "S")
                if not (c >= keep and keep > 0):
                    break
        elif pick == "L":
            for i in range(64):
                if pos[i] < n[i] and lanes[i][pos[i]] == "L":
                    pos[i] += 1
            issue += cost["L"]
            useful += cnt["L"] * cost["L"]
            steps["L"] += 1
            act["L"] += cnt["L"]
        else:
            for k in "HE":
                if cnt[k]:
                    issue += cost[k]
                    useful += cnt[k] * cost[k]
                    steps[k] += 1
                    act[k] += cnt[k]
            for i in range(64):
                if pos[i] < n[i] and lanes[i][pos[i]] in "HE":
                    pos[i] += 1
    return issue, useful, steps, act


def sim_classes(lanes, weights, cost, run=("S",), keep8=4, vote_cost=10, combine=None):
    """Generic plurality vote over the event alphabet of `lanes` (one char per step); classes in
    `run` execute as runs (while >= keep/8 of the starters remain).  combine: dict class ->
    (classes executed together with it, e.g. {"Z": "HE"})."""
    pos = [0] * 64
    n = [len(x) for x in lanes]
    issue = 0
    steps = {k: 0 for k in weights}
    act = {k: 0 for k in weights}
    while True:
        cnt = {k: 0 for k in weights}
        for i in range(64):
            if pos[i] < n[i]:
                cnt[lanes[i][pos[i]]] += 1
        top, pick = 0, None
        for k in weights:
            v = cnt[k] * weights[k]
            if v > top:
                top, pick = v, k
        if pick is None:
            break
        issue += vote_cost
        if pick in run:
            keep = (cnt[pick] * keep8 + 7) >> 3
            while True:
                c = 0
                for i in range(64):
                    if pos[i] < n[i] and lanes[i][pos[i]] == pick:
                        pos[i] += 1
                        c += 1
                issue += cost[pick]
                steps[pick] += 1
                act[pick] += c
                c = sum(1 for i in range(64) if pos[i] < n[i] and lanes[i][pos[i]] == pick)
                if not (c >= keep and keep > 0):
                    break
        else:
            c = 0
            for i in range(64):
                if pos[i] < n[i] and lanes[i][pos[i]] == pick:
                    pos[i] += 1
                    c += 1
            issue += cost[pick]
            steps[pick] += 1
            act[pick] += c
    return issue, 0, steps, act


def report2(name, tiles, paths, **kw):
    issue = 0
    steps, act = {}, {}
    for lanes in tiles:
        i, _, s, a = sim_classes(lanes, **kw)
        issue += i
        for k in s:
            steps[k] = steps.get(k, 0) + s[k]
            act[k] = act.get(k, 0) + a[k]
    print("%-34s issue/path %6.1f  lanes/step %s" % (name, issue / paths, " ".join("%s %.1f" % (k, act[k] / max(1, steps[k])) for k in steps)))
