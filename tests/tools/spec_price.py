"""TEST INFRASTRUCTURE (it drives the CPU oracle, hence it lives under tests/): prices speculative leaf evaluation in the tail of a self-play batch BEFORE anything is built (round-4 review, item 3).

At <= 32 live games one network evaluation costs the same ~95 us whether its launch carries 1 or 32 boards, and a search
iteration is one such launch + ~8 us of tree work.  Every leaf a search selects exists (state, frozen dice) from the moment its
parent was expanded, and below 129 boards a row's network output is a pure function of its state -- so a launch's free rows can
carry unexpanded nodes the search is LIKELY to select soon, and an iteration whose selected leaves were all evaluated earlier
needs no launch at all.  Results stay bit-identical (the search itself is untouched; only when a row is computed changes).

This script replays the headline workload (1024 games x iterations 100, or what --games / --iterations say) on the CPU oracle
(the reference's lockstep search, oracle/diee_oracle.c), and inside the evaluator callback -- i.e. between the selections of an
iteration and its expansions, where the engine would decide -- simulates the row cache under several candidate policies.  Per
live-game count it reports the iterations, the launches each policy still needs, and the time that buys under a simple model
(launch + tree kernel = --launch-us, a skipped iteration = --skip-us).  Evaluator: the engine's network on a GPU box
(--eval engine; the priors and values of the real random-init net shape the trees), or the oracle's hash evaluator (--eval hash).

    python tests/tools/spec_price.py --eval engine --out profiles/r05a_spec_price.json
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc                                    # noqa: E402  (test infrastructure: this is a pricing tool)

NODE = np.dtype({"names": ["state", "parent", "first_child", "n_children", "visits", "value", "policy", "action", "drained"],
                 "formats": [("u1", 32), "i4", "i4", "i4", "f4", "f4", "f4", "i4", "u1"],
                 "offsets": [0, 32, 36, 40, 44, 48, 52, 56, 60], "itemsize": 64})
assert C.sizeof(orc.Node) == 64


def store_view(L, run):
    st = C.cast(L.or_mcts_store(run), C.POINTER(orc.Store)).contents
    if st.n == 0:
        return np.zeros(0, NODE)
    return np.ctypeslib.as_array(C.cast(st.nodes, C.POINTER(C.c_uint8)), shape=(st.n * 64,)).view(NODE)


def terminal(state_bytes):
    return state_bytes[26] == 15 or state_bytes[27] == 15            # off[0] / off[1], backgammon_logic.rs:527-534


# --------------------------------------------------------------------------- candidate policies
def ucb_scores(vis, val, pol, pv, c):
    q = np.where(vis == 0, 0.0, val / np.where(vis == 0, 1.0, vis)).astype(np.float32)
    return q + (np.float32(c) * (np.sqrt(np.float32(pv)) / (vis + 1.0))) * pol


def pick_child(vis, val, pol, pv, c):
    s = ucb_scores(vis, val, pol, pv, c)
    best = 0
    for j in range(1, len(s)):                                       # max_by: the last of equal maxima, NaN = Equal
        if not (s[best] > s[j]):
            best = j
    return best


class Sim:
    """one policy's row cache over one move-step"""
    def __init__(self, name, budget, kind, **kw):
        self.name, self.budget, self.kind, self.kw = name, budget, kind, kw
        self.reset(0)
        self.by_m = {}

    def reset(self, n_nodes_hint):
        self.cached = np.zeros(max(n_nodes_hint, 1024), bool)
        self.cval = np.zeros(max(n_nodes_hint, 1024), np.float32)

    def grow(self, n):
        if n > len(self.cached):
            k = max(n, 2 * len(self.cached))
            self.cached = np.concatenate([self.cached, np.zeros(k - len(self.cached), bool)])
            self.cval = np.concatenate([self.cval, np.zeros(k - len(self.cval), np.float32)])

    def account(self, m, launched, rows_spec, per_game_miss):
        d = self.by_m.setdefault(m, {"iterations": 0, "launches": 0, "spec_rows": 0, "game_misses": 0, "game_selections": 0})
        d["iterations"] += 1; d["launches"] += int(launched); d["spec_rows"] += rows_spec
        d["game_misses"] += int(sum(per_game_miss)); d["game_selections"] += len(per_game_miss)


def frontier_ok(nodes, idx, sim):
    nd = nodes[idx]
    return nd["n_children"] == 0 and not nd["drained"] and not sim.cached[idx] and not terminal(nd["state"])


def candidates_top_prior(nodes, owner_nodes, sim, want, per_parent):
    """expanded nodes by visit count (descending); of each, the unvisited uncached children by prior (descending)"""
    exp = [i for i in owner_nodes if nodes[i]["n_children"] > 0]
    exp.sort(key=lambda i: -nodes[i]["visits"])
    out = []
    for rank in range(per_parent):
        for p in exp:
            fc, k = nodes[p]["first_child"], nodes[p]["n_children"]
            ch = [fc + j for j in range(k) if frontier_ok(nodes, fc + j, sim) and nodes[fc + j]["visits"] == 0 and (fc + j) not in out]
            ch.sort(key=lambda i: -nodes[i]["policy"])
            if ch:
                out.append(ch[0])
                if len(out) >= want:
                    return out
    return out


def candidates_rollout(nodes, root, sim, want, c, vhat, root_player, demanded):
    """virtual search from the current tree: unknown evaluations count as `vhat` (0 or the parent's mean), selected frontier
    nodes that are not cached become candidates; they stay childless (their children do not exist yet)"""
    vis = {}; val = {}

    def gv(i): return vis.get(i, float(nodes[i]["visits"]))
    def gw(i): return val.get(i, float(nodes[i]["value"]))
    out = []
    steps = 0
    while len(out) < want and steps < 6 * want + 8:
        steps += 1
        idx = root; path = [root]
        while nodes[idx]["n_children"] > 0:
            fc, k = int(nodes[idx]["first_child"]), int(nodes[idx]["n_children"])
            v = np.array([gv(fc + j) for j in range(k)], np.float32); w = np.array([gw(fc + j) for j in range(k)], np.float32)
            idx = fc + pick_child(v, w, nodes["policy"][fc:fc + k], np.float32(gv(idx)), c)
            path.append(idx)
        st = nodes[idx]["state"]
        if terminal(st):
            winner = -1 if st[26] == 15 else 1
            x = 1.0 if winner == root_player else -1.0
        elif sim.cached[idx]:
            x = float(sim.cval[idx])
        else:
            par = path[-2] if len(path) > 1 else idx
            x = 0.0 if vhat == "zero" else (gw(par) / gv(par) if gv(par) > 0 else 0.0)
            if idx not in out and idx not in demanded and not nodes[idx]["drained"]:
                out.append(idx)
        for p in path:
            vis[p] = gv(p) + 1.0; val[p] = gw(p) + x
    return out


# --------------------------------------------------------------------------- the replay
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=1024)
    ap.add_argument("--iterations", type=int, default=100)
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0xD1EE0001)
    ap.add_argument("--eval", choices=("hash", "engine"), default="hash")
    ap.add_argument("--max-live", type=int, default=16, help="simulate the cache only at move-steps with at most this many live games")
    ap.add_argument("--budget", type=int, default=32, help="rows of a launch (demanded + speculative)")
    ap.add_argument("--launch-us", type=float, default=104.0, help="one search iteration with a launch (tower + tree kernel, profiles/r04e_tail_gaps.txt)")
    ap.add_argument("--skip-us", type=float, default=14.0, help="one search iteration without (tree kernel + an early-exit launch)")
    ap.add_argument("--batch-s", type=float, default=10.1, help="wall clock of the whole batch (BENCH_r04: 10 111.9 ms)")
    ap.add_argument("--max-steps", type=int, default=0)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()

    orc.build(); L = orc.lib()
    L.or_mcts_last.restype = C.c_void_p
    for f in (L.or_mcts_sel, L.or_mcts_fresh, L.or_mcts_store):
        f.argtypes = [C.c_void_p]; f.restype = C.c_void_p
    L.or_mcts_phase.argtypes = [C.c_void_p]; L.or_mcts_phase.restype = C.c_int
    L.or_hash_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]; L.or_hash_eval.restype = None
    g = orc.game(1)
    A = 1352
    cfgc = 2.0

    eng = None
    if args.eval == "engine":
        import diee_amd
        eng = diee_amd.Engine(0); eng.load_weights(diee_amd.random_weights(0))

    def evaluate(states_u8):
        """[n][32] u8 -> policy [n][A], value [n]"""
        n = len(states_u8)
        if eng is not None:
            pol, val = eng.forward_t(np.ascontiguousarray(states_u8).view(orc.BG_STATE).reshape(-1))
            return np.ascontiguousarray(pol, np.float32), np.ascontiguousarray(val, np.float32)
        pol = np.empty((n, A), np.float32); val = np.empty(n, np.float32)
        st = np.ascontiguousarray(states_u8)
        L.or_hash_eval(g, st.ctypes.data, n, pol.ctypes.data, val.ctypes.data)
        return pol, val

    B = args.budget
    sims = [Sim("none (today)", B, "none"),
            Sim("root children + top-prior x2", B, "top", per_parent=2),
            Sim("root children + top-prior x3", B, "top", per_parent=3),
            Sim("root children + virtual rollout, unknown = 0", B, "roll", vhat="zero"),
            Sim("root children + virtual rollout, unknown = parent mean", B, "roll", vhat="parent")]
    live_hist = {}
    state = {"step": -1, "owner": None, "n_seen": 0, "t_sim": 0.0}

    def cb(ctx, sp, n, pp, vp):
        states = np.ctypeslib.as_array(C.cast(sp, C.POINTER(C.c_uint8)), shape=(n, 32))
        pol, val = evaluate(states)
        C.memmove(pp, pol.ctypes.data, pol.nbytes); C.memmove(vp, val.ctypes.data, val.nbytes)
        run = L.or_mcts_last()
        phase = L.or_mcts_phase(run)
        if phase == 0:
            state["step"] += 1
            live_hist[n] = live_hist.get(n, 0) + 1
        if n > args.max_live:
            return
        t0 = time.time()
        nodes = store_view(L, run)
        if phase == 0:
            for s in sims:
                s.reset(4096); s.account(n, True, 0, [])             # the root evaluation: always a launch
            state["owner"] = np.zeros(0, np.int32); state["n_seen"] = 0
            state["t_sim"] += time.time() - t0
            return
        # owner (game) of every node created since the last callback
        nn = len(nodes)
        if nn > state["n_seen"]:
            own = np.concatenate([state["owner"], np.zeros(nn - state["n_seen"], np.int32)])
            for i in range(state["n_seen"], nn):
                p = nodes[i]["parent"]
                own[i] = i if p < 0 else own[p]
            state["owner"] = own; state["n_seen"] = nn
        own = state["owner"]
        sel = np.ctypeslib.as_array(C.cast(L.or_mcts_sel(run), C.POINTER(C.c_int32)), shape=(n,))
        fresh = np.ctypeslib.as_array(C.cast(L.or_mcts_fresh(run), C.POINTER(C.c_uint8)), shape=(n,))
        by_game = None
        for s in sims:
            s.grow(nn)
            if phase == 1 and s.kind != "none":
                # what the ROOT launch could have carried: the roots' children exist before the evaluation (rules + frozen dice
                # only), their priors do not -- first come, first served, games in turn
                room = B - n
                ch = [[int(nodes[gi]["first_child"]) + j for j in range(int(nodes[gi]["n_children"]))] for gi in range(n)]
                picked = []
                j = 0
                while room > 0 and any(j < len(c) for c in ch):
                    for gi in range(n):
                        if j < len(ch[gi]) and room > 0 and not terminal(nodes[ch[gi][j]]["state"]):
                            picked.append(ch[gi][j]); room -= 1
                    j += 1
                if picked:
                    _, v2 = evaluate(np.stack([nodes[i]["state"] for i in picked]))
                    for i, x in zip(picked, v2):
                        s.cached[i] = True; s.cval[i] = x
            demanded = [int(sel[gi]) for gi in range(n) if fresh[gi]]
            miss = [not s.cached[i] for i in demanded]
            # a demanded row stays in the cache like any other (a drained leaf without children is evaluated every time it is reached, Q15)
            for gi in range(n):
                if fresh[gi]:
                    s.cval[sel[gi]] = val[gi]
            if not any(miss):
                s.account(n, False, 0, miss)
                continue
            for i in demanded:
                s.cached[i] = True
            room = B - sum(miss)
            spec = []
            if s.kind != "none" and room > 0:
                if by_game is None:
                    by_game = [np.nonzero(own == gi)[0] for gi in range(n)]
                share = [room // n + (1 if gi < room % n else 0) for gi in range(n)]
                for gi in range(n):
                    if share[gi] == 0:
                        continue
                    if s.kind == "top":
                        spec += candidates_top_prior(nodes, by_game[gi], s, share[gi], s.kw["per_parent"])
                    else:
                        rp = int(np.int8(nodes[gi]["state"][30]))
                        spec += candidates_rollout(nodes, gi, s, share[gi], cfgc, s.kw["vhat"], rp, set(demanded))
                spec = [i for i in dict.fromkeys(spec) if not s.cached[i]][:room]
                if spec:
                    _, v2 = evaluate(np.stack([nodes[i]["state"] for i in spec]))
                    for i, x in zip(spec, v2):
                        s.cached[i] = True; s.cval[i] = x
            s.account(n, True, len(spec), miss)
        state["t_sim"] += time.time() - t0

    fn = orc.EVAL_FN(cb)
    cfg = orc.MctsCfg(args.iterations, cfgc, 400, 0.3, 0.25)
    t = time.time()
    out = orc.self_play_parallel(1, args.games, cfg, 1.25, args.seed, fn, None, max_steps=args.max_steps)
    wall = time.time() - t

    bands = [(1, 1), (2, 2), (3, 4), (5, 8), (9, 16), (17, 32)]
    report = {"config": {"games": args.games, "iterations": args.iterations, "seed": hex(args.seed), "evaluator": args.eval, "budget_rows": B,
                         "max_live_simulated": args.max_live, "launch_us": args.launch_us, "skip_us": args.skip_us, "batch_s": args.batch_s},
              "move_steps": int(out["steps"]), "records": int(len(out["outcome"])), "oracle_wall_s": wall, "simulation_s": state["t_sim"],
              "move_steps_by_live": {f"{lo}-{hi}": sum(v for k, v in live_hist.items() if lo <= k <= hi) for lo, hi in bands + [(33, 1 << 20)]},
              "policies": []}
    for s in sims:
        rows = []
        saved_total = 0.0
        for lo, hi in bands:
            it = sum(d["iterations"] for m, d in s.by_m.items() if lo <= m <= hi)
            la = sum(d["launches"] for m, d in s.by_m.items() if lo <= m <= hi)
            gm = sum(d["game_misses"] for m, d in s.by_m.items() if lo <= m <= hi)
            gs = sum(d["game_selections"] for m, d in s.by_m.items() if lo <= m <= hi)
            sp = sum(d["spec_rows"] for m, d in s.by_m.items() if lo <= m <= hi)
            if not it:
                continue
            saved = (it - la) * (args.launch_us - args.skip_us) * 1e-6
            saved_total += saved
            rows.append({"live": f"{lo}-{hi}", "iterations": it, "launches": la, "launch_frac": round(la / it, 4),
                         "per_game_hit_rate": round(1.0 - gm / gs, 4) if gs else None, "spec_rows_per_launch": round(sp / max(la, 1), 2),
                         "saved_s": round(saved, 4)})
        report["policies"].append({"policy": s.name, "by_live": rows, "saved_s": round(saved_total, 4),
                                   "saved_frac_of_batch": round(saved_total / args.batch_s, 4)})
    txt = json.dumps(report, indent=1)
    print(txt)
    if args.out:
        with open(args.out, "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    main()
