"""TEST INFRASTRUCTURE (drives tests/tools/free_price.c, which includes the CPU oracle's source): prices a free-running search for
move-steps with many live games BEFORE it is built (round-5 review, items 4 and 6).

    python tests/tools/free_price.py --eval engine --out profiles/r06a_free_price.json      # on a GPU box: the engine's net evaluates
    python tests/tools/free_price.py --eval hash --plies 120 --rows 1024                     # anywhere: the oracle's hash evaluator

For each sampled move-step (the live games of a seeded random-walk batch at ply p: a random-init net plays close to uniformly at
random) and each launch size R it simulates (a) today's lockstep tail (calibration: compare `iterations_per_launch` with the
measured profiles/r05H_*, r05M_*) and (b) the free-running search under a few wish-list policies, and prints launches, rows,
waste and a time model against today's one launch per iteration:

    today      : iterations x (tower(m) + k_policy_fc + k_row_map + k_expand)
    free-running: launches x (tower(R) + k_policy_fc + k_free + pack)

with tower times from profiles/r05P_headline_kernel_stats.csv / the bench line's bands (us): see TOWER_US below.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


class Cfg(C.Structure):
    _fields_ = [("iterations", C.c_int), ("c", C.c_float), ("dir_alpha", C.c_float), ("dir_eps", C.c_float), ("rows", C.c_int),
                ("child_rows", C.c_int), ("cand_max", C.c_int), ("rollouts", C.c_int), ("order", C.c_int), ("lockstep", C.c_int),
                ("root_children", C.c_int), ("share_cap", C.c_int), ("preexpand", C.c_int), ("prio", C.c_int)]


class Out(C.Structure):
    _fields_ = [("launches", C.c_long), ("rows", C.c_long), ("rows_demanded", C.c_long), ("rows_used", C.c_long), ("iterations_run", C.c_long),
                ("game_rounds", C.c_long), ("stalls", C.c_long), ("adv_hist", C.c_long * 16), ("rounds_of_game_max", C.c_long),
                ("rounds_of_game_sum", C.c_long)]


EVAL_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p)


def build():
    so = os.path.join(HERE, "_free_price.so")
    src = os.path.join(HERE, "free_price.c")
    dep = os.path.join(ROOT, "oracle", "diee_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(dep)):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", so, src, "-lm"])
    L = C.CDLL(so)
    L.fp_run.argtypes = [C.c_void_p, C.c_int, C.POINTER(Cfg), C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, EVAL_FN, C.c_void_p, C.POINTER(Out)]
    L.fp_run.restype = C.c_int
    L.fp_states_at_ply.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    L.fp_states_at_ply.restype = C.c_int
    L.or_game_by_id.restype = C.c_void_p
    L.or_hash_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    return L


# measured average launch durations (us), round 5 final build: profiles/r05P_headline_kernel_stats.csv, bands of profiles/r05P_headline_line.json
def tower_us(rows):
    if rows <= 32: return 82.0
    if rows <= 64: return 110.0
    if rows <= 128: return 162.0
    if rows <= 256: return 258.0
    if rows <= 512: return 300.0
    if rows <= 640: return 264.0 + 60.0      # <4,8,6>: 264 us at ~580 rows (the compacted remainder launches pull the average down)
    if rows <= 928: return 484.0
    return 577.0 * ((rows + 1023) // 1024)


FC_US, ROWMAP_US, EXPAND_US, FREE_US, PACK_US, GAP_US = 12.0, 5.0, 18.0, 45.0, 5.0, 4.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--eval", choices=("hash", "engine"), default="hash")
    ap.add_argument("--games", type=int, default=1024)
    ap.add_argument("--iterations", type=int, default=100)
    ap.add_argument("--plies", type=int, nargs="*", default=[70, 90, 105, 115, 125, 140, 160, 190])
    ap.add_argument("--rows", type=int, nargs="*", default=[512, 1024])
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0xD1EE0001)
    ap.add_argument("--max-games", type=int, default=1024, help="cap the live games of a sampled move-step (speed)")
    ap.add_argument("--quick", action="store_true", help="fewer policies")
    ap.add_argument("--hash-value-scale", type=float, default=0.06, help="--eval hash: scale of the stand-in values (a random-init net's values are near 0)")
    ap.add_argument("--hash-logit-scale", type=float, default=0.25, help="--eval hash: flattens the stand-in policy (p ** scale, renormalised)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    L = build()
    game = L.or_game_by_id(1)
    A = 1352
    eng = None
    if args.eval == "engine":
        import diee_amd
        from oracle import oracle as orc
        eng = diee_amd.Engine(0); eng.load_weights(diee_amd.random_weights(0))
        BG = orc.BG_STATE

    def cb(ctx, sp, n, pp, vp):
        if eng is None:
            L.or_hash_eval(game, sp, n, pp, vp)
            pol = np.ctypeslib.as_array(C.cast(pp, C.POINTER(C.c_float)), shape=(n, A)); val = np.ctypeslib.as_array(C.cast(vp, C.POINTER(C.c_float)), shape=(n,))
            val *= np.float32(args.hash_value_scale)
            pol[:] = pol ** np.float32(args.hash_logit_scale); pol /= pol.sum(1, keepdims=True)
            return
        st = np.ctypeslib.as_array(C.cast(sp, C.POINTER(C.c_uint8)), shape=(n, 32)).copy().view(BG).reshape(-1)
        # rows of every launch are evaluated at a batch > 128 so that one arithmetic family (the fused one) serves every row, as in the engine
        pad = max(0, 129 - n)
        if pad:
            st = np.concatenate([st, np.repeat(st[:1], pad)])
        pol, val = eng.forward_t(st)
        pol = np.ascontiguousarray(pol[:n], np.float32); val = np.ascontiguousarray(val[:n], np.float32)
        C.memmove(pp, pol.ctypes.data, pol.nbytes); C.memmove(vp, val.ctypes.data, val.nbytes)
    fn = EVAL_FN(cb)

    def run(states, rounds, **kw):
        m = len(states)
        d = dict(iterations=args.iterations, c=2.0, dir_alpha=0.3, dir_eps=0.25, rows=1024, child_rows=16, cand_max=8, rollouts=24, order=1,
                 lockstep=0, root_children=0, share_cap=0, preexpand=0, prio=0)
        d.update(kw)
        cfg = Cfg(**d); out = Out()
        gids = np.arange(m, dtype=np.uint32)
        t = time.time()
        L.fp_run(states.ctypes.data, m, C.byref(cfg), args.seed, 7, gids.ctypes.data, rounds.ctypes.data, fn, None, C.byref(out))
        r = {k: getattr(out, k) for k, _ in Out._fields_ if k != "adv_hist"}
        r["adv_hist"] = list(out.adv_hist)
        r["sim_s"] = round(time.time() - t, 2)
        r["iterations_per_launch"] = round(args.iterations / max(out.launches - 1, 1), 3)          # (the root launch is not an iteration's)
        r["waste_frac"] = round(1.0 - out.rows_used / max(out.rows, 1), 4)
        r["mean_advance"] = round(out.iterations_run / max(out.game_rounds, 1), 3)
        return r

    report = {"config": {"games": args.games, "iterations": args.iterations, "evaluator": args.eval, "seed": hex(args.seed),
                         "time_model_us": {"fc": FC_US, "row_map": ROWMAP_US, "k_expand": EXPAND_US, "k_free": FREE_US, "pack": PACK_US, "gap": GAP_US}},
              "move_steps": []}
    for ply in args.plies:
        states = np.zeros((args.games, 32), np.uint8); rounds = np.zeros(args.games, np.uint32)
        m = L.fp_states_at_ply(args.seed, args.games, ply, states.ctypes.data, rounds.ctypes.data)
        m = min(m, args.max_games)
        if m == 0:
            continue
        states = np.ascontiguousarray(states[:m]); rounds = np.ascontiguousarray(rounds[:m])
        today_us = args.iterations * (tower_us(m) + FC_US + (ROWMAP_US if m > 256 else 0) + EXPAND_US + GAP_US)
        entry = {"ply": ply, "live_games": m, "today_ms": round(today_us / 1e3, 2), "runs": []}
        for R in args.rows:
            if R < m:
                continue
            variants = [("lockstep (today's k_tail policy)", dict(lockstep=1, cand_max=24)),
                        ("free, demanded only", dict(child_rows=0, cand_max=0)),
                        ("free, 8 cands then children (order 0)", dict(order=0)),
                        ("free, first cand, children, cands (order 1)", dict(order=1)),
                        ("free, 8 cands, no children, PRE-EXPANSION (path-keyed dice)", dict(order=0, child_rows=0, preexpand=1))]
            if not args.quick:
                variants += [("free, 8 cands, no children", dict(order=0, child_rows=0)),
                             ("free, 16 cands then children", dict(order=0, cand_max=16)),
                             ("free, 24 cands then children, 48 descents", dict(order=0, cand_max=24, rollouts=48)),
                             ("free, 8 cands then children, games behind first (prio 1)", dict(order=0, prio=1)),
                             ("free, 8 cands then children, games behind take all (prio 2)", dict(order=0, prio=2)),
                             ("free, 16 cands then children, prio 1", dict(order=0, cand_max=16, prio=1)),
                             ("free, 16 cands then children, prio 1 + root children", dict(order=0, cand_max=16, prio=1, root_children=1)),
                             ("free, order 2 (children first when the descent deepens)", dict(order=2)),
                             ("free, 8 cands, no children, PRE-EXPANSION (path-keyed dice)", dict(order=0, child_rows=0, preexpand=1)),
                             ("free, 16 cands, no children, PRE-EXPANSION", dict(order=0, child_rows=0, preexpand=1, cand_max=16, rollouts=48))]
            for name, kw in variants:
                if kw.get("lockstep") and m > 512:
                    continue
                r = run(states, rounds, rows=R, **kw)
                per = tower_us(R) + FC_US + FREE_US + PACK_US + GAP_US
                r.update(policy=name, rows_per_launch=R, model_ms=round((r["launches"] - 1) * per / 1e3 + (tower_us(m) + FC_US + EXPAND_US) / 1e3, 2))
                r["saving_frac"] = round(1.0 - r["model_ms"] / entry["today_ms"], 4)
                entry["runs"].append(r)
                print(f"ply {ply:3d} m {m:4d} R {R:4d} {name:58s} launches {r['launches']:4d} it/launch {r['iterations_per_launch']:6.2f} "
                      f"waste {r['waste_frac']:.2f} max-rounds {r['rounds_of_game_max']:3d} model {r['model_ms']:7.2f} ms vs today {entry['today_ms']:7.2f} "
                      f"({100 * r['saving_frac']:+.1f} %)  [{r['sim_s']} s]", flush=True)
        report["move_steps"].append(entry)
    if args.out:
        with open(args.out, "w") as f:
            f.write(json.dumps(report, indent=1) + "\n")


if __name__ == "__main__":
    main()
