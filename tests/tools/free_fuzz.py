#!/usr/bin/env python3
"""Randomised self-consistency run of the free-running search: the engine against ITSELF (the launch-per-iteration search, which the
parity suite holds to the oracle), so that hundreds of sizes / seeds / option mixes cost seconds instead of oracle minutes.

    python tests/tools/free_fuzz.py [cases=300] [seed=1] [iters=8,16,...]
    python tests/tools/free_fuzz.py mode=selfplay [cases=20] [seed=1]      # whole batches: 20 ... 900 games (and 2-3 batches side by side)
                                                                          # played to completion through all three search paths

Every case: n in 2 ... 800 roots drawn from random self-play walks (a random share of bear-off positions), 8 ... 100 iterations, quirks on
or off, random game ids / rounds / seeds, and a random mix of the path's options (LDS capacity, ring length, rows, candidates, iteration
cap, lag bonus).  probs, root visits, child counts and every counter must agree bit for bit; prints one line per failure and a summary.
Not a test of the suite (the oracle is not in it): a development tool, its log goes to profiles/ (r06E, r06F)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))

KEYS = ("nn_evals", "expansions", "children", "terminal_hits", "depth_sum", "selections", "max_children")
DEFAULTS = dict(free_eval=1, free_min_games=17, free_max_games=800, free_rows1024_from=200, free_rollout_steps=24, free_cand_max=12,
                free_ring=128, free_lds_nodes=3072, free_iter_cap=4, free_lag_boost=4, free_lag_step=4, spec_eval=1)


TAIL_DEFAULTS = dict(tower_table="default", spec_rollout_steps=24, spec_child_rows=16, spec_extra_rows=2, spec_max_games=96, spec_rows64_from=5, spec_rows128_from=10)


def selfplay(cases, seed):
    """whole batches through the default dispatch (launch per iteration >= 801 live games, free-running 17 ... 800, k_tail <= 16) against the
    same batches with every move-step on the launch-per-iteration search: all records, outcomes and counters bit-identical"""
    import diee_amd
    rng = np.random.default_rng(seed)
    eng = diee_amd.Engine(0)
    eng.load_weights(diee_amd.random_weights(0))
    bad = 0
    t0 = time.time()
    for c in range(cases):
        iters = int(rng.choice([8, 12, 16, 24, 40]))
        cfg = diee_amd.MctsConfig(iterations=iters, c=2.0, round_limit=int(rng.choice([60, 120, 400])), dir_alpha=0.3, dir_eps=0.25)
        temp = float(rng.choice([1.0, 1.25]))
        quirks = bool(rng.integers(0, 4) != 0)
        K = int(rng.choice([1, 1, 2, 3]))
        sizes = [int(rng.integers(20, 900 // K + 1)) for _ in range(K)]
        batches = [(n, 5000 * k + int(rng.integers(0, 1000)), int(rng.integers(1, 2**31))) for k, n in enumerate(sizes)]
        opts = {}
        if rng.random() < 0.4:
            opts = dict(free_ring=int(rng.choice([4, 16, 128])), free_iter_cap=int(rng.choice([1, 4, 6])), free_lds_nodes=int(rng.choice([64, 1024, 3072])),
                        free_rows1024_from=int(rng.choice([129, 200, 1024])), free_max_games=int(rng.choice([300, 800, 928])))
        res = {}
        try:
            for name, o in (("plain", dict(free_eval=0, spec_eval=0)), ("default", {**DEFAULTS, **opts})):
                eng.set_options(**o)
                if K == 1:
                    n, first, sd = batches[0]
                    res[name] = [eng.self_play_parallel(n, cfg, temp, seed=sd, ref_quirks=quirks, first_game_id=first)]
                else:
                    res[name] = eng.self_play_multi(batches, cfg, temp, ref_quirks=quirks)
        except diee_amd.DieeError as ex:
            bad += 1
            print(f"ERROR case {c}: batches {batches} iters {iters} quirks {quirks} opts {opts}: {ex}", flush=True)
            continue
        finally:
            eng.set_options(**DEFAULTS)
        ok = all(a["ps"].tobytes() == b["ps"].tobytes() and a["state"].tobytes() == b["state"].tobytes() and (a["outcome"] == b["outcome"]).all()
                 and (a["game"] == b["game"]).all() and all(a["stats"][k_] == b["stats"][k_] for k_ in KEYS) for a, b in zip(res["plain"], res["default"]))
        ran = sum(b["stats"]["tail_iterations"] for b in res["default"])
        print(f"case {c}: batches {[b[0] for b in batches]} x {iters} iterations, {sum(len(b['outcome']) for b in res['default'])} records, "
              f"{ran} iterations outside the launch-per-iteration search: {'identical' if ok else 'MISMATCH ' + str(opts) + str(batches)}", flush=True)
        bad += 0 if ok else 1
    eng.close()
    print(f"{cases} self-play cases (seed {seed}), {bad} mismatches, {time.time() - t0:.1f} s")
    return 1 if bad else 0


def main():
    kv = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
    if kv.get("mode") == "selfplay":
        return selfplay(int(kv.get("cases", 20)), int(kv.get("seed", 1)))
    # iterations per search (iters=1,2,3 on the command line for other mixes; 260: more rounds than the ring of 128 launches holds)
    iter_choices = [int(x) for x in kv["iters"].split(",")] if "iters" in kv else [8, 16, 24, 40, 64, 100, 100, 260]
    return search(int(kv.get("cases", 300)), int(kv.get("seed", 1)), iter_choices)


def search(cases, seed, iter_choices):
    """single searches (diee_mcts_batch): the default dispatch and random option mixes against the launch-per-iteration search"""
    import diee_amd
    from oracle import oracle                           # only random_walk_states: positions to search from
    oracle.build()
    rng = np.random.default_rng(seed)
    eng = diee_amd.Engine(0)
    eng.load_weights(diee_amd.random_weights(0))
    walks = [oracle.random_walk_states(1000 + k, 120) for k in range(6)]
    late = [w[w["off"].max(axis=1) >= 12] for w in walks]
    bad = 0
    t0 = time.time()
    free_cases = 0
    for c in range(cases):
        n = int(rng.choice([rng.integers(2, 17), rng.integers(17, 41), rng.integers(41, 129), rng.integers(129, 257), rng.integers(257, 513), rng.integers(513, 801)]))
        iters = int(rng.choice(iter_choices))
        w = int(rng.integers(0, len(walks)))
        share = float(rng.choice([0.0, 0.0, 0.3, 1.0]))
        k = min(int(n * share), len(late[w]))
        idx_l = rng.integers(0, max(1, len(late[w])), k)
        idx_r = rng.integers(0, len(walks[w]), n - k)
        states = np.concatenate([late[w][idx_l], walks[w][idx_r]]) if k else walks[w][idx_r]
        states = states[rng.permutation(n)]
        quirks = bool(rng.integers(0, 2))
        cfg = diee_amd.MctsConfig(iterations=iters, c=float(rng.choice([1.0, 2.0, 4.0])), round_limit=400, dir_alpha=0.3, dir_eps=float(rng.choice([0.0, 0.25])))
        gids = rng.permutation(4096)[:n].astype(np.uint32)
        rds = rng.integers(0, 60, n).astype(np.uint32)
        sd, call = int(rng.integers(1, 2**31)), int(rng.integers(0, 50))
        opts = dict(free_min_games=int(rng.choice([1, 17])))
        if rng.random() < 0.5:
            opts.update(free_lds_nodes=int(rng.choice([64, 256, 1024, 3072])), free_ring=int(rng.choice([4, 8, 128])),
                        free_rows1024_from=int(rng.choice([129, 200, 1024])), free_cand_max=int(rng.choice([0, 1, 12, 23])),
                        free_rollout_steps=int(rng.choice([1, 24, 48])), free_iter_cap=int(rng.choice([1, 4, 6, 1000])),
                        free_lag_boost=int(rng.choice([0, 4, 16])), free_lag_step=int(rng.choice([1, 4])))
        tail_case = n <= 256 and rng.random() < 0.35      # round 5's dispatch: k_tail up to 96 and at 129 ... 256 live games, with a random mix of ITS options
        if tail_case:
            opts = dict(free_eval=0, tower_table="928:5,640:14,512:6,256:10,128:11", spec_rollout_steps=int(rng.choice([0, 1, 24, 48])),
                        spec_child_rows=int(rng.choice([0, 4, 16])), spec_extra_rows=int(rng.choice([0, 2, 8])), spec_max_games=int(rng.choice([16, 64, 96, 128])),
                        spec_rows64_from=int(rng.choice([1, 5, 9])), spec_rows128_from=int(rng.choice([10, 20])))
        flags = dict(ref_quirks=quirks)
        inv = opts.get("free_min_games") == 1 and n <= 40    # (below 17 games the free-running search needs the fused family)
        eng.set_invariant_nn(inv)
        try:
            eng.set_options(free_eval=0, spec_eval=0, **({"tower_table": opts["tower_table"]} if tail_case else {}))
            a = eng.alpha_mcts_parallel(states, cfg, sd, call, gids, rds, **flags)
            eng.set_options(**{**DEFAULTS, **opts})
            b = eng.alpha_mcts_parallel(states, cfg, sd, call, gids, rds, **flags)
        except diee_amd.DieeError as ex:
            bad += 1
            print(f"ERROR case {c}: n {n} iters {iters} quirks {quirks} late share {share} opts {opts} seed {sd} call {call}: {ex}", flush=True)
            continue
        finally:
            eng.set_options(**DEFAULTS, **TAIL_DEFAULTS)
            eng.set_invariant_nn(False)
        ran_free = b["stats"]["tail_iterations"] == iters and a["stats"]["tail_iterations"] == 0
        free_cases += int(ran_free)
        ok = a["probs"].tobytes() == b["probs"].tobytes() and (a["root_visits"] == b["root_visits"]).all() and (a["n_children"] == b["n_children"]).all() \
            and all(a["stats"][k_] == b["stats"][k_] for k_ in KEYS)
        if not ok:
            bad += 1
            print(f"MISMATCH case {c}: n {n} iters {iters} quirks {quirks} late share {share} opts {opts} seed {sd} call {call}: "
                  + ", ".join(f"{k_} {a['stats'][k_]}/{b['stats'][k_]}" for k_ in KEYS if a['stats'][k_] != b['stats'][k_]), flush=True)
    eng.close()
    print(f"{cases} cases (seed {seed}), {free_cases} through the free-running search, {bad} mismatches, {time.time() - t0:.1f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
