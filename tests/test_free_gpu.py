"""The free-running search (die-e_amd/csrc/search_types.h `Free`, k_free / k_free_pack; round 6): at 257 ... 768 live games every game
runs its OWN iteration counter -- a round is one launch of up to 512 / 1024 rows (demanded leaves + the nodes virtual descents predict)
and one kernel in which each game iterates for as long as its leaf's evaluation is at hand.  Nothing of that may show in a result: every
case holds the engine to the CPU oracle's LOCKSTEP search (the reference's loop, alpha_mcts.rs:149-200) bit for bit, with the path on and
off, and asserts that the path really ran and really saved launches.  What couples the games of a batch -- `node_selected`, the stale
slots (Q14) -- is exercised where it bites: bear-off positions, few games, quirks on."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 0xD1EE0001
KEYS = ("nn_evals", "expansions", "children", "terminal_hits", "depth_sum", "selections", "max_children")
DEFAULTS = dict(free_eval=1, free_min_games=129, free_max_games=800, free_rows1024_from=200, free_rollout_steps=24, free_cand_max=12,
                free_ring=128, free_lds_nodes=3072, free_iter_cap=4)


@pytest.fixture(scope="module")
def eng():
    import diee_amd
    e = diee_amd.Engine(0)
    e.load_weights(diee_amd.random_weights(0))
    yield e
    e.close()


def gpu_eval(eng, oracle):
    def fn(states_u8):
        return eng.forward_t(states_u8.view(oracle.BG_STATE).reshape(-1))
    return oracle.make_eval(fn, 1352)


def cfgs(oracle, iters, **kw):
    import diee_amd
    d = dict(iterations=iters, c=2.0, round_limit=400, dir_alpha=0.3, dir_eps=0.25); d.update(kw)
    return oracle.MctsCfg(**d), diee_amd.MctsConfig(**d)


def roots_of(oracle, n, pick):
    walk = oracle.random_walk_states(321, 80)
    late = walk[walk["off"].max(axis=1) >= 12]
    if pick == "late" and len(late) < n + 2:             # (many bear-off positions: more games)
        walk = oracle.random_walk_states(321, 400)
        late = walk[walk["off"].max(axis=1) >= 12]
    if pick == "late":                                   # bear-off: terminal leaves, idle iterations, stale slots (Q14), drained leaves (Q15)
        return late[2:2 + n]
    if pick == "mixed":                                  # a third bear-off, the rest from the opening to the middle game
        k = min(n // 3, len(late))
        rest = walk[np.linspace(5, len(walk) - 1, n - k).astype(int)]
        return np.concatenate([late[:k], rest])
    return walk[np.linspace(5, len(walk) - 1, n).astype(int)]


def check(res, roots, probs, ostats, name):
    assert res["probs"].tobytes() == probs.tobytes(), (name, np.abs(np.nan_to_num(res["probs"]) - np.nan_to_num(probs)).max())
    assert (res["root_visits"] == np.array([x["visits"] for x in roots], dtype=np.float32)).all(), name
    assert (res["n_children"] == np.array([len(x["children"]) for x in roots], dtype=np.uint32)).all(), name
    for key in KEYS:
        assert res["stats"][key] == ostats[key], (name, key, res["stats"][key], ostats[key])


@pytest.mark.parametrize("quirks", [1, 0])
@pytest.mark.parametrize("n,iters,pick", [(129, 40, "mixed"), (199, 40, "mid"), (200, 40, "mixed"), (257, 40, "mixed"), (300, 100, "mid"), (448, 30, "mixed"), (513, 40, "mixed"),
                                          (600, 100, "mid"), (700, 24, "late"), (768, 40, "mixed")])
def test_free_running_search_bit_exact_vs_oracle(eng, oracle, n, iters, pick, quirks):
    states = roots_of(oracle, n, pick)
    assert len(states) == n
    ocfg, gcfg = cfgs(oracle, iters)
    gids = np.arange(7, 7 + n, dtype=np.uint32); rds = (np.arange(n, dtype=np.uint32) * 5) % 13
    roots, probs, ostats, _ = oracle.alpha_mcts_parallel(1, states, ocfg, gpu_eval(eng, oracle), None, SEED, 4, gids, rds, quirks)
    os_ = ostats.as_dict()
    res = {}
    for name, opts in (("free-running", dict(free_eval=1)), ("demanded rows only", dict(free_eval=1, free_cand_max=0)), ("launch per iteration", dict(free_eval=0, spec_eval=0))):
        eng.set_options(**opts)
        try:
            res[name] = eng.alpha_mcts_parallel(states, gcfg, SEED, 4, gids, rds, ref_quirks=bool(quirks))
        finally:
            eng.set_options(spec_eval=1, **DEFAULTS)
        check(res[name], roots, probs, os_, name)
    f, d, p = (res[k]["stats"] for k in ("free-running", "demanded rows only", "launch per iteration"))
    assert f["tail_iterations"] == d["tail_iterations"] == iters and p["tail_iterations"] == 0
    assert d["tail_spec_rows"] == 0 and d["tail_launches"] <= iters + 1
    assert f["tail_launches"] <= d["tail_launches"]
    if pick != "late":                                   # the speculation pays: about 0.91 * n / rows launches per iteration (+ what the predictions miss)
        rows = 1024 if n >= 200 else 512
        assert f["tail_spec_rows"] > 0 and f["tail_launches"] <= min(1.0, 1.6 * n / rows + 0.1) * iters, (f["tail_launches"], iters)
    print(f"[free] {n} games x {iters} iterations ({pick}, quirks {quirks}): {f['tail_launches']} launches with rows, {f['tail_spec_rows']} speculative rows "
          f"(demanded only: {d['tail_launches']})")


@pytest.mark.parametrize("opts", [dict(free_lds_nodes=64), dict(free_ring=4), dict(free_rows1024_from=129), dict(free_rows1024_from=1024),
                                  dict(free_rollout_steps=48, free_cand_max=23), dict(free_rollout_steps=1, free_cand_max=1), dict(free_ring=4, free_lds_nodes=128), dict(free_iter_cap=1), dict(free_iter_cap=1000), dict(free_lag_boost=0), dict(free_lag_boost=16, free_lag_step=1),
                                  dict(free_iter_cap=1, free_ring=4, free_cand_max=23), dict(free_iter_cap=1, free_ring=8, free_rollout_steps=1, free_lds_nodes=256)])
def test_free_running_options_change_nothing(eng, oracle, opts):
    """the tree's nodes beyond the LDS capacity are read in place, a ring of 4 launches makes evaluations age out (they are demanded again:
    the same bits), 512- or 1024-row launches, many or few candidates, grants with and without a bonus for the games behind: the same search"""
    n, iters = 300, 48
    states = roots_of(oracle, n, "mixed")
    ocfg, gcfg = cfgs(oracle, iters)
    gids = np.arange(n, dtype=np.uint32); rds = np.arange(n, dtype=np.uint32) % 4
    roots, probs, ostats, _ = oracle.alpha_mcts_parallel(1, states, ocfg, gpu_eval(eng, oracle), None, SEED, 2, gids, rds, 1)
    eng.set_options(**opts)
    try:
        r = eng.alpha_mcts_parallel(states, gcfg, SEED, 2, gids, rds, ref_quirks=True)
    finally:
        eng.set_options(**DEFAULTS)
    check(r, roots, probs, ostats.as_dict(), str(opts))
    assert r["stats"]["tail_iterations"] == iters


@pytest.mark.parametrize("quirks", [1, 0])
@pytest.mark.parametrize("n,iters,pick", [(2, 60, "late"), (3, 100, "late"), (7, 64, "late"), (16, 48, "late"), (40, 40, "late"), (64, 60, "mixed"), (129, 30, "mixed")])
def test_free_running_where_the_games_are_coupled(eng, oracle, n, iters, pick, quirks):
    """few games in the bear-off: most leaves are finished games, iterations are idle, slots go stale (Q14), the batch's first slot
    re-backpropagates its root -- every flag word a game reads from another matters here.  The path is forced down to these sizes
    (free_min_games = 1) with the batch-invariant network flag, so that rows of the fused family are what the oracle's evaluator computes too."""
    states = roots_of(oracle, n, pick)
    assert len(states) == n
    ocfg, gcfg = cfgs(oracle, iters)
    gids = np.arange(90, 90 + n, dtype=np.uint32); rds = (np.arange(n, dtype=np.uint32) * 3) % 7
    eng.set_invariant_nn(True)
    eng.set_options(free_min_games=1, spec_eval=0)          # (spec_eval = 0: k_tail, which would take 129 games, stays out of the way)
    try:
        roots, probs, ostats, _ = oracle.alpha_mcts_parallel(1, states, ocfg, gpu_eval(eng, oracle), None, SEED, 5, gids, rds, quirks)
        r = eng.alpha_mcts_parallel(states, gcfg, SEED, 5, gids, rds, ref_quirks=bool(quirks))
        eng.set_options(free_eval=0)
        plain = eng.alpha_mcts_parallel(states, gcfg, SEED, 5, gids, rds, ref_quirks=bool(quirks))
    finally:
        eng.set_options(spec_eval=1, **DEFAULTS)
        eng.set_invariant_nn(False)
    check(r, roots, probs, ostats.as_dict(), "free-running")
    check(plain, roots, probs, ostats.as_dict(), "launch per iteration")
    assert r["stats"]["tail_iterations"] == iters and plain["stats"]["tail_iterations"] == 0


def test_free_running_self_play_whole_games_bit_exact(eng, oracle):
    """whole games: 400 games at iterations = 16 played to completion -- the move-steps with 257 ... 400 live games run free, the tail of the
    batch runs k_tail, the seam between them included -- records, policy targets, outcomes and counters equal to the oracle's"""
    ocfg, gcfg = cfgs(oracle, 16, round_limit=90)                                   # (a round limit some games hit: Q18 / Q19 too)
    ref = oracle.self_play_parallel(1, 400, ocfg, 1.25, 77, gpu_eval(eng, oracle), None, ref_quirks=1, first_game_id=1000)
    out = eng.self_play_parallel(400, gcfg, 1.25, seed=77, ref_quirks=True, first_game_id=1000)
    eng.set_option("free_eval", 0)
    try:
        plain = eng.self_play_parallel(400, gcfg, 1.25, seed=77, ref_quirks=True, first_game_id=1000)
    finally:
        eng.set_option("free_eval", 1)
    for o in (out, plain):
        assert o["ps"].tobytes() == ref["ps"].tobytes() and o["state"].tobytes() == ref["state"].tobytes()
        assert (o["outcome"] == ref["outcome"]).all() and (o["game"] == ref["game"]).all()
        for key in KEYS:
            assert o["stats"][key] == ref["stats"][key], key
    assert out["stats"]["tail_iterations"] > plain["stats"]["tail_iterations"] > 0
    assert out["stats"]["tail_launches"] < 0.9 * out["stats"]["tail_iterations"]


def test_free_running_with_batches_side_by_side(eng, oracle):
    """diee_self_play_multi: three batches of 120 games share the slot space (360 live games: the free-running path), each with its own
    `node_selected` flags and slot-0 bookkeeping (Q14): per batch the oracle's lockstep restatement, bit for bit"""
    ocfg, gcfg = cfgs(oracle, 12, round_limit=60)
    batches = [(120, 0, 171), (120, 1000, 172), (120, 2000, 173)]
    ref, _ = oracle.self_play_multi(1, batches, ocfg, 1.25, gpu_eval(eng, oracle), None, ref_quirks=1)
    out = eng.self_play_multi(batches, gcfg, 1.25, ref_quirks=True)
    for o, r in zip(out, ref):
        assert o["ps"].tobytes() == r["ps"].tobytes() and o["state"].tobytes() == r["state"].tobytes() and (o["outcome"] == r["outcome"]).all()
        for key in KEYS:
            assert o["stats"][key] == r["stats"][key], key
    assert out[0]["stats"]["tail_iterations"] > 0


def test_free_running_on_a_shared_gpu(oracle):
    """`shared_gpu` = 1 takes every kernel that waits for a co-resident workgroup out of the dispatch (cluster tower, pair tower, k_tail); the
    free-running search waits for nobody and stays on above 256 games: its rounds then run on the one-pass fused tower and, at 512 rows or fewer,
    on the 2-board geometry instead of the pair tower -- one arithmetic, the same bits as the oracle's lockstep search"""
    import diee_amd
    e = diee_amd.Engine(0)
    e.load_weights(diee_amd.random_weights(0))
    e.set_option("shared_gpu", 1)
    try:
        assert not any(k.startswith(("k_tower_cl", "k_tower16p")) for _, _, k in e.dispatch_bands(1024))
        for n, iters in ((300, 24), (600, 16)):
            states = roots_of(oracle, n, "mixed")
            ocfg, gcfg = cfgs(oracle, iters)
            gids = np.arange(n, dtype=np.uint32); rds = np.arange(n, dtype=np.uint32) % 3
            roots, probs, ostats, _ = oracle.alpha_mcts_parallel(1, states, ocfg, gpu_eval(e, oracle), None, SEED, 8, gids, rds, 1)
            r = e.alpha_mcts_parallel(states, gcfg, SEED, 8, gids, rds, ref_quirks=True)
            check(r, roots, probs, ostats.as_dict(), f"shared_gpu, {n} games")
            assert r["stats"]["tail_iterations"] == iters and r["stats"]["tail_launches"] < iters
    finally:
        e.close()


def test_free_running_reports_a_tree_arena_that_is_too_small(oracle):
    """nodes_per_expansion = 1: the games' trees outgrow their arena in mid-search; the free-running search completes its rounds (no game
    waits for a node that was never created), the call reports DIEE_ERR_CAPACITY, and the engine stays usable"""
    import diee_amd
    e = diee_amd.Engine(0)
    e.load_weights(diee_amd.random_weights(0))
    try:
        states = roots_of(oracle, 300, "mixed")
        e.set_option("nodes_per_expansion", 1)
        with pytest.raises(diee_amd.DieeError) as ei:
            e.alpha_mcts_parallel(states, diee_amd.MctsConfig.default(64), SEED, 0)
        assert ei.value.status == diee_amd.ERR_CAPACITY
        e.set_option("nodes_per_expansion", 128)
        r = e.alpha_mcts_parallel(states, diee_amd.MctsConfig.default(16), SEED, 0)
        e.set_options(free_eval=0, spec_eval=0)
        plain = e.alpha_mcts_parallel(states, diee_amd.MctsConfig.default(16), SEED, 0)
        assert r["stats"]["tail_iterations"] == 16 and plain["stats"]["tail_iterations"] == 0
        assert r["probs"].tobytes() == plain["probs"].tobytes() and (r["n_children"] == plain["n_children"]).all()
    finally:
        e.close()


def test_randomised_self_consistency_of_the_three_search_paths():
    """tests/tools/free_fuzz.py in small: random sizes, seeds, iteration counts and option mixes (the free-running search's and k_tail's), the
    default dispatch against the launch-per-iteration search -- which the cases above and tests/test_search_gpu.py hold to the oracle -- and
    three whole self-play workloads; everything bit-identical.  (The long runs are in profiles/r06E ... r06M; this tool found free_run's round bound.)"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("free_fuzz", os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "free_fuzz.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    assert fz.search(120, 777, [1, 3, 8, 24, 64, 100]) == 0
    assert fz.selfplay(3, 778) == 0
