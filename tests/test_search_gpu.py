"""GPU parity of the batched MCTS and of self-play: the HIP path (through the C ABI) against the CPU
oracle's restatement of alpha_mcts_parallel / self_play_parallel.  The oracle's evaluator is the
engine's own ResNet (called back through diee_nn_forward), so priors and values are identical on both
sides and every tree statistic, policy target and game record must agree BIT-EXACTLY."""
import ctypes as C

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 0xD1EE0001


@pytest.fixture(scope="module")
def eng():
    import diee_amd
    e = diee_amd.Engine(0)
    e.load_weights(diee_amd.random_weights(0))
    yield e
    e.close()


def gpu_eval(eng, oracle):
    calls = {"n": 0}

    def fn(states_u8):
        calls["n"] += 1
        st = states_u8.view(oracle.BG_STATE).reshape(-1)
        return eng.forward_t(st)
    return oracle.make_eval(fn, 1352), calls


def cfgs(oracle, iters, **kw):
    import diee_amd
    d = dict(iterations=iters, c=2.0, round_limit=400, dir_alpha=0.3, dir_eps=0.25); d.update(kw)
    return oracle.MctsCfg(**d), diee_amd.MctsConfig(**d)


@pytest.mark.parametrize("quirks", [1, 0])
@pytest.mark.parametrize("n,iters,pick", [(8, 24, "opening"), (24, 40, "mid"), (16, 30, "late"), (1, 16, "mid")])
def test_mcts_batch_bit_exact(eng, oracle, n, iters, pick, quirks):
    walk = oracle.random_walk_states(77, 40)
    if pick == "opening":
        states = walk[:n]
    elif pick == "mid":
        states = walk[200:200 + 7 * n:7]
    else:   # late: positions close to the end of games (terminal leaves, stale-slot quirk, bear-off)
        off = walk["off"].max(axis=1)
        states = walk[off >= 12][:n]
    assert len(states) == n
    ocfg, gcfg = cfgs(oracle, iters)
    ev, _ = gpu_eval(eng, oracle)
    gids = np.arange(100, 100 + n, dtype=np.uint32); rds = np.arange(n, dtype=np.uint32) % 5
    roots, probs, ostats, n_nodes = oracle.alpha_mcts_parallel(1, states, ocfg, ev, None, SEED, 3, gids, rds, quirks)
    r = eng.alpha_mcts_parallel(states, gcfg, SEED, 3, gids, rds, ref_quirks=bool(quirks))
    onch = np.array([len(x["children"]) for x in roots], dtype=np.uint32)
    assert (r["n_children"] == onch).all()
    assert (r["root_visits"] == np.array([x["visits"] for x in roots], dtype=np.float32)).all()
    assert r["probs"].tobytes() == probs.tobytes(), np.abs(np.nan_to_num(r["probs"]) - np.nan_to_num(probs)).max()
    gs, os_ = r["stats"], ostats.as_dict()
    for key in ("nn_evals", "expansions", "children", "terminal_hits", "depth_sum", "selections", "max_children"):
        assert gs[key] == os_[key], (key, gs[key], os_[key])
    assert gs["illegal_decodes"] == 0 and os_["code_collisions"] == 0
    if pick == "late":
        assert os_["terminal_hits"] > 0          # the terminal / stale-slot paths were exercised


@pytest.mark.parametrize("n", [40, 100, 200, 300, 500, 720])
def test_mcts_batch_bit_exact_across_tower_kernels(eng, oracle, n):
    """one batch size per network path: cluster tower with 2 / 4 / 8 boards per cluster (40, 100, 200 roots), fused tower
    with 2 boards per workgroup (300) and with 4 boards in border-aware order (500, 720): the search stays bit-exact"""
    walk = oracle.random_walk_states(90, 40)
    states = walk[50:50 + 4 * n:4]
    assert len(states) == n
    ocfg, gcfg = cfgs(oracle, 5)
    ev, _ = gpu_eval(eng, oracle)
    gids = np.arange(n, dtype=np.uint32); rds = np.arange(n, dtype=np.uint32) % 7
    roots, probs, ostats, _ = oracle.alpha_mcts_parallel(1, states, ocfg, ev, None, SEED, 1, gids, rds, 1)
    # round 6: above 128 roots the default search is the free-running one (tests/test_free_gpu.py); the launch-per-iteration search behind it
    # (free_eval = 0, spec_eval = 0: one plain / compacted evaluation per iteration on the kernel of that size) stays held to the oracle here
    for opts in (dict(), dict(free_eval=0, spec_eval=0)):
        eng.set_options(**opts)
        try:
            r = eng.alpha_mcts_parallel(states, gcfg, SEED, 1, gids, rds, ref_quirks=True)
        finally:
            eng.set_options(free_eval=1, spec_eval=1)
        assert r["probs"].tobytes() == probs.tobytes(), (opts, np.abs(np.nan_to_num(r["probs"]) - np.nan_to_num(probs)).max())
        assert (r["root_visits"] == np.array([x["visits"] for x in roots], dtype=np.float32)).all()
        gs, os_ = r["stats"], ostats.as_dict()
        for key in ("nn_evals", "expansions", "children", "terminal_hits", "depth_sum", "selections", "max_children"):
            assert gs[key] == os_[key], (opts, key, gs[key], os_[key])
        if opts:
            assert gs["tail_iterations"] == 0


@pytest.mark.parametrize("n,iters", [(4, 400), (2, 1600)])
def test_deep_tree_configs_bit_exact(eng, oracle, n, iters):
    """BASELINE configs[2] (iterations=400) and configs[3] (iterations=1600 deep tree) at reduced N"""
    walk = oracle.random_walk_states(31, 6)
    states = walk[50:50 + 9 * n:9]
    ocfg, gcfg = cfgs(oracle, iters)
    ev, _ = gpu_eval(eng, oracle)
    gids = np.arange(n, dtype=np.uint32); rds = np.zeros(n, dtype=np.uint32)
    roots, probs, ostats, n_nodes = oracle.alpha_mcts_parallel(1, states, ocfg, ev, None, SEED, 0, gids, rds, 1)
    r = eng.alpha_mcts_parallel(states, gcfg, SEED, 0, gids, rds, ref_quirks=True)
    assert r["probs"].tobytes() == probs.tobytes()
    gs, os_ = r["stats"], ostats.as_dict()
    for key in ("nn_evals", "expansions", "children", "terminal_hits", "depth_sum", "selections", "max_children"):
        assert gs[key] == os_[key], (key, gs[key], os_[key])
    assert gs["depth_sum"] / gs["selections"] > 1.5          # beyond the root level (uniform-ish priors give wide trees)


def test_self_play_bit_exact(eng, oracle):
    """whole games: records, policy targets (visits/sum)^(1/T) and outcomes identical to the oracle"""
    n, iters = 12, 12
    ocfg, gcfg = cfgs(oracle, iters, round_limit=60)     # a low round limit exercises the Q18 flush too
    ev, calls = gpu_eval(eng, oracle)
    ref = oracle.self_play_parallel(1, n, ocfg, 1.25, SEED, ev, None, ref_quirks=1, first_game_id=5)
    out = eng.self_play_parallel(n, gcfg, 1.25, SEED, ref_quirks=True, first_game_id=5)
    assert out["stats"]["move_steps"] == ref["steps"]
    assert len(out["outcome"]) == len(ref["outcome"]) > 0
    assert (out["game"] == ref["game"]).all()
    assert (out["outcome"] == ref["outcome"]).all()
    assert out["state"].tobytes() == ref["state"].tobytes()
    assert out["ps"].tobytes() == ref["ps"].tobytes()
    for key in ("nn_evals", "expansions", "children", "terminal_hits", "depth_sum", "selections"):
        assert out["stats"][key] == ref["stats"][key], key
    assert out["stats"]["plies"] == int(ref["plies"].sum())
    assert out["stats"]["games"] == n
    assert out["stats"]["illegal_decodes"] == 0 == ref["stats"]["illegal_decodes"]
    # invariants of the records (tests/mcts_test.rs:40-60 row-sum property, before the temperature)
    ps = out["ps"].astype(np.float64)
    assert np.allclose((ps ** 1.25).sum(1), 1.0, atol=1e-5)
    assert set(np.unique(out["outcome"])) <= {-1, 0, 1}


def test_self_play_plays_to_completion(eng, oracle):
    """full-length games, no round limit pressure: every game ends with a winner and +-1 labels"""
    import diee_amd
    n = 16
    _, gcfg = cfgs(oracle, 16)
    out = eng.self_play_parallel(n, gcfg, 1.25, SEED + 1, ref_quirks=True)
    st = out["stats"]
    assert st["games"] == n and st["fragments"] == len(out["outcome"])
    assert (np.bincount(out["game"], minlength=n) > 0).all()
    assert set(np.unique(out["outcome"])) <= {-1, 1}
    # every recorded state is a legal NN input of its game: player plane constant +-1
    pl = out["state"].reshape(-1, 6, 24)[:, 1, :]
    assert (np.abs(pl) == 1).all() and (pl == pl[:, :1]).all()
    # the label is +1 exactly when the mover of that state is the winner: within a game the sign flips with the mover
    for g in range(n):
        m = out["game"] == g
        assert (out["outcome"][m] * pl[m, 0] == (out["outcome"][m] * pl[m, 0])[0]).all()


# ---- pipelined self-play: K batches side by side, merged network launches -------------------------------------------
def _cmp_batch(m, ref, n, check_steps=True):
    if check_steps:
        assert m["stats"]["move_steps"] == ref["steps"]
    assert len(m["outcome"]) == len(ref["outcome"]) > 0
    assert (m["game"] == ref["game"]).all() and (m["outcome"] == ref["outcome"]).all()
    assert m["state"].tobytes() == ref["state"].tobytes()
    assert m["ps"].tobytes() == ref["ps"].tobytes()
    for key in ("nn_evals", "expansions", "children", "terminal_hits", "depth_sum", "selections", "max_children"):
        assert m["stats"][key] == ref["stats"][key], key
    assert m["stats"]["games"] == n and m["stats"]["illegal_decodes"] == 0


@pytest.mark.parametrize("quirks", [1, 0])
def test_self_play_multi_bit_exact_vs_lockstep_oracle(eng, oracle, quirks):
    """diee_self_play_multi against the oracle's restatement of K self_play_parallel calls in lockstep with ONE merged
    evaluator call per search phase (same row order: live games of batch 0, then 1, ...): records, policy targets,
    outcomes, per-batch counters and per-batch lengths identical"""
    ocfg, gcfg = cfgs(oracle, 10, round_limit=50)
    batches = [(10, 0, SEED), (6, 50, SEED + 1), (1, 7, SEED + 2), (12, 200, SEED)]
    ev, calls = gpu_eval(eng, oracle)
    ref, total = oracle.self_play_multi(1, batches, ocfg, 1.25, ev, None, ref_quirks=quirks)
    out = eng.self_play_multi(batches, gcfg, 1.25, ref_quirks=bool(quirks))
    assert max(o["stats"]["move_steps"] for o in out) == total
    for (n, first, seed), m, r in zip(batches, out, ref):
        _cmp_batch(m, r, n)
        assert set(np.unique(m["game"])) <= set(range(first, first + n))


def test_self_play_multi_equals_sequential_calls_when_the_network_is_batch_invariant(eng, oracle):
    """with DIEE_FLAG_INVARIANT_NN every batch size runs the fused 16x16x32 tower, the network output of a state no
    longer depends on what shares its launch, and K batches played side by side produce EXACTLY what K sequential
    diee_self_play calls produce -- at sizes where the default dispatch would switch kernels (130 + 140 games merged =
    fused tower, alone = cluster tower) -- and exactly what K oracle runs produce"""
    _, gcfg = cfgs(oracle, 3, round_limit=6)
    ocfg, _ = cfgs(oracle, 3, round_limit=6)
    batches = [(130, 0, SEED + 5), (140, 1000, SEED + 6), (9, 5000, SEED + 7)]
    multi = eng.self_play_multi(batches, gcfg, 1.25, ref_quirks=True, invariant_nn=True)
    eng.set_invariant_nn(True)
    try:
        ev, _ = gpu_eval(eng, oracle)
        for (n, first, seed), m in zip(batches, multi):
            single = eng.self_play_parallel(n, gcfg, 1.25, seed, ref_quirks=True, first_game_id=first, invariant_nn=True)
            for key in ("outcome", "game"):
                assert (m[key] == single[key]).all()
            assert m["ps"].tobytes() == single["ps"].tobytes() and m["state"].tobytes() == single["state"].tobytes()
            for key in ("nn_evals", "expansions", "children", "terminal_hits", "depth_sum", "selections", "move_steps", "plies"):
                assert m["stats"][key] == single["stats"][key], key
            if n <= 9:      # and the oracle, run on its own with the same (now pure) evaluator
                ref = oracle.self_play_parallel(1, n, ocfg, 1.25, seed, ev, None, ref_quirks=1, first_game_id=first)
                _cmp_batch(m, ref, n)
    finally:
        eng.set_invariant_nn(False)


def test_mcts_batch_bit_exact_above_one_chip_pass(eng, oracle):
    """more roots than one pass of the chip through the 4-board fused tower (1024): whole passes in one launch, the
    remainder (here 1100 - 1024 = 76 boards: cluster tower) in a launch of its own; the search stays bit-exact against
    the oracle, whose evaluator calls go through the same split"""
    n = 1100
    walk = oracle.random_walk_states(123, 40)
    states = walk[10:10 + 3 * n:3]
    assert len(states) == n
    ocfg, gcfg = cfgs(oracle, 3)
    ev, _ = gpu_eval(eng, oracle)
    gids = np.arange(n, dtype=np.uint32); rds = np.arange(n, dtype=np.uint32) % 3
    roots, probs, ostats, _ = oracle.alpha_mcts_parallel(1, states, ocfg, ev, None, SEED, 2, gids, rds, 1)
    r = eng.alpha_mcts_parallel(states, gcfg, SEED, 2, gids, rds, ref_quirks=True)
    assert r["probs"].tobytes() == probs.tobytes()
    gs, os_ = r["stats"], ostats.as_dict()
    for key in ("nn_evals", "expansions", "children", "terminal_hits", "depth_sum", "selections", "max_children"):
        assert gs[key] == os_[key], (key, gs[key], os_[key])


def test_compacted_evaluation_changes_nothing_but_the_row_count(oracle, monkeypatch):
    """above 256 live games the engine evaluates only the slots whose selected leaf was not terminal (the reference
    pushes all N rows and never reads the stale ones): visit distributions, counters and nn_evals (the reference's row
    count) are identical with the compaction on and off, only nn_rows (rows really evaluated) differs"""
    import diee_amd
    walk = oracle.random_walk_states(55, 60)
    late = walk[walk["off"].max(axis=1) >= 11]            # bear-off positions: terminal leaves within a few plies
    states = np.concatenate([late[:200], walk[100:100 + 3 * 400:3]])
    n = len(states)
    assert n == 600
    _, gcfg = cfgs(oracle, 12)
    gids = np.arange(n, dtype=np.uint32); rds = np.zeros(n, dtype=np.uint32)
    res = []
    for compact in ("1", "0"):
        e = diee_amd.Engine(0); e.load_weights(diee_amd.random_weights(0)); e.set_option("compact", compact)
        e.set_option("free_eval", 0)                      # (600 games: the launch-per-iteration search, where the compaction lives)
        res.append(e.alpha_mcts_parallel(states, gcfg, SEED, 4, gids, rds, ref_quirks=True))
        e.close()
    a, b = res
    assert a["probs"].tobytes() == b["probs"].tobytes() and (a["root_visits"] == b["root_visits"]).all()
    for key in ("nn_evals", "expansions", "children", "terminal_hits", "depth_sum", "selections", "max_children"):
        assert a["stats"][key] == b["stats"][key], key
    assert b["stats"]["nn_rows"] == b["stats"]["nn_evals"] == 13 * n
    assert a["stats"]["terminal_hits"] > 0
    assert a["stats"]["nn_rows"] == a["stats"]["nn_evals"] - a["stats"]["terminal_hits"]      # one row saved per terminal selection


@pytest.mark.parametrize("cap", ["1", "2", "3"])
def test_parent_walk_backpropagation_equals_the_recorded_path(oracle, monkeypatch, cap):
    """k_expand backpropagates over the root-to-leaf path its selection recorded (one lane per level); selections deeper
    than the record fall back to walking the parent indices.  option path_cap lowers the record so that ordinary searches
    reach the fallback in the stale-slot re-backpropagation (Q14), the terminal-leaf backpropagation of the descent and the
    ordinary one: results and counters identical to the default"""
    import diee_amd
    walk = oracle.random_walk_states(91, 60)
    late = walk[walk["off"].max(axis=1) >= 11][:24]
    states = np.concatenate([late, walk[150:150 + 5 * 40:5]])
    n = len(states)
    _, gcfg = cfgs(oracle, 48)
    gids = np.arange(n, dtype=np.uint32); rds = np.arange(n, dtype=np.uint32) % 7
    res = []
    for c in (None, cap):
        e = diee_amd.Engine(0); e.load_weights(diee_amd.random_weights(0))
        if c is not None: e.set_option("path_cap", c)
        res.append(e.alpha_mcts_parallel(states, gcfg, SEED, 5, gids, rds, ref_quirks=True))
        e.close()
    a, b = res
    assert a["probs"].tobytes() == b["probs"].tobytes() and (a["root_visits"] == b["root_visits"]).all()
    for key in ("nn_evals", "expansions", "children", "terminal_hits", "depth_sum", "selections", "max_children"):
        assert a["stats"][key] == b["stats"][key], key
    assert a["stats"]["terminal_hits"] > 0 and a["stats"]["depth_sum"] > 2 * a["stats"]["selections"]     # deeper than every cap tried


@pytest.mark.parametrize("n", [5, 20, 24, 40, 80])
def test_growth_workgroups_in_the_cluster_launch_change_nothing(oracle, monkeypatch, n):
    """while the cluster tower's grid leaves CUs free (1 ... 24, 33 ... 48, 65 ... 96 boards) the launch grows the tree of
    the leaf under evaluation on extra workgroups and k_expand<true> only commits the children; option cl_grow = 0 creates
    them after the evaluation (k_expand<false>): same visit distributions, same counters -- bear-off roots keep terminal
    and drained leaves (nothing to grow) in the mix, iterations = 60 reach arenas that are well filled"""
    import diee_amd
    walk = oracle.random_walk_states(77, 60)
    late = walk[walk["off"].max(axis=1) >= 11][:n // 3]
    states = np.concatenate([late, walk[120:120 + 4 * (n - len(late)):4]])
    assert len(states) == n
    _, gcfg = cfgs(oracle, 60)
    gids = np.arange(n, dtype=np.uint32) + 11; rds = np.arange(n, dtype=np.uint32) % 5
    e = diee_amd.Engine(0); e.load_weights(diee_amd.random_weights(0))
    # (the launch-per-iteration search on the cluster tower, where the growth workgroups live: since round 6 behind the free-running search,
    # the tail and -- above 40 boards -- the round-5 dispatch table)
    e.set_options(free_eval=0, spec_eval=0, tower_table="928:5,640:14,512:6,256:10,128:11")
    res = []
    for grow in ("1", "0"):
        e.set_option("cl_grow", grow)
        res.append(e.alpha_mcts_parallel(states, gcfg, SEED, 6, gids, rds, ref_quirks=True))
    e.close()
    a, b = res
    assert a["probs"].tobytes() == b["probs"].tobytes() and (a["root_visits"] == b["root_visits"]).all()
    assert (a["n_children"] == b["n_children"]).all()
    for key in ("nn_evals", "expansions", "children", "terminal_hits", "depth_sum", "selections", "max_children"):
        assert a["stats"][key] == b["stats"][key], key
    assert a["stats"]["terminal_hits"] > 0


@pytest.mark.parametrize("n,iters", [(20, 60), (40, 40), (300, 24), (1024, 8)])
def test_one_wave_and_two_wave_tree_kernels_agree(oracle, monkeypatch, n, iters):
    """k_expand runs on two waves per slot by default -- a growth wave (k_expand<true, 1>) or, where the tower launch grew the
    tree, a commit wave (<true, 2>) beside the main wave; options expand2 = 0 / expand2c = 0 bring the one-wave kernels back
    (<false, 0> / <true, 0>).  All four combinations: the same visit distributions and counters, from the tail's cluster
    launches (20, 40 roots) through the compacted pair tower (300) to the full chip (1024)"""
    import diee_amd
    walk = oracle.random_walk_states(83, 60)
    late = walk[walk["off"].max(axis=1) >= 11][:n // 4]
    rest = walk[100:100 + 3 * (n - len(late)):3]
    states = np.concatenate([late, rest])[:n]
    assert len(states) == n
    _, gcfg = cfgs(oracle, iters)
    gids = np.arange(n, dtype=np.uint32) + 5; rds = np.arange(n, dtype=np.uint32) % 3
    e = diee_amd.Engine(0); e.load_weights(diee_amd.random_weights(0))
    e.set_option("free_eval", 0)                          # (300 roots: the launch-per-iteration search, where k_expand runs every iteration)
    res = []
    for two, two_c in (("1", "1"), ("0", "1"), ("1", "0"), ("0", "0")):
        e.set_options(expand2=two, expand2c=two_c)
        res.append(e.alpha_mcts_parallel(states, gcfg, SEED, 7, gids, rds, ref_quirks=True))
    e.close()
    a = res[0]
    for b in res[1:]:
        assert a["probs"].tobytes() == b["probs"].tobytes() and (a["root_visits"] == b["root_visits"]).all()
        assert (a["n_children"] == b["n_children"]).all()
        for key in ("nn_evals", "expansions", "children", "terminal_hits", "depth_sum", "selections", "max_children"):
            assert a["stats"][key] == b["stats"][key], key
    assert a["stats"]["terminal_hits"] > 0


@pytest.mark.skipif(os.environ.get("DIEE_LONG_TESTS") != "3", reason="~8 minutes (single-core C oracle in lockstep, 37 k merged evaluator call-backs of 4096 rows): DIEE_LONG_TESTS=3")
def test_pipelined_leg_whole_workload_bit_exact_vs_lockstep_oracle(eng, oracle):
    """bench.py's second figure at its full size -- FOUR self_play_parallel batches of 1024 games, iterations = 100, played side by side
    to completion through diee_self_play_multi (the learn loop's self_play_iterations = 4) -- against the oracle's lockstep restatement:
    every batch's records, policy targets, outcomes and counters identical.  Opt-in for its length."""
    ocfg, gcfg = cfgs(oracle, 100)
    batches = [(1024, 0, SEED + 0x9E37 * i) for i in range(4)]
    ev, calls = gpu_eval(eng, oracle)
    ref, total = oracle.self_play_multi(1, batches, ocfg, 1.25, ev, None, ref_quirks=1)
    out = eng.self_play_multi(batches, gcfg, 1.25, ref_quirks=True)
    assert max(o["stats"]["move_steps"] for o in out) == total
    recs = 0
    for (n, first, seed), m, r in zip(batches, out, ref):
        _cmp_batch(m, r, n)
        recs += len(m["outcome"])
    print(f"[parity] pipelined leg, 4 x 1024 games x iterations 100 to completion: {recs} records, {total} move-steps: bit-exact per batch vs the lockstep oracle")
