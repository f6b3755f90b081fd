"""The tail of a batch (<= 96 and 129 ... 256 live games by default; die-e_amd/csrc/search_types.h `Tail`, k_tail): the iterations of a search run inside one
launch for as long as every selected leaf's evaluation is at hand, and the launches in between carry speculative rows.  Nothing of
that may show in a result: every case here holds the engine -- with the path on, with it off, and with the speculation alone off --
to the CPU oracle's lockstep search BIT FOR BIT, and asserts that the path really ran (`tail_iterations`) and really saved launches."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 0xD1EE0001
CHILD_ROWS = 16          # the default of option spec_child_rows
KEYS = ("nn_evals", "expansions", "children", "terminal_hits", "depth_sum", "selections", "max_children")


@pytest.fixture(scope="module")
def eng():
    import diee_amd
    e = diee_amd.Engine(0)
    e.load_weights(diee_amd.random_weights(0))
    # Round 6: by default k_tail keeps the move-steps with at most 16 live games -- the free-running search (tests/test_free_gpu.py) takes 17 ... 768,
    # and the fused kernel family starts at 41 boards instead of 129.  The looping kernel's whole range (<= 128 games on the cluster family's launches,
    # 129 ... 256 on the fused family's) stays in the product behind its options and stays held to the oracle here: round 5's dispatch table, free_eval = 0
    # per case.  (The oracle's evaluator runs on this engine: the same table decides its arithmetic.)
    e.set_option("tower_table", "928:5,640:14,512:6,256:10,128:11")
    e.set_option("free_eval", 0)
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
    walk = oracle.random_walk_states(123, 60)
    if pick == "late":                                   # bear-off: terminal leaves, idle iterations, stale slots (Q14), drained leaves (Q15)
        return walk[walk["off"].max(axis=1) >= 12][3:3 + n]
    if pick == "mixed":
        late = walk[walk["off"].max(axis=1) >= 11][:n // 2]
        return np.concatenate([late, walk[300:300 + 5 * (n - len(late)):5]])
    return walk[150:150 + 6 * n:6]


@pytest.mark.parametrize("quirks", [1, 0])
@pytest.mark.parametrize("n,iters,pick", [(1, 100, "mid"), (1, 100, "late"), (2, 100, "mixed"), (3, 64, "late"), (5, 100, "mixed"),
                                          (8, 100, "mid"), (16, 100, "mixed"), (16, 48, "late"), (4, 400, "mixed"),
                                          (24, 100, "mixed"), (33, 60, "mid"), (48, 100, "late"), (64, 100, "mixed"), (3, 1600, "mixed"),
                                          (65, 40, "mixed"), (96, 60, "mid"), (128, 40, "mixed"),
                                          (129, 30, "mid"), (200, 40, "mixed"), (256, 30, "mid")])       # (above 128: 512-row launches of the fused family)
def test_tail_search_bit_exact_vs_oracle(eng, oracle, n, iters, pick, quirks):
    states = roots_of(oracle, n, pick)
    assert len(states) == n
    ocfg, gcfg = cfgs(oracle, iters)
    gids = np.arange(40, 40 + n, dtype=np.uint32); rds = (np.arange(n, dtype=np.uint32) * 3) % 11
    roots, probs, ostats, _ = oracle.alpha_mcts_parallel(1, states, ocfg, gpu_eval(eng, oracle), None, SEED, 9, gids, rds, quirks)
    res = {}
    for name, opts in (("tail", dict(spec_eval=1, spec_rollout_steps=24)), ("demanded rows only", dict(spec_eval=1, spec_rollout_steps=0, spec_child_rows=0)),
                       ("launch per iteration", dict(spec_eval=0, spec_rollout_steps=24))):
        # (the default reach is 96 games: the 128-game case asks for the kernel's whole range; from 129 games on the free-running search
        # takes a move-step by default since round 6 -- tests/test_free_gpu.py --: off here, so that k_tail's fused-family launches stay held to the oracle)
        eng.set_options(spec_max_games=128, **opts)
        try:
            res[name] = eng.alpha_mcts_parallel(states, gcfg, SEED, 9, gids, rds, ref_quirks=bool(quirks))
        finally:
            eng.set_options(spec_eval=1, spec_rollout_steps=24, spec_max_games=96, spec_child_rows=CHILD_ROWS)
    os_ = ostats.as_dict()
    for name, r in res.items():
        assert r["probs"].tobytes() == probs.tobytes(), (name, np.abs(np.nan_to_num(r["probs"]) - np.nan_to_num(probs)).max())
        assert (r["root_visits"] == np.array([x["visits"] for x in roots], dtype=np.float32)).all(), name
        assert (r["n_children"] == np.array([len(x["children"]) for x in roots], dtype=np.uint32)).all(), name
        for key in KEYS:
            assert r["stats"][key] == os_[key], (name, key, r["stats"][key], os_[key])
    t, d, p = (res[k]["stats"] for k in ("tail", "demanded rows only", "launch per iteration"))
    assert t["tail_iterations"] == d["tail_iterations"] == iters and p["tail_iterations"] == 0
    assert d["tail_spec_rows"] == 0 and d["tail_launches"] <= iters          # no speculation: a launch per iteration that evaluates anything, none for idle ones
    assert t["tail_launches"] <= d["tail_launches"]
    if pick != "late":
        assert t["tail_spec_rows"] > 0 and t["tail_launches"] <= (0.6 if n <= 16 else 0.85 if n <= 64 else 1.0) * iters, (t["tail_launches"], iters)   # the speculation pays
        # (beyond 64 games a 128-row launch has fewer spare rows than games: at 128 games nearly every iteration has some game missing, and the
        # saving left is the idle iterations and the terminal leaves)
    assert t["nn_rows"] >= t["tail_spec_rows"]


@pytest.mark.parametrize("opts", [dict(spec_rows64_from=1, spec_rows128_from=2), dict(spec_rows64_from=65, spec_rows128_from=65),
                                  dict(spec_rows64_from=3, spec_rows128_from=65), dict(spec_max_games=7),
                                  dict(spec_rows64_from=65, spec_rows128_from=65, spec_extra_rows=0), dict(spec_rows64_from=65, spec_rows128_from=65, spec_extra_rows=9),
                                  dict(spec_rows64_from=65, spec_rows128_from=65, spec_child_rows=0), dict(spec_rows64_from=65, spec_rows128_from=65, spec_child_rows=3),
                                  dict(spec_rows64_from=65, spec_rows128_from=65, spec_child_rows=200, spec_extra_rows=0)])
def test_tail_rows_per_launch_change_nothing(eng, oracle, opts):
    """a tail launch carries 32, 64 or 128 rows depending on the live games (k_tower_cl<1, 8> / <2, 8> / <4, 8>: one arithmetic per row):
    whatever the thresholds say, wherever the path hands over to the launch-per-iteration search, and whether or not the games take the
    rows their neighbours leave free (spec_extra_rows: 9 games on 32 rows, scarce, so that the second round of claims runs) or send the
    children of a demanded leaf along with it (spec_child_rows), the same bits"""
    n, iters = 9, 48
    states = roots_of(oracle, n, "mixed")
    ocfg, gcfg = cfgs(oracle, iters)
    gids = np.arange(n, dtype=np.uint32); rds = np.arange(n, dtype=np.uint32) % 4
    roots, probs, ostats, _ = oracle.alpha_mcts_parallel(1, states, ocfg, gpu_eval(eng, oracle), None, SEED, 2, gids, rds, 1)
    eng.set_options(**opts)
    try:
        r = eng.alpha_mcts_parallel(states, gcfg, SEED, 2, gids, rds, ref_quirks=True)
        rows = eng.last_dispatch()
    finally:
        eng.set_options(spec_rows64_from=5, spec_rows128_from=10, spec_max_games=96, spec_extra_rows=2, spec_child_rows=CHILD_ROWS)
    assert r["probs"].tobytes() == probs.tobytes()
    for key in KEYS:
        assert r["stats"][key] == ostats.as_dict()[key], key
    want = {1: ("k_tower_cl<4, 8>", 128), 65: ("k_tower_cl<1, 8>", 32), 3: ("k_tower_cl<2, 8>", 64)}.get(opts.get("spec_rows64_from"))
    if want:
        assert rows == [want] and r["stats"]["tail_iterations"] == iters
    else:
        assert r["stats"]["tail_iterations"] == 0                    # 9 games > spec_max_games = 7: one launch per iteration


def test_tail_self_play_whole_games_bit_exact(eng, oracle):
    """whole games through the tail: 14 games at iterations = 24 played to completion -- every move-step of this batch runs the looping
    kernel -- records, policy targets, outcomes and counters equal to the oracle's, and to the engine with the path off"""
    ocfg, gcfg = cfgs(oracle, 24, round_limit=70)                                   # (a round limit some games hit: Q18 / Q19 too)
    ref = oracle.self_play_parallel(1, 14, ocfg, 1.25, 31, gpu_eval(eng, oracle), None, ref_quirks=1, first_game_id=500)
    out = eng.self_play_parallel(14, gcfg, 1.25, seed=31, ref_quirks=True, first_game_id=500)
    eng.set_option("spec_eval", 0)
    try:
        plain = eng.self_play_parallel(14, gcfg, 1.25, seed=31, ref_quirks=True, first_game_id=500)
    finally:
        eng.set_option("spec_eval", 1)
    for o in (out, plain):
        assert o["ps"].tobytes() == ref["ps"].tobytes() and o["state"].tobytes() == ref["state"].tobytes()
        assert (o["outcome"] == ref["outcome"]).all() and (o["game"] == ref["game"]).all()
        for key in KEYS:
            assert o["stats"][key] == ref["stats"][key], key
    st = out["stats"]
    assert st["tail_iterations"] == 24 * st["move_steps"] and plain["stats"]["tail_iterations"] == 0
    assert st["tail_launches"] < 0.7 * st["tail_iterations"]
    print(f"[tail] 14 games x iterations 24: {st['move_steps']} move-steps, {st['tail_iterations']} iterations on {st['tail_launches']} launches "
          f"({st['tail_spec_rows']} speculative rows); {st['seconds']:.2f} s against {plain['stats']['seconds']:.2f} s launch by launch")


def test_tail_with_batches_side_by_side(eng, oracle):
    """diee_self_play_multi in the tail: three batches of a few games share the looping kernel's lockstep, each with its own
    `node_selected` flags and slot-0 bookkeeping (Q14): per batch the oracle's lockstep restatement, bit for bit"""
    ocfg, gcfg = cfgs(oracle, 16, round_limit=50)
    batches = [(5, 0, 71), (4, 100, 72), (6, 200, 73)]
    ref, _ = oracle.self_play_multi(1, batches, ocfg, 1.25, gpu_eval(eng, oracle), None, ref_quirks=1)
    out = eng.self_play_multi(batches, gcfg, 1.25, ref_quirks=True)
    for o, r in zip(out, ref):
        assert o["ps"].tobytes() == r["ps"].tobytes() and o["state"].tobytes() == r["state"].tobytes() and (o["outcome"] == r["outcome"]).all()
        for key in KEYS:
            assert o["stats"][key] == r["stats"][key], key
    assert out[0]["stats"]["tail_iterations"] > 0


@pytest.mark.parametrize("quirks", [1, 0])
def test_finished_roots_in_the_tail(eng, oracle, quirks):
    """diee_mcts_batch does not refuse a root whose game is over (round-5 advisor): with children (the loser is to move and has plays:
    every child is a finished game) and without (the side to move has no checker left: every selection ends on the root itself).  The
    looping kernel tells a finished game from the node header's bits where the launch-per-iteration path reads the state: both equal the oracle."""
    walk = oracle.random_walk_states(123, 60)
    mid = walk[200:202].copy()
    fin = walk[walk["off"].max(axis=1) >= 13][:2].copy()
    for s in fin:                                     # player -1 (negative points, bar[0], off[0]) has borne off everything
        s["pts"][s["pts"] < 0] = 0; s["bar"][0] = 0; s["off"][0] = 15
    fin[0]["player"] = 1                              # the loser to move: legal plays exist, every child is a finished game
    fin[1]["player"] = -1                             # the winner "to move": no checker, no play, the root is the only node
    states = np.concatenate([fin[:1], mid[:1], fin[1:], mid[1:]])
    n, iters = len(states), 24
    ocfg, gcfg = cfgs(oracle, iters)
    gids = np.arange(n, dtype=np.uint32); rds = np.arange(n, dtype=np.uint32)
    roots, probs, ostats, _ = oracle.alpha_mcts_parallel(1, states, ocfg, gpu_eval(eng, oracle), None, SEED, 3, gids, rds, quirks)
    for spec in (1, 0):
        eng.set_option("spec_eval", spec)
        try:
            r = eng.alpha_mcts_parallel(states, gcfg, SEED, 3, gids, rds, ref_quirks=bool(quirks))
        finally:
            eng.set_option("spec_eval", 1)
        assert r["probs"].tobytes() == probs.tobytes(), spec
        assert (r["root_visits"] == np.array([x["visits"] for x in roots], dtype=np.float32)).all(), spec
        for key in KEYS:
            assert r["stats"][key] == ostats.as_dict()[key], (spec, key, r["stats"][key], ostats.as_dict()[key])


def test_row_thresholds_that_leave_fewer_rows_than_games_fall_back(eng, oracle):
    """options spec_rows64_from / spec_rows128_from can ask for 64-row launches at 80 games (round-5 advisor): a launch must hold every
    live game's demanded row, so such a move-step runs one launch per iteration instead -- same bits, no tail iterations"""
    n, iters = 80, 12
    states = roots_of(oracle, n, "mid")
    ocfg, gcfg = cfgs(oracle, iters)
    gids = np.arange(n, dtype=np.uint32); rds = np.zeros(n, dtype=np.uint32)
    _, probs, ostats, _ = oracle.alpha_mcts_parallel(1, states, ocfg, gpu_eval(eng, oracle), None, SEED, 1, gids, rds, 1)
    eng.set_options(spec_rows128_from=200)
    try:
        r = eng.alpha_mcts_parallel(states, gcfg, SEED, 1, gids, rds, ref_quirks=True)
    finally:
        eng.set_options(spec_rows128_from=10)
    assert r["probs"].tobytes() == probs.tobytes() and r["stats"]["tail_iterations"] == 0
