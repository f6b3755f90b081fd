"""GPU tests at BASELINE.json's full sizes through size-independent properties (the oracle cannot
follow at these sizes in seconds): legal-play round trips on ~1M states, and a 1024-game / 100-iteration
self-play batch (truncated to a few move-steps) whose records must satisfy the domain invariants."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import diee_amd
    e = diee_amd.Engine(0)
    e.load_weights(diee_amd.random_weights(0))
    yield e
    e.close()


def test_million_state_round_trips(eng, oracle):
    states = oracle.random_walk_states(424242, 5000)                 # ~0.55 M reachable states
    states = np.concatenate([states, states[::-1][:450000]])
    plays, counts = eng.get_valid_moves(states, 160)
    assert counts.max() < 160
    idx_s, idx_p = np.nonzero(np.arange(160)[None, :] < counts[:, None])
    st = states[idx_s]; pl = plays[idx_s, idx_p]
    codes = eng.encode(st, pl)
    assert (codes < 1351).all()
    assert (eng.decode(st, codes) == pl).all()                      # decode(encode(play)) == play for every legal play
    # codes are unique within a state (the policy scatter of utils.rs:42-58 relies on it)
    key = idx_s.astype(np.int64) * 2048 + codes
    assert len(np.unique(key)) == len(key)
    # every play has 1 or 2 moves and the first move starts on a point of the mover (or the bar)
    assert ((pl[:, 0] >= -1) & (pl[:, 0] <= 23)).all()
    pts = st["pts"].astype(np.int64); player = st["player"].astype(np.int64)
    frm = pl[:, 0].astype(np.int64)
    on_board = frm >= 0
    assert (pts[np.arange(len(st)), np.where(on_board, frm, 0)][on_board] * player[on_board] >= 1).all()
    bar_own = np.where(player < 0, st["bar"][:, 0], st["bar"][:, 1])
    assert (bar_own[~on_board] > 0).all() and (bar_own[on_board] == 0).all()
    # applying a play conserves 15 checkers per side
    sel = np.random.default_rng(0).choice(len(st), 300000, replace=False)
    dice = np.random.default_rng(1).integers(1, 7, size=(len(sel), 2)).astype(np.uint8)
    nxt = eng.apply_move(st[sel], pl[sel], dice)
    p = nxt["pts"].astype(np.int64)
    assert ((-np.minimum(p, 0)).sum(1) + nxt["bar"][:, 0] + nxt["off"][:, 0] == 15).all()
    assert ((np.maximum(p, 0)).sum(1) + nxt["bar"][:, 1] + nxt["off"][:, 1] == 15).all()
    # a state without plays skips: count 0 states exist and are few
    assert 0 < (counts == 0).mean() < 0.2


def test_full_size_batch_invariants(eng):
    """BASELINE configs[1] geometry: 1024 games, iterations=100 (first 3 move-steps)"""
    import diee_amd
    cfg = diee_amd.MctsConfig(iterations=100, c=2.0, round_limit=400, dir_alpha=0.3, dir_eps=0.25)
    out = eng.self_play_parallel(1024, cfg, 1.25, 0xD1EE0001, ref_quirks=True, max_steps=3, fetch=False)
    st = out["stats"]
    assert st["move_steps"] == 3 and st["plies"] == 3 * 1024 and st["games"] == 0
    assert st["nn_evals"] == 3 * 101 * 1024                          # the batch is always N (alpha_mcts.rs:175-183)
    assert st["selections"] == 3 * 100 * 1024
    assert st["expansions"] <= 3 * 101 * 1024 and st["expansions"] >= 0.9 * 3 * 101 * 1024
    assert st["illegal_decodes"] == 0 and st["max_children"] < 256
    assert 5 < st["children"] / st["expansions"] < 40
    assert st["tower_launches"] > 0                                  # the fused tower kernel ran (batch > 500)
    # level-1 API at full width: rows sum to 1 (tests/mcts_test.rs:40-60), root visits = iterations + 1 (Q13)
    import numpy as np
    from oracle import oracle as orc
    roots = orc.random_walk_states(5, 40)[:1024]
    r = eng.alpha_mcts_parallel(roots, cfg, 7, 0)
    has = r["n_children"] > 0
    assert np.allclose(r["probs"][has].sum(1), 1.0, atol=1e-5)
    assert np.isnan(r["probs"][~has]).all()
    assert (r["root_visits"] >= 101).all() and (r["root_visits"] <= 101 + 1024 * 100).all()
    # Q14: a game whose leaf is terminal re-backpropagates its stale slot: at most one extra visit per iteration;
    # only slot 0 can additionally collect the stale-initial re-backprops of other games
    assert (r["root_visits"][1:] <= 201).all()
    assert (r["root_visits"] == 101).mean() > 0.5


def test_config4_deep_tree_full_size_properties(eng, oracle):
    """BASELINE configs[3] at its full size: 1024 roots x iterations = 1600 (205 k-node arenas, 11.8 GB of tree), through
    size-independent properties: visit distributions sum to 1 over the legal codes only, root visits = iterations + 1
    (+ Q14 extras), every selection is counted, children match the legal-play counts of the roots, mean leaf depth > 2"""
    import diee_amd
    iters = 1600
    cfg = diee_amd.MctsConfig(iterations=iters, c=2.0, round_limit=400, dir_alpha=0.3, dir_eps=0.25)
    roots = oracle.random_walk_states(99, 40)
    roots = roots[np.linspace(0, len(roots) - 1, 1024).astype(int)]
    r = eng.alpha_mcts_parallel(roots, cfg, 0xD1EE0001, 0)
    st = r["stats"]
    assert st["selections"] == iters * 1024 and st["illegal_decodes"] == 0
    assert st["nn_evals"] == (iters + 1) * 1024
    assert st["expansions"] + st["terminal_hits"] <= (iters + 1) * 1024
    assert st["depth_sum"] / st["selections"] > 2.0                  # a deep tree, not a wide root
    plays, counts = eng.get_valid_moves(roots, 160)
    assert (r["n_children"] == counts).all()                         # root children = legal plays of the root (A2 = M6)
    has = counts > 0
    assert np.allclose(r["probs"][has].sum(1), 1.0, atol=1e-5) and np.isnan(r["probs"][~has]).all()
    idx_s, idx_p = np.nonzero(np.arange(160)[None, :] < counts[:, None])
    legal = np.zeros((1024, 1352), dtype=bool)
    legal[idx_s, eng.encode(roots[idx_s], plays[idx_s, idx_p])] = True
    assert (np.nan_to_num(r["probs"])[~legal] == 0).all()            # visits only on legal codes
    rv = r["root_visits"]
    assert (rv >= iters + 1).all() and (rv[1:] <= 2 * iters + 1).all()
    # every visit below the root was a selection: sum of child visits = root visits - 1 is implied by rows summing to 1;
    # check the distribution is not degenerate: the most visited child holds less than everything when there is a choice
    multi = counts > 1
    assert (np.nan_to_num(r["probs"])[multi].max(1) < 1.0).all()
