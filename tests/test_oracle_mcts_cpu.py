"""CPU tests of the oracle's MCTS / self-play restatement and of the deterministic helpers: the two
invariants of the reference's tests/mcts_test.rs, the selection / backprop semantics of SURVEY section
3.3, the quirk switches, BASELINE config 1 (tic-tac-toe plumbing) and distribution checks for the
unseeded draws (parity there is distributional only)."""
import ctypes as C

import numpy as np
import pytest


def cfg(orc, iters, **kw):
    d = dict(iterations=iters, c=2.0, round_limit=400, dir_alpha=0.3, dir_eps=0.25); d.update(kw)
    return orc.MctsCfg(**d)


def test_philox_known_answers(oracle):
    # Random123 kat_vectors: philox4x32-10
    assert [hex(x) for x in oracle.philox([0, 0], [0, 0, 0, 0])] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    assert [hex(x) for x in oracle.philox([0xffffffff] * 2, [0xffffffff] * 4)] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    assert [hex(x) for x in oracle.philox([0xa4093822, 0x299f31d0], [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344])] == \
        ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_dice_distribution(oracle):
    """roll_die (backgammon_logic.rs:100-104): two iid uniform 1..6 -- chi^2 on 36 cells"""
    n = 36000
    d = np.array([oracle.dice(123, i, 0, 5, 0) for i in range(n)])
    assert d.min() == 1 and d.max() == 6
    cells = np.bincount((d[:, 0] - 1) * 6 + d[:, 1] - 1, minlength=36)
    chi2 = ((cells - n / 36) ** 2 / (n / 36)).sum()
    assert chi2 < 70      # 35 dof, p ~ 4e-4


def test_dirichlet_moments(oracle):
    """noise.rs:27-34 Dirichlet(0.3 * 1_1352): sums to 1, mean 1/K, var = (K-1)/(K^2 (K a + 1))"""
    K, a = 1352, 0.3
    xs = np.stack([oracle.dirichlet(9, s, a, K) for s in range(40)]).astype(np.float64)
    assert np.allclose(xs.sum(1), 1.0, atol=1e-5) and (xs >= 0).all()
    assert abs(xs.mean() - 1 / K) < 1e-9
    var = (K - 1) / (K * K * (K * a + 1))
    assert 0.85 * var < xs.var() < 1.15 * var
    assert (oracle.dirichlet(9, 0, a, K) == oracle.dirichlet(9, 0, a, K)).all()
    assert (oracle.dirichlet(9, 0, a, K) != oracle.dirichlet(9, 1, a, K)).any()


def test_det_pow_accuracy(oracle):
    x = np.concatenate([np.linspace(1e-6, 1, 4001), [0.0, 1.0, 1e-30, 0.5, 1 / 3]]).astype(np.float32)
    for y in (0.8, 1.0, 2.0, 0.1):
        got = np.array([oracle.det_powf(float(v), y) for v in x], dtype=np.float32)
        ref = (x.astype(np.float64) ** np.float64(np.float32(y))).astype(np.float32)
        ulp = np.abs(got.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64))
        assert ulp.max() <= 1, (y, ulp.max())
    assert oracle.det_powf(0.0, 0.8) == 0.0 and oracle.det_powf(1.0, 0.8) == 1.0


# ---------------------------------------------------------------- reference tests/mcts_test.rs
def test_masked_renormalised_policy_rows_sum_to_one(oracle):
    """tests/mcts_test.rs:16-33: 10 fresh TicTacToe roots, random policy [10,9]: expanded priors sum to 1"""
    rng = np.random.default_rng(0)
    pol = rng.random((10, 9)).astype(np.float32)
    states = np.zeros((10, 32), dtype=np.uint8); states[:, 9] = 0xFF      # TicTacToe::new(): player -1

    def fn(st):
        return pol[:len(st)], np.zeros(len(st), dtype=np.float32)
    ev = oracle.make_eval(fn, 9)
    roots, probs, stats, _ = oracle.alpha_mcts_parallel(0, states, cfg(oracle, 0, dir_eps=0.0), ev, None, 1, 0,
                                                        np.arange(10), np.zeros(10), 1)
    for i, r in enumerate(roots):
        pri = np.array([c[3] for c in r["children"]], dtype=np.float64)
        assert len(pri) == 9 and abs(pri.sum() - 1.0) < 1e-5
        assert np.allclose(pri, pol[i] / pol[i].sum(), rtol=1e-5)


def test_get_prob_tensor_rows_sum_to_one(oracle):
    """tests/mcts_test.rs:40-60: visit-count rows sum to 1"""
    states = np.zeros((3, 32), dtype=np.uint8); states[:, 9] = 0xFF
    roots, probs, stats, _ = oracle.alpha_mcts_parallel(0, states, cfg(oracle, 30), oracle.hash_eval_fn(),
                                                        oracle.game(0), 1, 0, np.arange(3), np.zeros(3), 1)
    assert probs.shape == (3, 9)
    assert np.allclose(probs.sum(1), 1.0, atol=1e-5)
    for r in roots:     # Q13: root visits start at 1; every iteration backpropagates once through the root
        assert r["visits"] == 31.0
        assert sum(c[1] for c in r["children"]) == 30.0


# ---------------------------------------------------------------- selection / backprop semantics
def make_store(orc, children):
    """a root (visits given) with leaf children [(visits, value, prior)]"""
    L = orc.lib()
    st = orc.Store(); L.or_store_init(C.byref(st))
    nodes = (orc.Node * (1 + len(children)))()
    nodes[0].parent = -1; nodes[0].first_child = 1; nodes[0].n_children = len(children); nodes[0].visits = 16.0
    for i, (vis, val, pr) in enumerate(children):
        n = nodes[1 + i]
        n.parent = 0; n.first_child = -1; n.n_children = 0; n.visits = vis; n.value = val; n.policy = pr
    st.nodes = C.cast(nodes, C.POINTER(orc.Node)); st.n = 1 + len(children); st.cap = st.n
    return st, nodes


def test_puct_formula_and_last_of_equal_maxima(oracle):
    L = oracle.lib()
    st, keep = make_store(oracle, [(0, 0, 0.25), (0, 0, 0.5), (0, 0, 0.5), (3, 1.5, 0.1)])
    c = np.float32(2.0)
    # node.rs:98-112: q + (c * (sqrt(N_parent) / (n + 1))) * p
    u1 = L.or_alpha_ucb(C.byref(st), 2, c)
    assert u1 == np.float32(0) + (c * (np.sqrt(np.float32(16)) / np.float32(1))) * np.float32(0.5)
    u3 = L.or_alpha_ucb(C.byref(st), 4, c)
    assert u3 == np.float32(1.5) / np.float32(3) + (c * (np.float32(4) / np.float32(4))) * np.float32(0.1)
    d = C.c_int(0)
    assert L.or_select_leaf(C.byref(st), 0, c, C.byref(d)) == 3      # children 1 and 2 tie: max_by keeps the LAST (Q11)
    assert d.value == 1


def test_nan_scores_compare_equal(oracle):
    L = oracle.lib()
    nan = float("nan")
    # [5, NaN, 3, 7, 2] -> the fold restarts after the NaN: 7 wins;  trailing NaN wins
    st, keep = make_store(oracle, [(0, 0, 5 / 8), (0, 0, nan), (0, 0, 3 / 8), (0, 0, 7 / 8), (0, 0, 2 / 8)])
    assert L.or_select_leaf(C.byref(st), 0, np.float32(2.0), None) == 4
    st, keep = make_store(oracle, [(0, 0, 5 / 8), (0, 0, 0.1), (0, 0, nan)])
    assert L.or_select_leaf(C.byref(st), 0, np.float32(2.0), None) == 3


def test_backprop_same_sign_every_level(oracle):
    L = oracle.lib()
    st, nodes = make_store(oracle, [(0, 0, 1.0)])
    L.or_backpropagate(C.byref(st), 1, np.float32(-0.75))       # simple_mcts.rs:96-103 (Q12)
    assert nodes[1].visits == 1 and nodes[1].value == -0.75
    assert nodes[0].visits == 17 and nodes[0].value == -0.75


# ---------------------------------------------------------------- self-play driver
def test_tictactoe_config1_plumbing(oracle):
    """BASELINE config 1: Tic-Tac-Toe, 1 self-play game, iterations=50, CPU path"""
    r = oracle.self_play_parallel(0, 1, cfg(oracle, 50), 1.25, 7, oracle.hash_eval_fn(), oracle.game(0))
    n = len(r["outcome"])
    assert 5 <= n <= 9 and r["plies"][0] == n
    assert r["ps"].shape == (n, 9) and r["state"].shape == (n, 27)
    assert set(np.unique(r["outcome"])) <= {-1, 0, 1}
    assert (r["state"].reshape(n, 3, 9).sum(1) == 1).all()        # one-hot planes (tictactoe/mod.rs:83-94)
    assert (r["state"].reshape(n, 3, 9)[0, 1] == 1).all()         # first recorded state is the empty board
    w = int(r["winners"][0])
    if w != 0:      # winner's states are labelled +1, loser's -1 (alpha_parallel.rs:216-217); -1 moves first
        movers = np.array([-1 if i % 2 == 0 else 1 for i in range(n)])
        assert (r["outcome"] == np.where(movers == w, 1, -1)).all()


def test_backgammon_self_play_records(oracle):
    c = cfg(oracle, 12)
    r = oracle.self_play_parallel(1, 6, c, 1.25, 42, oracle.hash_eval_fn(), oracle.game(1))
    assert (r["winners"] != 0).all() and r["stats"]["illegal_decodes"] == 0 and r["stats"]["code_collisions"] == 0
    assert len(r["outcome"]) == len(r["ps"]) == len(r["state"]) > 0
    assert set(np.unique(r["outcome"])) <= {-1, 1}
    # Q17: ps = (visits / sum)^(1/T), NOT renormalised
    assert np.allclose((r["ps"].astype(np.float64) ** 1.25).sum(1), 1.0, atol=1e-5)
    # label = +1 iff the mover of the recorded state is the game's winner
    pl = r["state"].reshape(-1, 6, 24)[:, 1, 0]
    assert (r["outcome"] == np.where(pl == r["winners"][r["game"]], 1, -1)).all()
    # deterministic given the seed; different seed -> different games
    r2 = oracle.self_play_parallel(1, 6, c, 1.25, 42, oracle.hash_eval_fn(), oracle.game(1))
    assert r2["ps"].tobytes() == r["ps"].tobytes()
    r3 = oracle.self_play_parallel(1, 6, c, 1.25, 43, oracle.hash_eval_fn(), oracle.game(1))
    assert r3["state"].tobytes() != r["state"].tobytes()


def test_round_limit_flush_quirk(oracle):
    """Q18/Q19: at the round limit the memory is flushed with outcome 0 and the game still plays that move"""
    c = cfg(oracle, 6, round_limit=10)
    r = oracle.self_play_parallel(1, 4, c, 1.25, 5, oracle.hash_eval_fn(), oracle.game(1), ref_quirks=1)
    assert (r["outcome"] == 0).all()
    assert r["steps"] == 11 and (r["plies"] == 11).all()           # rounds 0..9 recorded, the 11th move is played and lost
    assert (np.bincount(r["game"], minlength=4) <= 10).all()


def test_sharding_reproduces_the_unsharded_batch(oracle):
    """games are independent: with the cross-game quirks off, two shards keyed by
    first_game_id equal one batch of twice the size, game by game"""
    c = cfg(oracle, 8)
    full = oracle.self_play_parallel(1, 6, c, 1.25, 11, oracle.hash_eval_fn(), oracle.game(1), ref_quirks=0)
    a = oracle.self_play_parallel(1, 3, c, 1.25, 11, oracle.hash_eval_fn(), oracle.game(1), ref_quirks=0, first_game_id=0)
    b = oracle.self_play_parallel(1, 3, c, 1.25, 11, oracle.hash_eval_fn(), oracle.game(1), ref_quirks=0, first_game_id=3)
    for g in range(6):
        part = a if g < 3 else b
        for key in ("ps", "state", "outcome"):
            assert part[key][part["game"] == g].tobytes() == full[key][full["game"] == g].tobytes(), (g, key)


def test_self_play_multi_equals_sequential_calls(oracle):
    """K batches in lockstep with one merged evaluator call per search phase (what the engine's pipelined self-play
    does): with an evaluator that is a pure function of the state, every batch's records, statistics and length equal
    its own self_play_parallel call byte for byte -- quirks on (Q14 couples the games of ONE batch only) and off"""
    cfg = oracle.MctsCfg(iterations=10, c=2.0, round_limit=50, dir_alpha=0.3, dir_eps=0.25)
    batches = [(5, 0, 11), (3, 40, 12), (1, 7, 13), (6, 100, 11)]
    for quirks in (1, 0):
        multi, total = oracle.self_play_multi(1, batches, cfg, 1.25, oracle.hash_eval_fn(), oracle.game(1), ref_quirks=quirks)
        longest = 0
        for (n, first, seed), m in zip(batches, multi):
            ref = oracle.self_play_parallel(1, n, cfg, 1.25, seed, oracle.hash_eval_fn(), oracle.game(1),
                                            ref_quirks=quirks, first_game_id=first)
            assert m["steps"] == ref["steps"]
            longest = max(longest, ref["steps"])
            assert (m["game"] == ref["game"]).all() and (m["outcome"] == ref["outcome"]).all()
            assert m["ps"].tobytes() == ref["ps"].tobytes() and m["state"].tobytes() == ref["state"].tobytes()
            assert m["stats"] == ref["stats"]
        assert total == longest


def test_self_play_multi_merges_rows_in_batch_order(oracle):
    """the merged evaluator call holds the live games of batch 0, then batch 1, ...: row counts per phase shrink as
    games retire and never exceed the total number of games"""
    import numpy as np
    cfg = oracle.MctsCfg(iterations=4, c=2.0, round_limit=30, dir_alpha=0.3, dir_eps=0.25)
    sizes = []
    base = oracle.hash_eval_fn()

    def fn(states_u8):
        sizes.append(len(states_u8))
        n = len(states_u8)
        pol = np.zeros((n, 1352), dtype=np.float32); val = np.zeros(n, dtype=np.float32)
        st = np.ascontiguousarray(states_u8)
        base(oracle.game(1), st.ctypes.data, n, pol.ctypes.data_as(oracle.C.POINTER(oracle.C.c_float)),
             val.ctypes.data_as(oracle.C.POINTER(oracle.C.c_float)))
        return pol, val
    ev = oracle.make_eval(fn, 1352)
    multi, total = oracle.self_play_multi(1, [(4, 0, 5), (4, 10, 6)], cfg, 1.25, ev, None, ref_quirks=1)
    assert sizes[0] == 8 and max(sizes) == 8 and sizes[-1] >= 1
    assert all(a >= b for a, b in zip(sizes, sizes[1:]))             # constant within a move-step, shrinking across them
    ref = oracle.self_play_parallel(1, 4, cfg, 1.25, 6, oracle.hash_eval_fn(), oracle.game(1), ref_quirks=1, first_game_id=10)
    assert multi[1]["ps"].tobytes() == ref["ps"].tobytes()
