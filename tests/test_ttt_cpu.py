"""BASELINE.json configs[0] -- "Tic-Tac-Toe, 1 self-play game, iterations=50, random-init net, CPU reference path
(plumbing, no GPU)" -- on the PRODUCT: diee_create(.., DIEE_GAME_TTT, ..) runs rules, the 64 x 4 ResNet, the batched
search and the self-play driver in C++ on the host (die-e_amd/csrc/ttt_host.cpp; no GPU, no oracle).  Held to the
reference's own tests (tests/golden/tictactoe_cases.json = tests/tictactoe_test.rs + tests/mcts_test.rs as data), to a
PyTorch fp32 restatement of the network, and BIT-EXACTLY to the test oracle's restatement of alpha_mcts_parallel /
self_play_parallel driven by the product's own network."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import diee_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = json.load(open(os.path.join(ROOT, "tests", "golden", "tictactoe_cases.json")))


@pytest.fixture(scope="module")
def eng():
    e = diee_amd.Engine(0, diee_amd.GAME_TTT)          # no GPU in this test tier: the tic-tac-toe ctx does not need one
    e.load_weights(diee_amd.random_weights(0, diee_amd.GAME_TTT))
    yield e
    e.close()


def state(board, player=-1):
    s = diee_amd.ttt_new()
    s["board"] = board; s["player"] = player
    return s


def test_rules_match_the_references_own_tests():
    s = diee_amd.ttt_new()
    assert int(np.sum(s["board"])) == CASES["new"]["board_sum"] and int(s["player"]) == CASES["new"]["player"]
    c = CASES["apply_move"]
    assert int(diee_amd.ttt_apply_move(s, c[0]["move"])["player"]) == c[0]["expect_player"]
    cell, val = c[1]["expect_cell"]
    assert int(diee_amd.ttt_apply_move(s, c[1]["move"])["board"][cell]) == val
    for c in CASES["valid_moves"]:
        assert diee_amd.ttt_valid_moves(state(c["board"])) == c["expect"], c["ref"]
    for c in CASES["check_winner"]:
        assert diee_amd.ttt_check_winner(state(c["board"])) == c["expect"], c["ref"]
    assert diee_amd.ttt_check_winner(state([1, -1, 1, 1, -1, -1, -1, 1, 1])) == 0          # full board, no line: Some(0)
    p = diee_amd.ttt_planes(state([-1, 0, 1, 0, 0, 0, 1, -1, 0])).reshape(3, 9)              # [eq(-1), eq(0), eq(1)], mod.rs:83-94
    assert (p.sum(0) == 1).all() and p[0, 0] == 1 and p[2, 2] == 1 and p[1, 1] == 1 and p[0, 7] == 1


def test_rules_match_the_oracle_on_every_reachable_position(oracle):
    L = oracle.lib()
    seen, stack, n = set(), [diee_amd.ttt_new()], 0
    while stack:
        s = stack.pop()
        key = (bytes(s["board"]), int(s["player"]))
        if key in seen:
            continue
        seen.add(key); n += 1
        o = np.zeros(32, dtype=np.uint8); o[:9] = s["board"].view(np.uint8); o[9] = np.uint8(int(s["player"]) & 0xFF)
        mv = np.zeros(9, dtype=np.uint8)
        k = L.or_ttt_valid_moves(o.ctypes.data, mv.ctypes.data)
        assert diee_amd.ttt_valid_moves(s) == [int(x) for x in mv[:k]]
        w = C.c_int(0)
        ow = w.value if L.or_ttt_check_winner(o.ctypes.data, C.byref(w)) else None
        ow = w.value if ow is not None else None
        assert diee_amd.ttt_check_winner(s) == ow
        op = np.zeros(27, dtype=np.float32); L.or_ttt_planes(o.ctypes.data, op.ctypes.data)
        assert (diee_amd.ttt_planes(s) == op).all()
        if ow is None:
            stack += [diee_amd.ttt_apply_move(s, m) for m in diee_amd.ttt_valid_moves(s)]
    assert n == 5478                                        # every position reachable from the empty board


def test_network_matches_the_fp32_restatement(eng):
    from oracle import nn_ref
    blob = diee_amd.random_weights(0, diee_amd.GAME_TTT)
    conv = lambda co, ci: co * ci * 9 + co
    assert len(blob) == conv(64, 3) + 256 + 4 * (2 * conv(64, 64) + 512) + conv(32, 64) + 128 + 9 * 288 + 9 + conv(3, 64) + 12 + 27 + 1
    rng = np.random.default_rng(3)
    states = diee_amd.ttt_new(64)
    states["board"] = rng.integers(-1, 2, size=(64, 9)); states["player"] = rng.choice([-1, 1], size=64)
    p, v = eng.forward_t(states)
    planes = np.stack([diee_amd.ttt_planes(s) for s in states])
    for b in (blob, _blob_with_batchnorm_statistics(blob)):
        e = diee_amd.Engine(0, diee_amd.GAME_TTT); e.load_weights(b)
        p, v = e.forward_t(states)
        rp, rv, _ = nn_ref.forward_t(nn_ref.parse(b, 64, 4, 9, 3, 9), planes, shape=(3, 3, 3))
        assert np.abs(p - rp).max() < 2e-6 and np.abs(v - rv).max() < 2e-6       # fp32 both sides: summation order only
        assert np.allclose(p.sum(1), 1, atol=1e-6)
        e.close()


def _blob_with_batchnorm_statistics(blob):
    """non-trivial running statistics / beta in every BatchNorm (exercises the folding)"""
    b = blob.copy(); rng = np.random.default_rng(9)
    conv = lambda co, ci: co * ci * 9 + co
    o = conv(64, 3)
    def bn(o, c):
        b[o:o + c] = rng.uniform(0.5, 1.5, c); b[o + c:o + 2 * c] = rng.normal(0, 0.2, c)
        b[o + 2 * c:o + 3 * c] = rng.normal(0, 0.3, c); b[o + 3 * c:o + 4 * c] = rng.uniform(0.3, 2.0, c)
        return o + 4 * c
    o = bn(o, 64)
    for _ in range(4):
        o += 2 * conv(64, 64); o = bn(o, 64); o = bn(o, 64)
    o += conv(32, 64); o = bn(o, 32); o += 9 * 288 + 9
    o += conv(3, 64); o = bn(o, 3)
    assert o + 27 + 1 == len(b)
    return b


def _oracle_eval(eng, oracle):
    def fn(states_u8):
        return eng.forward_t(states_u8.view(diee_amd.TTT_STATE).reshape(-1))
    return oracle.make_eval(fn, 9)


def cfgs(oracle, iters, **kw):
    d = dict(iterations=iters, c=2.0, round_limit=400, dir_alpha=0.3, dir_eps=0.25); d.update(kw)
    return oracle.MctsCfg(**d), diee_amd.MctsConfig(**d)


@pytest.mark.parametrize("n,iters,quirks,kw", [(1, 50, 1, {}), (1, 50, 0, {}), (8, 30, 1, {}), (8, 30, 0, {}), (5, 12, 1, {"round_limit": 3}),
                                                 (5, 12, 0, {"round_limit": 3})],
                         ids=["config1_1x50", "config1_1x50_clean", "8x30", "8x30_clean", "round_limit_Q18", "round_limit_clean"])
def test_self_play_bit_exact_vs_oracle(eng, oracle, n, iters, quirks, kw):
    """BASELINE configs[0] and friends: records (ps, state, outcome, game, order) and counters identical to the oracle's
    self_play_parallel (alpha_parallel.rs:101-231) driven by the product's own network"""
    ocfg, gcfg = cfgs(oracle, iters, **kw)
    ref = oracle.self_play_parallel(0, n, ocfg, 1.25, 77, _oracle_eval(eng, oracle), None, ref_quirks=quirks, first_game_id=40)
    out = eng.self_play_parallel(n, gcfg, 1.25, 77, ref_quirks=bool(quirks), first_game_id=40)
    assert out["stats"]["move_steps"] == ref["steps"] and len(out["outcome"]) == len(ref["outcome"]) > 0
    assert (out["game"] == ref["game"]).all() and (out["outcome"] == ref["outcome"]).all()
    assert out["state"].tobytes() == ref["state"].tobytes() and out["ps"].tobytes() == ref["ps"].tobytes()
    for key in ("nn_evals", "expansions", "children", "terminal_hits", "depth_sum", "selections", "max_children", "illegal_decodes"):
        assert out["stats"][key] == ref["stats"][key], key
    assert out["stats"]["games"] == n and out["stats"]["plies"] == int(ref["plies"].sum()) and out["stats"]["illegal_decodes"] == 0
    if not kw:
        assert all(5 <= c <= 9 for c in np.bincount(out["game"] - 40, minlength=n))       # a game of tic-tac-toe takes 5..9 plies
        assert (out["state"].reshape(-1, 3, 9).sum(1) == 1).all()                        # one-hot planes
        assert np.allclose((out["ps"].astype(np.float64) ** 1.25).sum(1), 1.0, atol=1e-5)   # Q17: (visits / sum)^(1/T), not renormalised


@pytest.mark.parametrize("quirks", [1, 0])
def test_mcts_batch_bit_exact_vs_oracle(eng, oracle, quirks):
    """alpha_mcts_parallel + get_prob_tensor_parallel on mixed positions incl. one move from the end (terminal leaves, Q14)"""
    boards = [[0] * 9, [-1, 1, 0, 0, -1, 0, 0, 0, 1], [-1, -1, 0, 1, 1, 0, 0, 0, 0], [1, -1, 1, -1, -1, 1, 0, 1, 0], [-1, 1, -1, -1, 1, 1, 0, 0, 0]]
    states = diee_amd.ttt_new(len(boards))
    for i, b in enumerate(boards):
        states[i]["board"] = b
        states[i]["player"] = -1 if sum(1 for x in b if x) % 2 == 0 else 1
    ocfg, gcfg = cfgs(oracle, 60)
    n = len(states)
    roots, probs, ostats, _ = oracle.alpha_mcts_parallel(0, states.view(np.uint8).reshape(n, 32), ocfg, _oracle_eval(eng, oracle), None, 5, 2,
                                                         np.arange(n, dtype=np.uint32), np.zeros(n, dtype=np.uint32), quirks)
    r = eng.alpha_mcts_parallel(states, gcfg, 5, 2, ref_quirks=bool(quirks))
    assert r["probs"].tobytes() == probs.tobytes()
    assert (r["root_visits"] == np.array([x["visits"] for x in roots], dtype=np.float32)).all()
    assert (r["n_children"] == np.array([len(x["children"]) for x in roots])).all()
    os_ = ostats.as_dict()
    for key in ("nn_evals", "expansions", "children", "terminal_hits", "depth_sum", "selections", "max_children"):
        assert r["stats"][key] == os_[key], key
    assert os_["terminal_hits"] > 0
    # tests/mcts_test.rs:40-60 as a property: visit distributions of roots with children sum to 1
    assert np.allclose(r["probs"].sum(1), 1, rtol=1e-5)


def test_backgammon_only_entry_points_refuse_a_tictactoe_ctx(eng):
    L = diee_amd.load_library()
    s = np.zeros(1, dtype=diee_amd.BG_STATE); out = np.zeros(144, dtype=np.float32)
    assert L.diee_bg_planes(eng._h, s.ctypes.data, 1, out.ctypes.data) == diee_amd.ERR_UNSUPPORTED
    assert b"backgammon ctx" in L.diee_last_error(eng._h)
    assert L.diee_set_invariant_nn(eng._h, 1) == diee_amd.ERR_UNSUPPORTED
    fresh = diee_amd.Engine(0, diee_amd.GAME_TTT)
    with pytest.raises(diee_amd.DieeError) as ei:
        fresh.forward_t(diee_amd.ttt_new(1))
    assert ei.value.status == diee_amd.ERR_NO_WEIGHTS
    with pytest.raises(diee_amd.DieeError):
        fresh.load_weights(np.zeros(10, dtype=np.float32))
    fresh.close()


def test_cli_learn_train_play_for_tictactoe(tmp_path, monkeypatch, capsys):
    """`die-e -g tic-tac-toe learn | train | play` (main.rs:112-114: the same driver for either game) on the host engine:
    self-play batches -> data files -> epochs of training -> model -> arena against the best model"""
    import importlib
    cli = importlib.import_module("die-e_amd.cli")
    conf = tmp_path / "ttt.toml"
    conf.write_text("temperature = 1.25\nlearn_iterations = 2\nnum_epochs = 2\ntraining_batch_size = 16\nself_play_iterations = 2\n"
                    "num_self_play_batches = 6\niterations = 12\nexploration_const = 2.0\nsimulate_round_limit = 400\n"
                    "dirichlet_alpha = 0.3\ndirichlet_epsilon = 0.25\nwd = 0.0001\nlr = 0.001\n")
    monkeypatch.chdir(tmp_path)
    assert cli.main(["-c", str(conf), "-g", "tic-tac-toe", "learn"]) == 0
    out = capsys.readouterr().out
    assert "Iteration 1 saved successfully" in out and "saved-as-best" in out or "No best model was found" in out
    runs = os.listdir(tmp_path / "data" / "tictactoe")
    assert len(runs) == 1
    sp = tmp_path / "data" / "tictactoe" / runs[0] / "lrn-0" / "sp-1"
    ps, st, oc = np.load(sp / "ps.npy"), np.load(sp / "states.npy"), np.load(sp / "outcomes.npy")
    assert ps.shape[1] == 9 and st.shape[1:] == (3, 3, 3) and len(ps) == len(st) == len(oc) >= 2 * 6 * 5      # cumulative memory (Q20)
    assert (tmp_path / "models" / "tictactoe" / "model_1.npy").exists() and (tmp_path / "models" / "tictactoe" / "best_model.npy").exists()
    rid = runs[0][len("run-"):]
    assert cli.main(["-c", str(conf), "-g", "tic-tac-toe", "train", "-r", rid, "-l", "0", "-o", str(tmp_path / "t.npy")]) == 0
    trained = np.load(tmp_path / "t.npy")
    assert trained.shape == (diee_amd.weights_count(diee_amd.GAME_TTT),) and np.isfinite(trained).all()
    assert cli.main(["-c", str(conf), "-g", "tic-tac-toe", "play", "-a", "model", "-m", str(tmp_path / "t.npy"), "--agent-two", "random",
                     "-o", str(tmp_path)]) == 0
    assert "Number of Games: 400" in capsys.readouterr().out


def test_arena_skips_the_turn_on_an_all_zero_probability_row(eng):
    """get_actions_for_player returns EMPTY_MOVE when the pow'ed row sums to zero, not only when the root has no children
    (versus.rs:286-293): with iterations = 0 the root is expanded but no child is ever visited -- every row is 0 / 0 -- so the
    Model player passes every turn (it used to pick cell 0 and trip the legality assert once cell 0 was taken) and the Random
    player, moving alone, wins every game"""
    import importlib
    versus = importlib.import_module("die-e_amd.versus")
    cfg = diee_amd.MctsConfig(iterations=0, c=2.0, round_limit=400, dir_alpha=0.3, dir_eps=0.25)
    r = versus.play_tictactoe(versus.Player(versus.Agent.MODEL, eng), versus.Player(versus.Agent.RANDOM), cfg, 1.25, seed=5, num_games=12)
    assert (r.wins_p1, r.wins_p2, r.draws) == (0, 12, 0)
