"""Pins the CPU oracle against the reference's own test vectors (tests/golden/*.json are
hand-transcribed data from /root/reference/tests/*.rs) and the hand-derived known answers of
SURVEY.md section 8(c)."""
import json
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(__file__), "golden")
BG = json.load(open(os.path.join(G, "backgammon_cases.json")))
ENC = json.load(open(os.path.join(G, "encoding_cases.json")))
TTT = json.load(open(os.path.join(G, "tictactoe_cases.json")))


def mk(orc, b, **kw):
    return orc.bg_state(pts=b["pts"], bar=tuple(b["bar"]), off=tuple(b["off"]), **kw)


def as_board(b):
    return (list(b["pts"]), tuple(b["bar"]), tuple(b["off"]))


def test_initial_state(oracle):
    assert oracle.board_tuple(oracle.bg_new()) == as_board(BG["initial_state"]["expect"])
    s = oracle.bg_new()
    assert int(s["player"]) == -1 and tuple(s["roll"]) == (0, 0) and int(s["second"]) == 0


@pytest.mark.parametrize("case", BG["next_state"], ids=lambda c: c["ref"])
def test_next_state(oracle, case):
    s = mk(oracle, case["board"])
    t = oracle.bg_next_state(s, case["moves"], case["player"])
    assert oracle.board_tuple(t) == as_board(case["expect"])


@pytest.mark.parametrize("case", BG["normal_moves"], ids=lambda c: c["ref"])
def test_normal_moves(oracle, case):
    s = mk(oracle, case["board"])
    tree = oracle.tree_to_nested(oracle.bg_normal_moves(case["dice"], s, case["player"]))
    if case.get("empty"):
        assert tree == []
    if "len" in case:
        assert len(tree) == case["len"]
    if "root0_value" in case:
        assert tree[0][:2] == case["root0_value"]
    if "root0_child0_value" in case:
        assert tree[0][2][0][:2] == case["root0_child0_value"]
    if "root0_tree" in case:
        assert tree[0] == case["root0_tree"]
    for t in case.get("contains", []):
        assert t in tree


@pytest.mark.parametrize("case", BG["is_collectible"], ids=lambda c: c["ref"])
def test_is_collectible(oracle, case):
    s = mk(oracle, case["board"])
    for p in (-1, 1):
        assert bool(oracle.lib().or_bg_is_collectible(s.ctypes.data, p)) == case["expect"][str(p)]


@pytest.mark.parametrize("case", BG["check_win"], ids=lambda c: c["ref"])
def test_check_win(oracle, case):
    s = mk(oracle, case["board"])
    # check_win(state, player), backgammon_logic.rs:519-525
    assert (int(s["off"][0]) == 15) == case["expect"]["-1"]
    assert (int(s["off"][1]) == 15) == case["expect"]["1"]
    w = oracle.bg_check_winner(s)
    assert (w is not None) == (case["expect"]["-1"] or case["expect"]["1"])
    if case["expect"]["-1"]:
        assert w == -1  # p-1 is checked first, :527-534


def L(seqs):
    return [[tuple(m) for m in sq] for sq in seqs]


@pytest.mark.parametrize("case", BG["extract_sequences_node0"], ids=lambda c: c["ref"])
def test_extract_sequences_node(oracle, case):
    s = mk(oracle, case["board"])
    t = oracle.bg_normal_moves(case["dice"], s, case["player"])
    # first root's subtree = nodes until the next depth-0 node
    idx = [i for i, n in enumerate(t) if n["depth"] == 0]
    end = idx[1] if len(idx) > 1 else len(t)
    assert oracle.bg_extract_sequences(t[:end]) == L(case["expect"])


@pytest.mark.parametrize("case", BG["extract_sequences_list"], ids=lambda c: c["ref"])
def test_extract_sequences_list(oracle, case):
    s = mk(oracle, case["board"])
    t = oracle.bg_normal_moves(case["dice"], s, case["player"])
    if "min_roots" in case:
        assert sum(1 for n in t if n["depth"] == 0) >= case["min_roots"]
    assert oracle.bg_extract_sequences(t) == L(case["expect"])


@pytest.mark.parametrize("case", BG["remove_duplicate_states"], ids=lambda c: c["ref"])
def test_remove_duplicate_states(oracle, case):
    s = mk(oracle, case["board"])
    assert oracle.bg_remove_duplicate_states(s, L(case["sequences"]), case["player"]) == L(case["expect"])


@pytest.mark.parametrize("case", BG["entry_moves"], ids=lambda c: c["ref"])
def test_entry_moves(oracle, case):
    s = mk(oracle, case["board"])
    tree = oracle.tree_to_nested(oracle.bg_entry_moves(case["dice"], s, case["player"]))
    assert [t[:2] for t in tree] == case["roots"]


@pytest.mark.parametrize("case", BG["valid_moves"], ids=lambda c: c["ref"])
def test_valid_moves(oracle, case):
    s = mk(oracle, case["board"], roll=tuple(case["roll"]), player=case["player"])
    assert oracle.bg_valid_moves(s) == L(case["expect"])


def test_stale_reference_case_follows_the_source(oracle):
    """tests/backgammon_test.rs:917-925 is stale: the source as written yields a 2-move play."""
    case = BG["stale"][0]
    s = mk(oracle, case["board"], roll=tuple(case["roll"]), player=case["player"])
    assert oracle.bg_valid_moves(s) == L(case["source_as_written_yields"])
    # Q1: same player moves again with the same roll
    t = oracle.bg_apply_move(s, [(20, 19), (19, 18)], 3, 4)
    assert int(t["player"]) == -1 and tuple(t["roll"]) == (1, 1) and int(t["second"]) == 1
    u = oracle.bg_apply_move(t, [(18, 17), (17, 16)], 3, 4)
    assert int(u["player"]) == 1 and tuple(u["roll"]) == (3, 4) and int(u["second"]) == 0


# ------------------------------------------------------------------ codec
def enc_state(orc, roll, player):
    return orc.bg_state(pts=[0] * 24, roll=tuple(roll), player=player)


@pytest.mark.parametrize("roll,player,actions", ENC["round_trip"], ids=lambda v: str(v))
def test_encoding_round_trip(oracle, roll, player, actions):
    s = enc_state(oracle, roll, player)
    acts = [tuple(a) for a in actions]
    code = oracle.bg_encode(s, acts)
    assert 0 <= code < 1352
    assert oracle.bg_decode(s, code) == acts


@pytest.mark.parametrize("roll,player,actions,code", ENC["known_codes"], ids=lambda v: str(v))
def test_known_codes(oracle, roll, player, actions, code):
    s = enc_state(oracle, roll, player)
    assert oracle.bg_encode(s, [tuple(a) for a in actions]) == code


@pytest.mark.parametrize("case", ENC["known_valid_moves"], ids=lambda c: c["why"][:40])
def test_known_valid_moves(oracle, case):
    if case.get("board") == "initial":
        s = oracle.bg_new()
        s["roll"] = case["roll"]; s["player"] = case["player"]
    else:
        pts = [0] * 24
        for k, v in case["points"].items():
            pts[int(k)] = v
        s = oracle.bg_state(pts=pts, roll=tuple(case["roll"]), player=case["player"])
    plays, raw = oracle.bg_valid_moves(s, with_raw_count=True)
    assert plays == L(case["expect"])
    if "n_before_dedup" in case:
        assert raw == case["n_before_dedup"]
    if "first_code" in case:
        assert oracle.bg_encode(s, plays[0]) == case["first_code"]
        assert oracle.bg_encode(s, plays[-1]) == case["last_code"]
    # inverted roll gives the same plays (tests/backgammon_test.rs:882-894)
    s2 = s.copy(); s2["roll"] = case["roll"][::-1]
    assert oracle.bg_valid_moves(s2) == plays


def test_planes_layout(oracle):
    """as_tensor, backgammon_logic.rs:198-252 (Q7): raw counts, roll in rolled order, split at point 12"""
    s = oracle.bg_new()
    s["roll"] = (2, 5); s["bar"] = (1, 3); s["off"] = (4, 6); s["second"] = 1
    p = oracle.bg_planes(s).reshape(6, 4, 6)
    assert p[0].reshape(-1).tolist() == [float(x) for x in s["pts"]]
    assert (p[1] == -1).all()
    assert (p[2].reshape(-1)[:12] == 1).all() and (p[2].reshape(-1)[12:] == 3).all()
    assert (p[3].reshape(-1)[:12] == 4).all() and (p[3].reshape(-1)[12:] == 6).all()
    assert (p[4].reshape(-1)[:12] == 2).all() and (p[4].reshape(-1)[12:] == 5).all()
    assert (p[5] == 1).all()


def test_skip_turn(oracle):
    s = oracle.bg_new(); s["roll"] = (3, 3); s["second"] = 1
    t = oracle.bg_skip_turn(s, 6, 1)
    assert int(t["player"]) == 1 and tuple(t["roll"]) == (6, 1) and int(t["second"]) == 0  # Q5


# ------------------------------------------------------------------ tic-tac-toe
def ttt_state(board=None, player=-1):
    s = np.zeros(32, dtype=np.int8)
    if board is not None:
        s[:9] = board
    s[9] = player
    return s


def test_ttt_new(oracle):
    s = np.ones(32, dtype=np.int8)
    oracle.lib().or_ttt_new(s.ctypes.data)
    assert int(s[:9].sum()) == TTT["new"]["board_sum"] and int(s[9]) == TTT["new"]["player"]


def test_ttt_apply(oracle):
    for c in TTT["apply_move"]:
        s = ttt_state()
        oracle.lib().or_ttt_apply_move(s.ctypes.data, c["move"])
        if "expect_player" in c:
            assert int(s[9]) == c["expect_player"]
        if "expect_cell" in c:
            assert int(s[c["expect_cell"][0]]) == c["expect_cell"][1]


@pytest.mark.parametrize("case", TTT["valid_moves"], ids=lambda c: c["ref"])
def test_ttt_valid_moves(oracle, case):
    s = ttt_state(case["board"])
    out = np.zeros(9, dtype=np.uint8)
    n = oracle.lib().or_ttt_valid_moves(s.ctypes.data, out.ctypes.data)
    assert out[:n].tolist() == case["expect"]


@pytest.mark.parametrize("case", TTT["check_winner"], ids=lambda c: c["ref"])
def test_ttt_check_winner(oracle, case):
    import ctypes as C
    s = ttt_state(case["board"])
    w = C.c_int(7)
    r = oracle.lib().or_ttt_check_winner(s.ctypes.data, C.byref(w))
    assert (w.value if r else None) == case["expect"]
