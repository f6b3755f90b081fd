"""ctypes binding to the CPU oracle (oracle/libdiee_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product path (die-e_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libdiee_oracle.so")

BG_ACTIONS = 1352
BG_PLANES = 144
NO_MOVE = -2

BG_STATE = np.dtype([("pts", "i1", 24), ("bar", "u1", 2), ("off", "u1", 2), ("roll", "u1", 2),
                     ("player", "i1"), ("second", "u1")])
assert BG_STATE.itemsize == 32
TREE_NODE = np.dtype([("depth", "i1"), ("from", "i1"), ("to", "i1")])
SEQ = np.dtype([("n", "i1"), ("mv", "i1", (4, 2))])
PLAY = np.dtype([("mv", "i1", 4)])


class MctsCfg(C.Structure):
    _fields_ = [("iterations", C.c_uint32), ("c", C.c_float), ("round_limit", C.c_uint32),
                ("dir_alpha", C.c_float), ("dir_eps", C.c_float)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("nn_evals", "expansions", "children", "terminal_hits",
                                           "depth_sum", "selections", "code_collisions",
                                           "illegal_decodes", "max_children")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class Node(C.Structure):
    _fields_ = [("state", C.c_uint8 * 32), ("parent", C.c_int32), ("first_child", C.c_int32),
                ("n_children", C.c_int32), ("visits", C.c_float), ("value", C.c_float),
                ("policy", C.c_float), ("action", C.c_int32), ("drained", C.c_uint8)]


class Store(C.Structure):
    _fields_ = [("nodes", C.POINTER(Node)), ("n", C.c_int), ("cap", C.c_int)]


class Fragments(C.Structure):
    _fields_ = [("n", C.c_int32), ("outcome", C.POINTER(C.c_int8)), ("ps", C.POINTER(C.c_float)),
                ("state", C.POINTER(C.c_float)), ("game", C.POINTER(C.c_uint32))]


EVAL_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float))


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("diee_oracle.c", "diee_oracle.h")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in src if os.path.exists(s))):
        return _SO
    subprocess.check_call(["make", "-s", "-C", _HERE, "libdiee_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        vp, i, u32, u64 = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64
        L.or_bg_new.argtypes = [vp]
        L.or_bg_next_state.argtypes = [vp, vp, i, i]
        L.or_bg_is_collectible.argtypes = [vp, i]; L.or_bg_is_collectible.restype = i
        L.or_bg_check_winner.argtypes = [vp, C.POINTER(i)]; L.or_bg_check_winner.restype = i
        for f in (L.or_bg_normal_moves, L.or_bg_entry_moves, L.or_bg_action_trees):
            f.argtypes = [vp, i, vp, i, vp, i]; f.restype = i
        L.or_bg_extract_sequences.argtypes = [vp, i, vp, i]; L.or_bg_extract_sequences.restype = i
        L.or_bg_remove_duplicate_states.argtypes = [vp, vp, i, i, vp]; L.or_bg_remove_duplicate_states.restype = i
        L.or_bg_valid_moves.argtypes = [vp, vp, i]; L.or_bg_valid_moves.restype = i
        L.or_bg_valid_moves_seq.argtypes = [vp, vp, i, C.POINTER(i)]; L.or_bg_valid_moves_seq.restype = i
        L.or_bg_apply_move.argtypes = [vp, vp, C.c_uint8, C.c_uint8]
        L.or_bg_skip_turn.argtypes = [vp, C.c_uint8, C.c_uint8]
        L.or_bg_encode.argtypes = [vp, vp]; L.or_bg_encode.restype = u32
        L.or_bg_decode.argtypes = [vp, u32, vp]
        L.or_bg_planes.argtypes = [vp, vp]
        L.or_ttt_new.argtypes = [vp]
        L.or_ttt_valid_moves.argtypes = [vp, vp]; L.or_ttt_valid_moves.restype = i
        L.or_ttt_apply_move.argtypes = [vp, C.c_uint8]
        L.or_ttt_check_winner.argtypes = [vp, C.POINTER(i)]; L.or_ttt_check_winner.restype = i
        L.or_ttt_planes.argtypes = [vp, vp]
        L.or_philox4x32.argtypes = [vp, vp, vp]
        L.or_dice.argtypes = [u64, u32, u32, u32, u32, vp, vp]
        L.or_uniform01.argtypes = [u64, u32, u32, u32, u32]; L.or_uniform01.restype = C.c_double
        L.or_det_powf.argtypes = [C.c_float, C.c_float]; L.or_det_powf.restype = C.c_float
        L.or_dirichlet.argtypes = [u64, u32, C.c_float, i, vp]
        L.or_game_by_id.argtypes = [i]; L.or_game_by_id.restype = vp
        L.or_store_init.argtypes = [vp]; L.or_store_free.argtypes = [vp]
        L.or_alpha_ucb.argtypes = [vp, i, C.c_float]; L.or_alpha_ucb.restype = C.c_float
        L.or_select_leaf.argtypes = [vp, i, C.c_float, C.POINTER(i)]; L.or_select_leaf.restype = i
        L.or_backpropagate.argtypes = [vp, i, C.c_float]
        L.or_alpha_mcts_parallel.argtypes = [vp, vp, vp, i, vp, EVAL_FN, vp, u64, u32, vp, vp, i, vp]
        L.or_get_prob_tensor_parallel.argtypes = [vp, vp, i, vp]
        L.or_self_play_parallel.argtypes = [vp, u32, u32, vp, C.c_float, u64, EVAL_FN, vp, i, u32, vp, vp, vp, vp]
        L.or_self_play_parallel.restype = i
        L.or_self_play_multi.argtypes = [vp, u32, vp, vp, vp, vp, C.c_float, EVAL_FN, vp, i, u32, vp, vp, vp]
        L.or_self_play_multi.restype = i
        L.or_free_fragments.argtypes = [vp]
        L.or_random_walk_states.argtypes = [u64, u32, u32, vp, i]; L.or_random_walk_states.restype = i
        L.or_bg_valid_moves_batch.argtypes = [vp, i, vp, i, vp]
        L.or_bg_encode_batch.argtypes = [vp, vp, i, vp]
        L.or_bg_decode_batch.argtypes = [vp, vp, i, vp]
        L.or_bg_apply_batch.argtypes = [vp, vp, vp, i]
        L.or_bg_planes_batch.argtypes = [vp, i, vp]
        _lib = L
    return _lib


# --------------------------------------------------------------------------- helpers
def bg_state(pts=None, bar=(0, 0), off=(0, 0), roll=(0, 0), player=-1, second=0):
    s = np.zeros((), dtype=BG_STATE)
    if pts is None:
        lib().or_bg_new(s.ctypes.data)
    else:
        s["pts"] = pts
    s["bar"] = bar; s["off"] = off; s["roll"] = roll; s["player"] = player; s["second"] = second
    return s


def bg_new():
    s = np.zeros((), dtype=BG_STATE)
    lib().or_bg_new(s.ctypes.data)
    return s


def board_tuple(s):
    return ([int(x) for x in s["pts"]], (int(s["bar"][0]), int(s["bar"][1])), (int(s["off"][0]), int(s["off"][1])))


def bg_next_state(s, moves, player):
    t = s.copy()
    mv = np.array(moves, dtype=np.int8).reshape(-1, 2)
    lib().or_bg_next_state(t.ctypes.data, mv.ctypes.data, len(mv), player)
    return t


def _tree(fn, dice, s, player):
    d = np.array(dice, dtype=np.uint8)
    out = np.zeros(8192, dtype=TREE_NODE)
    n = fn(d.ctypes.data, len(d), s.ctypes.data, player, out.ctypes.data, len(out))
    assert n <= len(out)
    return out[:n]


def bg_normal_moves(dice, s, player):
    return _tree(lib().or_bg_normal_moves, dice, s, player)


def bg_entry_moves(dice, s, player):
    return _tree(lib().or_bg_entry_moves, dice, s, player)


def tree_to_nested(t):
    """pre-order (depth, from, to) -> nested [[from,to,[children]], ...] like the reference's ActionNode"""
    root = []
    stack = [root]
    for nd in t:
        d = int(nd["depth"])
        del stack[d + 1:]
        node = [int(nd["from"]), int(nd["to"]), []]
        stack[d].append(node)
        stack.append(node[2])
    return root


def bg_extract_sequences(t):
    t = np.ascontiguousarray(t)
    out = np.zeros(max(len(t), 1), dtype=SEQ)
    n = lib().or_bg_extract_sequences(t.ctypes.data, len(t), out.ctypes.data, len(out))
    return [[(int(s["mv"][k][0]), int(s["mv"][k][1])) for k in range(int(s["n"]))] for s in out[:n]]


def _seqs_np(seqs):
    a = np.zeros(max(len(seqs), 1), dtype=SEQ)
    a["mv"] = NO_MOVE
    for i, sq in enumerate(seqs):
        a[i]["n"] = len(sq)
        for k, (f, t) in enumerate(sq):
            a[i]["mv"][k] = (f, t)
    return a


def bg_remove_duplicate_states(s, seqs, player):
    a = _seqs_np(seqs)
    out = np.zeros(max(len(seqs), 1), dtype=SEQ)
    n = lib().or_bg_remove_duplicate_states(s.ctypes.data, a.ctypes.data, len(seqs), player, out.ctypes.data)
    return [[(int(q["mv"][k][0]), int(q["mv"][k][1])) for k in range(int(q["n"]))] for q in out[:n]]


def bg_valid_moves(s, with_raw_count=False):
    out = np.zeros(2048, dtype=SEQ)
    raw = C.c_int(0)
    n = lib().or_bg_valid_moves_seq(s.ctypes.data, out.ctypes.data, len(out), C.byref(raw))
    assert n <= len(out)
    r = [[(int(q["mv"][k][0]), int(q["mv"][k][1])) for k in range(int(q["n"]))] for q in out[:n]]
    return (r, raw.value) if with_raw_count else r


def bg_valid_plays_np(s):
    out = np.zeros(2048, dtype=PLAY)
    n = lib().or_bg_valid_moves(s.ctypes.data, out.ctypes.data, len(out))
    return out[:n]["mv"].copy()


def play_np(actions):
    p = np.full(4, NO_MOVE, dtype=np.int8)
    for k, (f, t) in enumerate(actions):
        p[2 * k] = f; p[2 * k + 1] = t
    return p


def play_list(p):
    return [(int(p[2 * k]), int(p[2 * k + 1])) for k in range(2) if p[2 * k] != NO_MOVE]


def bg_encode(s, actions):
    p = play_np(actions)
    return int(lib().or_bg_encode(s.ctypes.data, p.ctypes.data))


def bg_decode(s, code):
    p = np.zeros(4, dtype=np.int8)
    lib().or_bg_decode(s.ctypes.data, code, p.ctypes.data)
    return play_list(p)


def bg_apply_move(s, actions, d0, d1):
    t = s.copy()
    p = play_np(actions)
    lib().or_bg_apply_move(t.ctypes.data, p.ctypes.data, d0, d1)
    return t


def bg_skip_turn(s, d0, d1):
    t = s.copy()
    lib().or_bg_skip_turn(t.ctypes.data, d0, d1)
    return t


def bg_planes(s):
    out = np.zeros(BG_PLANES, dtype=np.float32)
    lib().or_bg_planes(s.ctypes.data, out.ctypes.data)
    return out


def bg_check_winner(s):
    w = C.c_int(0)
    return w.value if lib().or_bg_check_winner(s.ctypes.data, C.byref(w)) else None


def philox(key, ctr):
    k = np.array(key, dtype=np.uint32); c = np.array(ctr, dtype=np.uint32); o = np.zeros(4, dtype=np.uint32)
    lib().or_philox4x32(k.ctypes.data, c.ctypes.data, o.ctypes.data)
    return o


def dice(seed, game, rnd, tag, ord_):
    a = C.c_uint8(0); b = C.c_uint8(0)
    lib().or_dice(seed, game, rnd, tag, ord_, C.byref(a), C.byref(b))
    return a.value, b.value


def dirichlet(seed, step, alpha, n):
    out = np.zeros(n, dtype=np.float32)
    lib().or_dirichlet(seed, step, alpha, n, out.ctypes.data)
    return out


def det_powf(x, y):
    return float(lib().or_det_powf(x, y))


# --------------------------------------------------------------------------- MCTS / self-play
def game(game_id):
    return lib().or_game_by_id(game_id)


def hash_eval_fn():
    """the oracle's cheap deterministic evaluator as an EVAL_FN (ctx must be the game pointer)"""
    return C.cast(lib().or_hash_eval, EVAL_FN)


def make_eval(py_fn, n_actions):
    """wrap py_fn(states: np.ndarray[n] of 32-byte records) -> (policy [n,A] f32, value [n] f32)"""
    def _cb(ctx, states_p, n, pol_p, val_p):
        st = np.ctypeslib.as_array(C.cast(states_p, C.POINTER(C.c_uint8)), shape=(n, 32)).copy()
        pol, val = py_fn(st)
        pol = np.ascontiguousarray(pol, dtype=np.float32); val = np.ascontiguousarray(val, dtype=np.float32)
        assert pol.shape == (n, n_actions) and val.shape == (n,)
        C.memmove(pol_p, pol.ctypes.data, pol.nbytes)
        C.memmove(val_p, val.ctypes.data, val.nbytes)
    return EVAL_FN(_cb)


def alpha_mcts_parallel(game_id, states, cfg, eval_fn, ectx, seed, step, game_ids, rounds, ref_quirks=1):
    """returns (store_nodes list-of-dicts for the first n roots' children, probs [n,A], stats)"""
    L = lib()
    g = game(game_id)
    st = Store(); L.or_store_init(C.byref(st))
    stats = Stats()
    states = np.ascontiguousarray(states)
    n = len(states)
    gi = np.ascontiguousarray(game_ids, dtype=np.uint32); rd = np.ascontiguousarray(rounds, dtype=np.uint32)
    L.or_alpha_mcts_parallel(g, C.byref(st), states.ctypes.data, n, C.byref(cfg), eval_fn, ectx, seed, step,
                             gi.ctypes.data, rd.ctypes.data, ref_quirks, C.byref(stats))
    A = 9 if game_id == 0 else BG_ACTIONS
    probs = np.zeros((n, A), dtype=np.float32)
    L.or_get_prob_tensor_parallel(g, C.byref(st), n, probs.ctypes.data)
    roots = []
    for i in range(n):
        r = st.nodes[i]
        ch = [(st.nodes[r.first_child + j].action, st.nodes[r.first_child + j].visits,
               st.nodes[r.first_child + j].value, st.nodes[r.first_child + j].policy)
              for j in range(r.n_children)]
        roots.append({"visits": r.visits, "value": r.value, "children": ch})
    n_nodes = st.n
    L.or_store_free(C.byref(st))
    return roots, probs, stats, n_nodes


def self_play_parallel(game_id, n_games, cfg, temperature, seed, eval_fn, ectx, ref_quirks=1,
                       first_game_id=0, max_steps=0):
    L = lib()
    g = game(game_id)
    fr = Fragments(); stats = Stats()
    plies = np.zeros(n_games, dtype=np.uint32); winners = np.zeros(n_games, dtype=np.int8)
    steps = L.or_self_play_parallel(g, n_games, first_game_id, C.byref(cfg), temperature, seed, eval_fn, ectx,
                                    ref_quirks, max_steps, C.byref(fr), C.byref(stats), plies.ctypes.data,
                                    winners.ctypes.data)
    A = 9 if game_id == 0 else BG_ACTIONS
    P = 27 if game_id == 0 else BG_PLANES
    n = fr.n
    out = {
        "outcome": np.ctypeslib.as_array(fr.outcome, shape=(n,)).copy() if n else np.zeros(0, np.int8),
        "ps": np.ctypeslib.as_array(fr.ps, shape=(n, A)).copy() if n else np.zeros((0, A), np.float32),
        "state": np.ctypeslib.as_array(fr.state, shape=(n, P)).copy() if n else np.zeros((0, P), np.float32),
        "game": np.ctypeslib.as_array(fr.game, shape=(n,)).copy() if n else np.zeros(0, np.uint32),
        "steps": steps, "plies": plies, "winners": winners, "stats": stats.as_dict(),
    }
    L.or_free_fragments(C.byref(fr))
    return out


def _fragments_dict(fr, game_id):
    A = 9 if game_id == 0 else BG_ACTIONS
    P = 27 if game_id == 0 else BG_PLANES
    n = fr.n
    return {
        "outcome": np.ctypeslib.as_array(fr.outcome, shape=(n,)).copy() if n else np.zeros(0, np.int8),
        "ps": np.ctypeslib.as_array(fr.ps, shape=(n, A)).copy() if n else np.zeros((0, A), np.float32),
        "state": np.ctypeslib.as_array(fr.state, shape=(n, P)).copy() if n else np.zeros((0, P), np.float32),
        "game": np.ctypeslib.as_array(fr.game, shape=(n,)).copy() if n else np.zeros(0, np.uint32),
    }


def self_play_multi(game_id, batches, cfg, temperature, eval_fn, ectx, ref_quirks=1, max_steps=0):
    """K self_play_parallel calls in lockstep with ONE merged evaluator call per search phase.
    batches = [(n_games, first_game_id, seed), ...] -> list of per-batch dicts (like self_play_parallel's)"""
    L = lib()
    g = game(game_id)
    K = len(batches)
    ng = np.array([b[0] for b in batches], dtype=np.uint32)
    fi = np.array([b[1] for b in batches], dtype=np.uint32)
    sd = np.array([b[2] for b in batches], dtype=np.uint64)
    frs = (Fragments * K)(); sts = (Stats * K)()
    steps = np.zeros(K, dtype=np.uint32)
    total = L.or_self_play_multi(g, K, ng.ctypes.data, fi.ctypes.data, sd.ctypes.data, C.byref(cfg), temperature,
                                 eval_fn, ectx, ref_quirks, max_steps, C.byref(frs), C.byref(sts), steps.ctypes.data)
    out = []
    for k in range(K):
        d = _fragments_dict(frs[k], game_id)
        d["stats"] = sts[k].as_dict(); d["steps"] = int(steps[k])
        out.append(d)
        L.or_free_fragments(C.byref(frs[k]))
    return out, total


# --------------------------------------------------------------------------- batch helpers
def random_walk_states(seed, n_games, max_plies=400, cap=None):
    cap = cap or n_games * max_plies
    out = np.zeros(cap, dtype=BG_STATE)
    n = lib().or_random_walk_states(seed, n_games, max_plies, out.ctypes.data, cap)
    return out[:n].copy()


def valid_moves_batch(states, cap=256):
    states = np.ascontiguousarray(states)
    n = len(states)
    plays = np.full((n, cap, 4), NO_MOVE, dtype=np.int8)
    counts = np.zeros(n, dtype=np.uint32)
    lib().or_bg_valid_moves_batch(states.ctypes.data, n, plays.ctypes.data, cap, counts.ctypes.data)
    return plays, counts


def encode_batch(states, plays):
    states = np.ascontiguousarray(states); plays = np.ascontiguousarray(plays, dtype=np.int8)
    codes = np.zeros(len(states), dtype=np.uint32)
    lib().or_bg_encode_batch(states.ctypes.data, plays.ctypes.data, len(states), codes.ctypes.data)
    return codes


def decode_batch(states, codes):
    states = np.ascontiguousarray(states); codes = np.ascontiguousarray(codes, dtype=np.uint32)
    plays = np.zeros((len(states), 4), dtype=np.int8)
    lib().or_bg_decode_batch(states.ctypes.data, codes.ctypes.data, len(states), plays.ctypes.data)
    return plays


def apply_batch(states, plays, dice):
    out = np.ascontiguousarray(states).copy(); plays = np.ascontiguousarray(plays, dtype=np.int8)
    dice = np.ascontiguousarray(dice, dtype=np.uint8)
    lib().or_bg_apply_batch(out.ctypes.data, plays.ctypes.data, dice.ctypes.data, len(out))
    return out


def planes_batch(states):
    states = np.ascontiguousarray(states)
    out = np.zeros((len(states), BG_PLANES), dtype=np.float32)
    lib().or_bg_planes_batch(states.ctypes.data, len(states), out.ctypes.data)
    return out
