"""Arena: the second caller of the hot path.  Mirrors the reference's
src/versus.rs: play() (:160-268), get_actions_for_player (:270-318), Game JSON save / load / replay
(:27-122).  Model agents search on the HIP engine (diee_mcts_batch); the Mcts agent (vanilla UCB1 with
rollouts) is not provided: its rollout is broken in the reference (node.rs:181).

`play` is written against a small backend interface so that the same driver can run on the engine or,
in the parity tests, on the CPU oracle.
"""
import json
import os
import secrets

import numpy as np

from . import BG_ACTIONS, BG_STATE, NO_MOVE

TAG_INIT_ROLL, TAG_MOVE_ROLL, TAG_SAMPLE = 0xFFFFFFFF, 0xFFFFFFFE, 0xFFFFFFFD
START = [2, 0, 0, 0, 0, -5, 0, -3, 0, 0, 0, 5, -5, 0, 0, 0, 3, 0, 5, 0, 0, 0, 0, -2]   # backgammon_logic.rs:83-88


class Agent:                                   # versus.rs:17-20
    RANDOM, MCTS, MODEL, NONE = "Random", "Mcts", "Model", "None"

    @staticmethod
    def parse(s):                              # main.rs:126-144
        m = {"model": Agent.MODEL, "mcts": Agent.MCTS, "random": Agent.RANDOM}
        if s is None or s.lower() not in m:
            raise ValueError("Incorrect specification for agent's type.")
        return m[s.lower()]


class Player:                                  # versus.rs:124-127
    def __init__(self, player_type, model=None):
        self.player_type, self.model = player_type, model


class EngineRules:
    """rules / RNG / arithmetic served by the HIP engine (any engine instance will do)"""

    def __init__(self, engine):
        self.e = engine

    def valid_moves(self, states):
        return self.e.get_valid_moves(states)

    def apply(self, states, plays, dice):
        return self.e.apply_move(states, plays, dice)

    def decode(self, states, codes):
        return self.e.decode(states, codes)

    def powf(self, x, y):
        return self.e.det_pow(x, y)

    def draws(self, seed, ctr):
        return self.e.probe_dice(seed, ctr)     # (dice [n,2], uniform [n])


class EngineSearch:
    """alpha_mcts_parallel + get_prob_tensor_parallel on an engine loaded with one player's weights"""

    def __init__(self, engine):
        self.e = engine

    def mcts(self, states, cfg, seed, step, ids, rounds):
        r = self.e.alpha_mcts_parallel(states, cfg, seed, step, ids, rounds, ref_quirks=True)
        return r["probs"], r["n_children"]


def new_states(n):
    s = np.zeros(n, dtype=BG_STATE)
    s["pts"] = START
    s["player"] = -1
    return s


def skip_turn(states, dice):                   # backgammon_logic.rs:192-196
    out = states.copy()
    out["second"] = 0
    out["player"] = -out["player"]
    out["roll"] = dice
    return out


def check_winner(states):                      # backgammon_logic.rs:527-534 (player -1 first)
    w = np.zeros(len(states), dtype=np.int8)
    w[states["off"][:, 1] == 15] = 1
    w[states["off"][:, 0] == 15] = -1
    return w


def weighted_select(row, u01):                 # alphazero.rs:129-137 (rand WeightedIndex over f64 weights)
    """cumulative weights in index order (sequential f64 sums, like the device code), u ~ U[0,total), first
    index whose cumulative weight exceeds u"""
    cum = np.cumsum(row.astype(np.float64))    # np.cumsum is a sequential running sum
    x = u01 * cum[-1]
    hit = np.nonzero(cum > x)[0]
    if len(hit):
        return int(hit[0])
    nz = np.nonzero(row)[0]
    return int(nz[-1]) if len(nz) else 0


class Game:                                    # versus.rs:27-52
    def __init__(self, player1, player2, state, idx):
        self.id = secrets.token_urlsafe(16)[:21]
        self.player1, self.player2 = player1, player2
        self.turns = []                        # never populated by the reference either (Q21)
        self.winner = Agent.NONE
        self.initial_state = {"board": [[int(x) for x in state["pts"]], [int(x) for x in state["bar"]],
                                        [int(x) for x in state["off"]]],
                              "roll": [int(x) for x in state["roll"]], "player": int(state["player"]),
                              "is_second_play": bool(state["second"]), "id": int(idx)}

    def to_json(self):
        return {"id": self.id, "player1": self.player1, "player2": self.player2, "turns": self.turns,
                "winner": self.winner, "initial_state": self.initial_state}


def save_game(game, game_path):                # versus.rs:54-63
    path = os.path.join(game_path, f"{game.id}.json")
    with open(path, "w") as f:
        json.dump(game.to_json(), f, indent=2)
    return path


def load_game(path):                           # versus.rs:65-73
    with open(path) as f:
        return json.load(f)


def to_pretty_str(st):                         # backgammon_logic.rs:110-174
    board = st["board"][0]
    bot = [str(i) for i in range(11, -1, -1)]
    top = [str(i) for i in range(12, 24)]

    def cell(i, n):
        if i == 6 and i <= abs(n):
            return f"+{abs(n) - 5}"
        if i <= abs(n):
            return "x" if n < 0 else "o"
        return " "
    inner = [[cell(i, board[p]) for p in range(11, -1, -1)] for i in range(1, 7)]
    inner.append([" "] * 12)
    inner += [[cell(i, board[p]) for p in range(12, 24)] for i in range(6, 0, -1)]
    rows = [bot] + inner + [top]
    for r in rows:
        r.insert(len(r) // 2, "|"); r.insert(0, "|"); r.append("|")
    body = "\n".join("\t".join(r) for r in reversed(rows))
    who = "Player 1" if st["player"] == -1 else "Player 2"
    info = (f"Current turn: {who}\tRoll: {tuple(st['roll'])}\n"
            f"Player 1:\n\tBroken Pieces: {st['board'][1][0]}\n\tPieces Collected:{st['board'][2][0]}\n"
            f"Player 2:\n\tBroken Pieces: {st['board'][1][1]}\n\tPieces Collected:{st['board'][2][1]}")
    bar = "=" * 110
    return f"{info}\n{bar}\n{body}\n{bar}"


def print_game(path, wait_user_input=False, out=print):     # versus.rs:75-105
    g = load_game(path)
    out(f"Game ID: {g['id']}")
    out(f"Player 1: {g['player1']}, Player 2: {g['player2']}")
    out(f"Game winner: {g['winner']}")
    out("Initial State:")
    out(to_pretty_str(g["initial_state"]))
    for turn in g["turns"]:
        out(f"Player: {turn['player']}"); out(f"Roll: {turn['roll']}"); out(f"Action: {turn['action']}")
        out("State after action has been played:"); out(to_pretty_str(g["initial_state"]))
        if wait_user_input:
            input("Press Enter to continue...")


class PlayResult:                              # versus.rs:129-152
    def __init__(self, player1, player2, wins_p1, wins_p2, n_games, games):
        self.player1, self.player2, self.wins_p1, self.wins_p2 = player1, player2, wins_p1, wins_p2
        self.n_games, self.games = n_games, games
        self.draws = n_games - (wins_p1 + wins_p2)
        self.winrate = wins_p1 / n_games       # from p1's perspective

    def __str__(self):
        return (f"Player 1: {self.player1}\nPlayer 2: {self.player2}\nWins Player 1: {self.wins_p1}\n"
                f"Wins Player 2: {self.wins_p2}\nDraws: {self.draws}\nNumber of Games: {self.n_games}\n"
                f"Winrate: {self.winrate * 100.}%\n")


def get_actions_for_player(player, states, ids, rnd, mcts_config, temp, seed, rules, search, side):
    """versus.rs:270-318 -> plays int8 [n,4] (all NO_MOVE = EMPTY_MOVE)"""
    n = len(states)
    plays = np.full((n, 4), NO_MOVE, dtype=np.int8)
    if n == 0:
        return plays
    ctr = np.stack([ids, np.full(n, rnd), np.full(n, TAG_SAMPLE), np.zeros(n)], axis=1).astype(np.uint32)
    _, uni = rules.draws(seed, ctr)
    if player.player_type == Agent.MODEL:
        probs, nch = search.mcts(states, mcts_config, seed, 2 * rnd + side, ids.astype(np.uint32),
                                 np.full(n, rnd, dtype=np.uint32))
        inv_t = np.float32(1.0 / float(temp))
        codes = np.full(n, 1351, dtype=np.uint32)
        ri, ci = np.nonzero(np.nan_to_num(probs) > 0)
        powed = np.zeros((n, BG_ACTIONS), dtype=np.float32)
        if len(ri):
            powed[ri, ci] = rules.powf(probs[ri, ci].astype(np.float32), inv_t)  # .pow_(1.0 / temp), :283 (one batched call)
        for i in range(n):
            if nch[i] == 0 or float(powed[i].sum()) == 0.0:   # :292 all-zero row (e.g. iterations = 0) or root.children.is_empty() -> EMPTY_MOVE
                continue
            codes[i] = weighted_select(powed[i], float(uni[i]))
        return rules.decode(states, codes)     # :301
    if player.player_type == Agent.RANDOM:     # :308-317
        vm, cnt = rules.valid_moves(states)
        for i in range(n):
            if cnt[i]:
                plays[i] = vm[i][min(int(uni[i] * cnt[i]), cnt[i] - 1)]
        return plays
    raise NotImplementedError("Agent::Mcts (vanilla MCTS with rollouts) is out of scope: SURVEY section 2 row 11")


def play(player1, player2, mcts_config, temp, seed=0xD1EE0001, num_games=400, round_limit=400, rules=None,
         search1=None, search2=None):
    """play(), versus.rs:160-268.  player1 plays side -1 in every game; the second half of the games starts
    with side +1 to move (:172-174)."""
    if rules is None:
        eng = player1.model or player2.model
        if eng is None:
            raise ValueError("an engine is needed for the rules (pass rules=...)")
        rules = EngineRules(eng)
    if search1 is None and player1.player_type == Agent.MODEL:
        search1 = EngineSearch(player1.model)
    if search2 is None and player2.player_type == Agent.MODEL:
        search2 = EngineSearch(player2.model)

    states = new_states(num_games)
    idx = np.arange(num_games)
    ctr = np.stack([idx, np.zeros(num_games), np.full(num_games, TAG_INIT_ROLL), np.zeros(num_games)], axis=1).astype(np.uint32)
    dice, _ = rules.draws(seed, ctr)
    states["player"][num_games // 2:] = 1      # skip_turn() for idx >= num_games/2 (its roll is overwritten by roll_die, Q23)
    states["roll"] = dice
    games = [Game(player1.player_type, player2.player_type, states[i], i) for i in range(num_games)]
    alive = np.ones(num_games, dtype=bool)
    wins_p1 = wins_p2 = 0
    played = []
    round_count = 0
    while alive.any():                         # :191
        live = np.nonzero(alive)[0]
        p1 = live[states["player"][live] == -1]  # :195-196 partition by side to move
        p2 = live[states["player"][live] != -1]
        acts = {}
        for side, (pl, ids, srch) in enumerate(((player1, p1, search1), (player2, p2, search2))):
            a = get_actions_for_player(pl, states[ids], ids, round_count, mcts_config, temp, seed, rules, srch, side)
            for k, g in enumerate(ids):
                acts[int(g)] = a[k]
        rnd = round_count
        round_count += 1                       # :219
        order = np.concatenate([p1, p2])
        ctr = np.stack([order, np.full(len(order), rnd), np.full(len(order), TAG_MOVE_ROLL), np.zeros(len(order))],
                       axis=1).astype(np.uint32)
        dice, _ = rules.draws(seed, ctr)
        pl = np.stack([acts[int(g)] for g in order])
        empty = pl[:, 0] == NO_MOVE
        if empty.any():                        # :225-228 skip_turn; continue (no winner / round-limit check)
            states[order[empty]] = skip_turn(states[order[empty]], dice[empty])
        mv = ~empty
        if mv.any():
            sub = order[mv]
            vm, cnt = rules.valid_moves(states[sub])
            for k in range(len(sub)):          # :229 assert!(valid_moves.contains(action))
                assert (vm[k][:cnt[k]] == pl[mv][k]).all(axis=1).any(), "decoded action is not a valid move"
            states[sub] = rules.apply(states[sub], pl[mv], dice[mv])
            win = check_winner(states[sub])
            for k, g in enumerate(sub):
                w = int(win[k]) if win[k] != 0 else (0 if round_count >= round_limit else None)   # :233-237
                if w is None:
                    continue
                alive[g] = False
                if w == -1:
                    games[g].winner = player1.player_type; wins_p1 += 1
                elif w == 1:
                    games[g].winner = player2.player_type; wins_p2 += 1
                else:
                    games[g].winner = Agent.NONE
                played.append(games[g])
    res = PlayResult(player1.player_type, player2.player_type, wins_p1, wins_p2, num_games, played)
    res.final_states = states
    res.rounds = round_count
    return res


# ---- the same arena for tic-tac-toe (play::<TicTacToe>, versus.rs:160-268 is generic over LearnableGame) ------------------
def play_tictactoe(player1, player2, mcts_config, temp, seed=0xD1EE0001, num_games=400, round_limit=400):
    """Model / Random agents on the tic-tac-toe host engine (BASELINE configs[0]); player1 plays side -1 in every game, the
    second half of the games starts after a skip_turn() (:172-174), a draw (check_winner = Some(0)) has no winner"""
    from . import TTT_ACTIONS, ttt_apply_move, ttt_check_winner, ttt_new, ttt_valid_moves
    states = ttt_new(num_games)
    states["player"][num_games // 2:] = 1
    alive = np.ones(num_games, dtype=bool)
    wins_p1 = wins_p2 = 0
    round_count = 0
    rules_eng = player1.model or player2.model
    while alive.any():
        live = np.nonzero(alive)[0]
        sides = (live[states["player"][live] == -1], live[states["player"][live] != -1])
        acts = {}
        for side, (pl, ids) in enumerate(((player1, sides[0]), (player2, sides[1]))):
            if len(ids) == 0:
                continue
            # the sampling uniforms: the engine's Philox stream (seed, game, round, TAG_SAMPLE); tic-tac-toe has no dice
            uni = np.array([_uniform01(seed, int(g), round_count) for g in ids])
            if pl.player_type == Agent.MODEL:
                r = pl.model.alpha_mcts_parallel(states[ids], mcts_config, seed, 2 * round_count + side, ids.astype(np.uint32),
                                                 np.full(len(ids), round_count, dtype=np.uint32), ref_quirks=True)
                inv_t = np.float32(1.0 / float(temp))
                for k, g in enumerate(ids):
                    if r["n_children"][k] == 0:
                        continue                                                      # EMPTY_MOVE, versus.rs:292
                    row = np.nan_to_num(r["probs"][k]).astype(np.float32)
                    nz = row > 0
                    powed = np.zeros(TTT_ACTIONS, dtype=np.float32)
                    powed[nz] = np.power(row[nz].astype(np.float64), float(inv_t)).astype(np.float32)   # .pow_(1.0 / temp), :283
                    if float(powed.sum()) == 0.0:
                        continue                                                      # `prob_tensor.sum() == 0` -> EMPTY_MOVE too, versus.rs:286-293
                    acts[int(g)] = weighted_select(powed, float(uni[k]))
            elif pl.player_type == Agent.RANDOM:
                for k, g in enumerate(ids):
                    vm = ttt_valid_moves(states[g])
                    if vm:
                        acts[int(g)] = vm[min(int(uni[k] * len(vm)), len(vm) - 1)]
            else:
                raise NotImplementedError("Agent::Mcts (vanilla MCTS with rollouts) is out of scope: SURVEY section 2 row 11")
        round_count += 1                                                                 # :219
        for g in np.concatenate(sides):
            g = int(g)
            if g not in acts:                                                            # :225-228 skip_turn; continue
                states[g]["player"] = -states[g]["player"]
                continue
            assert acts[g] in ttt_valid_moves(states[g]), "decoded action is not a valid move"       # :229
            states[g] = ttt_apply_move(states[g], acts[g])
            w = ttt_check_winner(states[g])
            if w is None and round_count >= round_limit:
                w = 0                                                                    # :235
            if w is None:
                continue
            alive[g] = False
            wins_p1 += w == -1; wins_p2 += w == 1
    res = PlayResult(player1.player_type, player2.player_type, int(wins_p1), int(wins_p2), num_games, [])
    res.final_states = states
    res.rounds = round_count
    return res


def _uniform01(seed, game, rnd):
    """draw_uniform of csrc/bg_device.h (Philox4x32-10 keyed by seed / game / round / TAG_SAMPLE) on the host, for the arena's
    sampling of a game without an engine-side draw entry point"""
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    c = [game & 0xFFFFFFFF, rnd & 0xFFFFFFFF, TAG_SAMPLE, 0]
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k0) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c[3] ^ k1) & 0xFFFFFFFF, p0 & 0xFFFFFFFF]
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    x = (c[3] << 32) | c[2]
    return (x >> 11) * (1.0 / 9007199254740992.0)
