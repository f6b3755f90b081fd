"""Oracle-side backends for die-e_amd.versus.play (TEST INFRASTRUCTURE): the same arena driver runs on the
CPU oracle's rules / search so that the engine-backed arena can be checked against it."""
import numpy as np

from . import oracle as orc


class OracleRules:
    def valid_moves(self, states):
        return orc.valid_moves_batch(states, 256)

    def apply(self, states, plays, dice):
        return orc.apply_batch(states, plays, dice)

    def decode(self, states, codes):
        return orc.decode_batch(states, codes)

    def powf(self, x, y):
        return np.array([orc.det_powf(float(v), float(y)) for v in x], dtype=np.float32)

    def draws(self, seed, ctr):
        dice = np.array([orc.dice(seed, *map(int, c)) for c in ctr], dtype=np.uint8).reshape(-1, 2)
        uni = np.array([orc.lib().or_uniform01(seed, *map(int, c)) for c in ctr], dtype=np.float64)
        return dice, uni


class OracleSearch:
    """alpha_mcts_parallel on the oracle with a given evaluator (EVAL_FN, ctx)"""

    def __init__(self, eval_fn, ectx=None):
        self.eval_fn, self.ectx = eval_fn, ectx

    def mcts(self, states, cfg, seed, step, ids, rounds):
        ocfg = orc.MctsCfg(iterations=cfg.iterations, c=cfg.c, round_limit=cfg.round_limit, dir_alpha=cfg.dir_alpha,
                           dir_eps=cfg.dir_eps)
        roots, probs, stats, _ = orc.alpha_mcts_parallel(1, np.ascontiguousarray(states).view(orc.BG_STATE), ocfg,
                                                         self.eval_fn, self.ectx, seed, step, ids, rounds, 1)
        nch = np.array([len(r["children"]) for r in roots], dtype=np.uint32)
        return probs, nch
