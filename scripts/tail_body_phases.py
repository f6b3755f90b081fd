"""per-phase shader-clock averages of the iteration body inside k_tail (development build -DDIEE_EXPAND_STAMPS -DDIEE_TAIL_STAMPS given as
DIEE_LIB): one move-step's search at argv[1:] live games (positions from the middle game to the bear-off), iterations = 100"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import diee_amd
from oracle import oracle as orc
L = diee_amd.load_library()
L.diee_dev_expand_stamps.argtypes = [C.c_void_p, C.c_int]; L.diee_dev_expand_stamps.restype = C.c_int
L.diee_dev_tail_stamps.argtypes = [C.c_void_p, C.c_int]; L.diee_dev_tail_stamps.restype = C.c_int
e = diee_amd.Engine(0); e.load_weights(diee_amd.random_weights(0))
walk = orc.random_walk_states(7, 60)
late = walk[walk["off"].max(axis=1) >= 8]
cfg = diee_amd.MctsConfig.default(100)
names = ["flags, value head, leaf meta", "leaf state + legal plays", "softmax constants", "encode, priors, ordered sum", "child creation (issue)",
         "barrier: stores acknowledged", "backpropagation", "barrier before the descent", "descent + leaf state + flags"]
for n in [int(a) for a in sys.argv[1:]] or [4, 16, 48]:
    states = np.concatenate([late[:n // 2], walk[200:200 + 5 * (n - n // 2):5]])
    gids = np.arange(n, dtype=np.uint32); rds = np.zeros(n, dtype=np.uint32)
    e.alpha_mcts_parallel(states, cfg, 1, 0, gids, rds, ref_quirks=True)
    L.diee_dev_expand_stamps(None, 1); L.diee_dev_tail_stamps(None, 1)
    r = e.alpha_mcts_parallel(states, cfg, 1, 0, gids, rds, ref_quirks=True)
    ex = (C.c_ulonglong * 16)(); tl = (C.c_ulonglong * 8)()
    L.diee_dev_expand_stamps(ex, 0); L.diee_dev_tail_stamps(tl, 0)
    waves = ex[15]; its = tl[6]
    tot = sum(ex[i] for i in range(9))
    print(f"{n} games: {r['stats']['tail_launches']} launches, {its} (game, iteration) bodies in k_tail (+ {waves - its} in k_expand: the roots); "
          f"k_tail per body: meeting {tl[1] / max(its, 1):.0f} + body {tl[2] / max(its, 1):.0f} cycles; body phases over all {waves} waves:")
    for i, nm in enumerate(names):
        print(f"    {nm:34s} {ex[i] / waves:8.1f} clocks  {100.0 * ex[i] / tot:5.1f} %")
