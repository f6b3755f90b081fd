"""per-phase shader-clock averages of k_expand (a -DDIEE_EXPAND_STAMPS build given as DIEE_LIB): one move-step of search on
argv[1] games at iterations = 100"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import diee_amd
games = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
L = diee_amd.load_library()
L.diee_dev_expand_stamps.argtypes = [C.c_void_p, C.c_int]; L.diee_dev_expand_stamps.restype = C.c_int
e = diee_amd.Engine(0); e.load_weights(diee_amd.random_weights(0))
cfg = diee_amd.MctsConfig(iterations=100, c=2.0, round_limit=400, dir_alpha=0.3, dir_eps=0.25)
e.self_play_parallel(games, cfg, 1.25, 1, ref_quirks=True, max_steps=1, fetch=False)       # warm-up
out = (C.c_ulonglong * 16)()
assert L.diee_dev_expand_stamps(None, 1) == 0
e.self_play_parallel(games, cfg, 1.25, 2, ref_quirks=True, max_steps=2, fetch=False)
assert L.diee_dev_expand_stamps(out, 0) == 0
n = out[15]
names = ["flags, value head, leaf meta", "leaf state + legal plays", "softmax constants", "encode, priors, ordered sum", "child creation (issue)",
         "barrier: stores acknowledged", "backpropagation", "barrier before the descent", "descent + leaf state + flags"]
names.append("first round of loads (entry -> flags known)")
tot = sum(out[i] for i in range(10))
print(f"{games} games: {n} waves, {tot / n:.0f} clocks per wave (shader clocks)")
for i, nm in enumerate(names):
    print(f"  {nm:34s} {out[i] / n:8.1f} clocks  {100.0 * out[i] / tot:5.1f} %")
