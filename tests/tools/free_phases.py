"""TEST INFRASTRUCTURE: where k_free's time goes -- shader-clock sums per phase over all its workgroups (development build -DDIEE_TAIL_STAMPS).

    DIEE_EXTRA_FLAGS=-DDIEE_TAIL_STAMPS DIEE_OUT=libdiee_tail_stamps.so python die-e_amd/build.py --dev
    DIEE_LIB=die-e_amd/libdiee_tail_stamps.so python tests/tools/free_phases.py [games ...] [opt=value ...]
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import diee_amd
from oracle import oracle as orc

L = diee_amd.load_library()
L.diee_dev_free_stamps.argtypes = [C.c_void_p, C.c_int]; L.diee_dev_free_stamps.restype = C.c_int
e = diee_amd.Engine(0); e.load_weights(diee_amd.random_weights(0))
for kv in [a for a in sys.argv[1:] if "=" in a]:
    e.set_option(*kv.split("="))
walk = orc.random_walk_states(99, 400)
cfg = diee_amd.MctsConfig.default(100)
names = ["record + tree into LDS + take-in", "flag words", "iteration body (expand, backpropagate)", "selection + publish", "record out + plan"]
for n in [int(a) for a in sys.argv[1:] if "=" not in a] or [300, 430, 600, 760]:
    states = walk[np.linspace(len(walk) // 3, len(walk) - 1, n).astype(int)]
    gids = np.arange(n, dtype=np.uint32); rds = np.zeros(n, dtype=np.uint32)
    e.alpha_mcts_parallel(states, cfg, 1, 0, gids, rds, ref_quirks=True)            # warm
    L.diee_dev_free_stamps(None, 1)
    r = e.alpha_mcts_parallel(states, cfg, 1, 0, gids, rds, ref_quirks=True)
    out = (C.c_ulonglong * 16)()
    L.diee_dev_free_stamps(out, 0)
    wgs, its = max(out[8], 1), max(out[9], 1)
    st = r["stats"]
    print(f"{n:4d} games: {st['tail_launches']} launches with rows, {st['seconds'] * 1e3:.2f} ms; {wgs} workgroup runs ({wgs / n:.1f} per game), {its / wgs:.2f} iterations and "
          f"{out[11] / wgs:.2f} virtual descents each; cycles per workgroup run: "
          + ", ".join(f"{names[i]} {out[i] / wgs:.0f}" for i in range(5)) + f"; per iteration: flags {out[1] / its:.0f}, body {out[2] / its:.0f}, selection {out[3] / its:.0f}; "
          f"longest workgroup {out[10]} cycles")
