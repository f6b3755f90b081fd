"""Where k_tail's time goes: shader-clock sums per phase over all its workgroups (development build -DDIEE_TAIL_STAMPS).

    DIEE_EXTRA_FLAGS=-DDIEE_TAIL_STAMPS DIEE_OUT=libdiee_tail_stamps.so python die-e_amd/build.py --dev
    DIEE_LIB=die-e_amd/libdiee_tail_stamps.so python tests/tools/tail_phases.py [games ...]

One move-step's search at each number of live games; cycles per workgroup and iteration at the shader clock (~2.4 GHz unloaded)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import diee_amd
from oracle import oracle as orc

L = diee_amd.load_library()
L.diee_dev_tail_stamps.argtypes = [C.c_void_p, C.c_int]; L.diee_dev_tail_stamps.restype = C.c_int
e = diee_amd.Engine(0); e.load_weights(diee_amd.random_weights(0))
walk = orc.random_walk_states(7, 60)
cfg = diee_amd.MctsConfig.default(100)
names = ["tree into LDS + take-in of the launch's rows", "meeting (waiting for the other games)", "iteration body", "record out + plan (virtual descents, rows)"]
for n in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16, 32, 64]:
    states = walk[100:100 + 5 * n:5]
    gids = np.arange(n, dtype=np.uint32); rds = np.zeros(n, dtype=np.uint32)
    e.alpha_mcts_parallel(states, cfg, 1, 0, gids, rds, ref_quirks=True)            # warm
    L.diee_dev_tail_stamps(None, 1)
    r = e.alpha_mcts_parallel(states, cfg, 1, 0, gids, rds, ref_quirks=True)
    out = (C.c_ulonglong * 8)()
    L.diee_dev_tail_stamps(out, 0)
    its, launches = out[6], out[7]
    st = r["stats"]
    print(f"{n:3d} games: {st['tail_launches']} launches for {st['tail_iterations']} iterations, {st['seconds'] * 1e3:.2f} ms; per workgroup: "
          + ", ".join(f"{names[i]} {out[i] / max(its if i in (1, 2) else launches, 1):.0f} cycles per {'iteration' if i in (1, 2) else 'launch'}" for i in range(4)))
