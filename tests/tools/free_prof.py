"""TEST INFRASTRUCTURE (positions come from the CPU oracle's random walks): times one move-step's search -- diee_mcts_batch, 100 iterations -- on n
mid-game positions with the free-running path on and off; under `rocprofv3 --kernel-trace --stats` it gives the kernels of one configuration.

    python tests/tools/free_prof.py 300,430,600,760 [reps] [free_eval=0|1|both] [opt=value ...]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import diee_amd                                    # noqa: E402
from oracle import oracle as orc                   # noqa: E402

sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "300,600").split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
which = sys.argv[3] if len(sys.argv) > 3 else "both"
opts = dict(kv.split("=") for kv in sys.argv[4:])
orc.build()
e = diee_amd.Engine(0)
e.load_weights(diee_amd.random_weights(0))
for k, v in opts.items():
    e.set_option(k, v)
cfg = diee_amd.MctsConfig.default(iterations=100)
walk = orc.random_walk_states(99, 400)
for n in sizes:
    # positions from ply ~60 on (the live games of a move-step with n games alive are middle-game and bear-off positions)
    states = walk[np.linspace(len(walk) // 3, len(walk) - 1, n).astype(int)]
    gids = np.arange(n, dtype=np.uint32); rds = np.zeros(n, dtype=np.uint32)
    for fe in ((0, 1) if which == "both" else (int(which),)):
        e.set_option("free_eval", fe)
        e.alpha_mcts_parallel(states, cfg, 1, 0, gids, rds, ref_quirks=True)
        t = time.time()
        for r in range(reps):
            out = e.alpha_mcts_parallel(states, cfg, 2 + r, 0, gids, rds, ref_quirks=True)
        dt = (time.time() - t) / reps
        st = out["stats"]
        print(f"n={n} free_eval={fe}: {dt * 1e3:8.2f} ms per search; {st['tail_iterations']} free/tail iterations on {st['tail_launches']} launches, "
              f"{st['tail_spec_rows']} speculative rows, nn_rows {st['nn_rows']}, expansions {st['expansions']}", flush=True)
