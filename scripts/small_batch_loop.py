"""MCTS on a handful of roots (the self-play tail): per-iteration latency; run under rocprofv3 --kernel-trace --stats."""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diee_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
e = diee_amd.Engine(0)
e.load_weights(diee_amd.random_weights(0))
cfg = diee_amd.MctsConfig.default(iterations=100)
# roots: the opening position with a few different rolls (player -1 to move)
out = e.self_play_parallel(n, cfg, temperature=1.25, seed=3, max_steps=1)
t = time.time()
for r in range(reps):
    out = e.self_play_parallel(n, cfg, temperature=1.25, seed=4 + r, max_steps=2)
dt = (time.time() - t) / (reps * 2)
print(f"n={n}: {dt * 1e3:.2f} ms per move-step = {dt * 1e4:.1f} us per MCTS iteration (100 iterations + root)")
