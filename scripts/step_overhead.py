"""per-move-step overhead of self-play beside its search iterations: time per move-step at several `iterations`, fitted as a + b * (iterations + 1)
(the root evaluation costs one iteration): a = what a move-step costs besides its evaluations (host round trips, small kernels)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import diee_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
e = diee_amd.Engine(0); e.load_weights(diee_amd.random_weights(0))
pts = []
for iters in (1, 5, 20, 50, 100):
    cfg = diee_amd.MctsConfig.default(iterations=iters)
    e.self_play_parallel(n, cfg, temperature=1.25, seed=3, max_steps=2)
    steps = 12
    t = time.time()
    for r in range(4):
        e.self_play_parallel(n, cfg, temperature=1.25, seed=4 + r, max_steps=steps)
    dt = (time.time() - t) / (4 * steps)
    pts.append((iters + 1, dt * 1e6))
    print(f"n={n} iterations={iters:3d}: {dt * 1e6:8.1f} us per move-step", flush=True)
x = np.array([p[0] for p in pts], dtype=np.float64); y = np.array([p[1] for p in pts])
b, a = np.polyfit(x, y, 1)
print(f"n={n}: {b:.1f} us per evaluation + {a:.1f} us per move-step (incl. the call's own set-up, shared by {steps} move-steps)")
