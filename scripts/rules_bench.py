"""Stand-alone step-kernel microbench: get_valid_moves over resident states.
Algorithmic bytes per state: 32 B read + k x (32 + 2) B written (the plays' next states and codes the kernel stands for);
the kernel itself writes 4 B per play + 4 B count."""
import sys
sys.path.insert(0, ".")
import numpy as np
import diee_amd

e = diee_amd.Engine(0)
e.load_weights(diee_amd.random_weights(0))
cfg = diee_amd.MctsConfig.default(iterations=8)
# reachable positions: the NN-input states of a short self-play (mid-game mix)
out = e.self_play_parallel(256, cfg, temperature=1.25, seed=11)           # to completion: ~27 k positions
planes = out["state"]                                   # [n, 144] as_tensor planes -> back to states
n = len(planes)
s = np.zeros(n, dtype=diee_amd.BG_STATE)
pl = planes.reshape(n, 6, 24)
s["pts"] = pl[:, 0].astype(np.int8); s["player"] = pl[:, 1, 0].astype(np.int8)
s["bar"][:, 0] = pl[:, 2, 0]; s["bar"][:, 1] = pl[:, 2, 12]; s["off"][:, 0] = pl[:, 3, 0]; s["off"][:, 1] = pl[:, 3, 12]
s["roll"][:, 0] = pl[:, 4, 0]; s["roll"][:, 1] = pl[:, 4, 12]; s["second"] = pl[:, 5, 0].astype(np.uint8)
for m in (1024, 8192, min(n, 1 << 15)):
    us, k = e.rules_bench(s[:m], 50)
    alg = m * (32 + k * 34)
    print(f"{m:7d} states: {us:8.1f} us per launch = {m / us:7.1f} M states/s, mean plays {k:5.2f}, "
          f"algorithmic {alg / us / 1e3:6.1f} GB/s ({alg / us / 1e3 / 8000 * 100:.2f} % of 8 TB/s)", flush=True)
