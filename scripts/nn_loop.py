"""Runs the ResNet forward in a loop (for rocprofv3 --kernel-trace --stats)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import diee_amd

G = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
e = diee_amd.Engine(0)
e.load_weights(diee_amd.random_weights(0))
def random_states(n, seed):
    """random (not necessarily reachable) boards: enough for the network, no oracle needed"""
    rng = np.random.default_rng(seed)
    s = np.zeros(n, dtype=diee_amd.BG_STATE)
    s["pts"] = rng.integers(-4, 5, size=(n, 24)); s["bar"] = rng.integers(0, 3, size=(n, 2)); s["off"] = rng.integers(0, 6, size=(n, 2))
    s["roll"] = rng.integers(1, 7, size=(n, 2)); s["player"] = rng.choice([-1, 1], size=n); s["second"] = rng.integers(0, 2, size=n)
    return s
st = random_states(G, 1)
e.forward_t(st)
t = time.time()
for _ in range(reps):
    e.forward_t(st)
dt = (time.time() - t) / reps
print(f"G={G} forward incl. H2D/D2H: {dt*1e3:.3f} ms  -> {G/dt:.0f} evals/s, {G*1.0825e9/dt/1e12:.1f} TFLOP/s")
