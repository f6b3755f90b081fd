"""Runs the ResNet forward in a loop (for rocprofv3 --kernel-trace --stats)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import diee_amd
from oracle import oracle as orc

G = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
e = diee_amd.Engine(0)
e.load_weights(diee_amd.random_weights(0))
st = orc.random_walk_states(1, 40)[:G]
assert len(st) == G
e.forward_t(st)
t = time.time()
for _ in range(reps):
    e.forward_t(st)
dt = (time.time() - t) / reps
print(f"G={G} forward incl. H2D/D2H: {dt*1e3:.3f} ms  -> {G/dt:.0f} evals/s, {G*1.0825e9/dt/1e12:.1f} TFLOP/s")
