import os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "scripts")
import diee_amd
from scripts_common import random_states
blob = diee_amd.random_weights(0)
states = random_states(700, 3)
# dirty the allocator like the earlier tests of the file do
os.environ["DIEE_TOWER_TABLE"] = "0:8"
e0 = diee_amd.Engine(0); e0.load_weights(blob); e0.forward_t(states); e0.close()
del os.environ["DIEE_TOWER_TABLE"]
os.environ["DIEE_TOWER_CL"] = "none"
ref = diee_amd.Engine(0); ref.load_weights(blob)
os.environ["DIEE_TOWER_CL"] = "32:1,64:2,128:4"
cl = diee_amd.Engine(0); cl.load_weights(blob)
for G in (1, 9, 33, 64):
    p0, v0 = ref.forward_t(states[:G]); p, v = cl.forward_t(states[:G])
    print("G", G, "exact", bool((p == p0).all()))
p128, v128 = cl.forward_t(states[:128])
p0, v0 = ref.forward_t(states[:128])
print("G 128 max diff vs per-layer", np.abs(p128 - p0).max())
for rep in range(3):
    for G in (65, 101, 127, 128):
        p, v = cl.forward_t(states[:G])
        bad = np.where((p != p128[:G]).any(1))[0]
        print("rep", rep, "G", G, "rows differing from the 128-run:", bad[:20], "max", np.abs(p - p128[:G]).max() if len(bad) else 0.0,
              "| vs per-layer max", np.abs(p - p0[:G]).max())
