import sys
sys.path.insert(0, "."); sys.path.insert(0, "scripts")
import numpy as np
import diee_amd, os
from scripts_common import random_states
st = random_states(1024, 5)
blob = diee_amd.random_weights(0)
os.environ["DIEE_TOWER_TABLE"] = "0:8"; e8 = diee_amd.Engine(0); e8.load_weights(blob)
os.environ["DIEE_TOWER_TABLE"] = "0:9"; e9 = diee_amd.Engine(0); e9.load_weights(blob)
for G in (1024, 777, 5):
    a = e8.forward_t(st[:G]); b = e9.forward_t(st[:G])
    print("G", G, "border-aware 4-board == dense 3-board:", bool((a[0] == b[0]).all() and (a[1] == b[1]).all()), np.abs(a[0]-b[0]).max())
for G, v in ((1024, 108), (1024, 105), (768, 108), (768, 107), (600, 108), (600, 107), (512, 108), (512, 103), (400, 108), (400, 103)):
    print("G", G, "variant", v, "forward us %.1f" % e8.conv_bench(G, v, 30)[2], flush=True)
