"""forward latency (us) of the 4-wave fused geometries (variant 105 = k_tower16<4,4,3>, 102 = <4,4,6>) next to the 8-wave ones
(106, 103) at mid batch sizes; with DIEE_LIB = a -DDIEE_TOWER_HALFN build the 4-wave workgroups compute half their channels
(timing only): what a two-workgroup cluster's member would take per evaluation without its hand-offs"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diee_amd
e = diee_amd.Engine(0); e.load_weights(diee_amd.random_weights(0))
print(os.environ.get("DIEE_LIB", "product build"))
for G in (260, 384, 512):
    print(f"  G {G}: " + "  ".join(f"v{v} {e.conv_bench(G, v, 60)[2]:6.1f}" for v in (105, 102, 106, 103)), flush=True)
