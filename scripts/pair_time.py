"""forward latency (us) of the pair tower (variant 110 = k_tower16p) next to the 2-board and 4-board fused geometries (103, 106)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diee_amd
e = diee_amd.Engine(0); e.load_weights(diee_amd.random_weights(0))
for G in (260, 320, 384, 416, 448, 512):
    print(f"  G {G}: " + "  ".join(f"v{v} {e.conv_bench(G, v, 60)[2]:6.1f}" for v in (110, 103, 106)), flush=True)
