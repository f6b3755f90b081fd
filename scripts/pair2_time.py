"""forward latency (us) at 129 ... 256 boards: pair tower with 2 boards per pair (variant 111) against the cluster tower with 8 boards per
cluster (208) and the 4-board pair tower (110); at 65 ... 128: against the cluster tower with 4 boards per cluster (204)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diee_amd
e = diee_amd.Engine(0); e.load_weights(diee_amd.random_weights(0))
for G in (96, 128, 136, 160, 200, 256):
    print(f"  G {G}: " + "  ".join(f"v{v} {e.conv_bench(G, v, 60)[2]:6.1f}" for v in (111, 110, 208 if G > 128 else 204)), flush=True)
