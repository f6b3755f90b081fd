"""one pass of the chip (1024 boards): forward latency (us) of k_tower16<4,8,3> (variant 108) against <4,8,6> (106), alternating"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diee_amd
e = diee_amd.Engine(0); e.load_weights(diee_amd.random_weights(0))
for G in (960, 1024):
    r = {108: [], 106: []}
    for rep in range(4):
        for v in r: r[v].append(e.conv_bench(G, v, 100)[2])
    print(f"G {G}: " + "   ".join(f"v{v} " + " ".join(f"{x:6.1f}" for x in xs) for v, xs in r.items()), flush=True)
