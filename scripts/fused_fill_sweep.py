"""forward latency (us) of the 4-board fused tower against the number of workgroups on the chip (boards / 4): where does the
time per launch jump from ~435 us (<= 928 boards) to ~630 us (1024)?  variant 106 = <4,8,6>, 108 = <4,8,3>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diee_amd
e = diee_amd.Engine(0); e.load_weights(diee_amd.random_weights(0))
for G in (512, 640, 768, 832, 896, 928, 944, 960, 976, 992, 1008, 1016, 1020, 1024):
    print(f"G {G:5d} ({(G + 3) // 4:3d} workgroups): <4,8,6> " + " ".join(f"{e.conv_bench(G, 106, 40)[2]:6.1f}" for _ in range(2))
          + "   <4,8,3> " + " ".join(f"{e.conv_bench(G, 108, 40)[2]:6.1f}" for _ in range(2)), flush=True)
