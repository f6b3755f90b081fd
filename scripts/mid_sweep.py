"""257..416 boards: the 2-board fused geometry (variant 103, k_tower16<2,8,9>, L2-bound) against 4 boards per workgroup on
fewer CUs (106, k_tower16<4,8,6>) and 3 boards (107): whole-forward latency (us), alternating, 3 repeats."""
import sys
sys.path.insert(0, ".")
import diee_amd
e = diee_amd.Engine(0); e.load_weights(diee_amd.random_weights(0))
for G in (260, 288, 320, 352, 384, 416, 448, 512):
    r = {103: [], 107: [], 106: []}
    for rep in range(3):
        for v in r:
            r[v].append(e.conv_bench(G, v, 60)[2])
    print(f"G {G:4d}: " + "   ".join(f"v{v} " + " ".join(f"{x:6.1f}" for x in xs) for v, xs in r.items()), flush=True)
