"""The fused tower on ONE wave per SIMD (4 waves x 4 column fragments: half the A-fragment LDS reads of the 8-wave geometries, the
same weight traffic) with 3 / 6 / 9 / 18 weight k-steps in flight, against the product's 8-wave geometry: whole forward in us at
1024 / 768 boards (diee_dev_conv_bench; geometry ids of launch_tower)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diee_amd
e = diee_amd.Engine(0); e.load_weights(diee_amd.random_weights(0))
# (9 / 18 weight k-steps in flight -- geometries 12 / 13 of commit 6cfd4f5, since removed: 618 / 1946 us at 1024 boards, 63 / 670 spilled dwords)
for name, v in (("<4,8,3> (rounds 1-3)", 108), ("<4,8,6> (rounds 1-3)", 106), ("<4,4,3> product", 105), ("<4,4,6>", 102)):
    row = []
    for G in (1024, 768, 520):
        row.append(f"G={G}: " + " ".join(f"{e.conv_bench(G, v, 60)[2]:6.1f}" for _ in range(3)))
    print(f"{name:18s} " + "   ".join(row), flush=True)
