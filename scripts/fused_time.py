"""fused tower geometries 6 (<4,8,6>) vs 8 (<4,8,3>): whole-forward latency, alternating (us)"""
import sys
sys.path.insert(0, ".")
import diee_amd
e = diee_amd.Engine(0); e.load_weights(diee_amd.random_weights(0))
for G in (1024, 960, 900, 832, 768, 720):
    r = {106: [], 108: []}
    for rep in range(3):
        for v in (106, 108):
            r[v].append(e.conv_bench(G, v, 60)[2])
    print(f"G {G}: g6 " + " ".join(f"{x:6.1f}" for x in r[106]) + "   g8 " + " ".join(f"{x:6.1f}" for x in r[108]), flush=True)
