"""forward latency of the cluster tower variants (whole forward, us); optional argument: another build of the library"""
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "scripts")
import diee_amd
if len(sys.argv) > 1:
    diee_amd._lib = diee_amd.load_library(os.path.join("die-e_amd", sys.argv[1]))
e = diee_amd.Engine(0); e.load_weights(diee_amd.random_weights(0))
print(" ".join(f"G{G}/v{v}: {e.conv_bench(G, v, 100)[2]:6.1f}" for G, v in ((16, 201), (64, 202), (128, 204), (256, 208))), flush=True)
