"""Diagnostic (build with DIEE_EXTRA_FLAGS=-DDIEE_TOWER_ABLATE=3 into libdiee_clock.so): where a cluster-tower layer's cycles go."""
import os, sys
sys.path.insert(0, ".")
os.environ["DIEE_CLUSTER_CLOCK"] = "1"
import diee_amd
L = diee_amd.load_library(os.path.join("die-e_amd", "libdiee_clock.so")); diee_amd._lib = L
e = diee_amd.Engine(0); e.load_weights(diee_amd.random_weights(0))
for G, v in ((2, 201), (16, 202), (64, 202), (128, 204), (256, 208)):
    us = e.conv_bench(G, v, 50)
    print("G", G, "variant", v, "forward us %.1f" % us[2], flush=True)
