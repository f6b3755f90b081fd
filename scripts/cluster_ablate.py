"""Timing experiment (libdiee_clock.so built with -DDIEE_TOWER_ABLATE=4): cluster tower with L2-resident weights."""
import os, sys
sys.path.insert(0, ".")
import diee_amd
which = sys.argv[1] if len(sys.argv) > 1 else "libdiee_clock.so"
L = diee_amd.load_library(os.path.join("die-e_amd", which)); diee_amd._lib = L
e = diee_amd.Engine(0); e.load_weights(diee_amd.random_weights(0))
for G, v in ((2, 201), (16, 201), (32, 201), (64, 202), (128, 204)):
    us = e.conv_bench(G, v, 100)
    print(which, "G", G, "variant", v, "forward us %.1f" % us[2], flush=True)
