"""Diagnostic: in-kernel clock of the fused tower (needs die-e_amd/libdiee_clock.so built with
DIEE_EXTRA_FLAGS=-DDIEE_TOWER_ABLATE=3 python die-e_amd/build.py --force)."""
import os, sys
sys.path.insert(0, ".")
os.environ["DIEE_TOWER_CLOCK"] = "1"
import diee_amd
L = diee_amd.load_library(os.path.join("die-e_amd", "libdiee_clock.so")); diee_amd._lib = L
e = diee_amd.Engine(0); e.load_weights(diee_amd.random_weights(0))
for G, v in ((1024, 105), (1024, 108), (768, 104), (768, 107), (512, 103)):
    print("G", G, "variant", v, "forward us", round(e.conv_bench(G, v, 20)[2], 1), flush=True)
