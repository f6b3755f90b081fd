"""what the one barrier per layer costs the fused tower: forward latency (us) of the product build vs a timing build without it
(DIEE_OUT=libdiee_ab.so DIEE_EXTRA_FLAGS=-DDIEE_TOWER_ABLATE=5: results are wrong, only the time means something)"""
import os, sys
sys.path.insert(0, ".")
import diee_amd
libs = {"product": None, "no barrier": os.path.join("die-e_amd", "libdiee_ab.so")}
for G in (1024, 2048, 700):
    out = []
    for rep in range(3):
        for name, path in libs.items():
            L = diee_amd.load_library(path) if path else diee_amd.load_library()
            diee_amd._lib = L
            e = diee_amd.Engine(0); e.load_weights(diee_amd.random_weights(0))
            out.append((name, e.conv_bench(G, 108 if G > 928 else 106, 150)[2]))
            e.close()
    print(G, " ".join(f"{n}:{t:.1f}" for n, t in out), flush=True)
