import ctypes as C, sys, os
sys.path.insert(0, ".")
import diee_amd
for name in ("libdiee.so", "libdiee_ablate1.so", "libdiee_ablate2.so"):
    path = os.path.join("die-e_amd", name)
    if not os.path.exists(path): continue
    diee_amd._lib = None
    L = diee_amd.load_library(path); diee_amd._lib = L
    e = diee_amd.Engine(0); e.load_weights(diee_amd.random_weights(0))
    print(name, [round(e.conv_bench(G, 105, 20)[2], 1) for G in (1024, 512)], "us forward (fused 4-board) at G=1024, 512")
    e.close()
