"""fused tower with waves 4..7 delayed by s_sleep(n) at the start of every layer (builds libdiee_st{1,2,4}.so): forward latency (us)"""
import os, sys
sys.path.insert(0, ".")
import diee_amd
libs = {"0": None, "1": "libdiee_st1.so", "2": "libdiee_st2.so", "4": "libdiee_st4.so"}
for G in (1024, 2048):
    res = {k: [] for k in libs}
    for rep in range(3):
        for name, path in libs.items():
            L = diee_amd.load_library(os.path.join("die-e_amd", path)) if path else diee_amd.load_library()
            diee_amd._lib = L
            e = diee_amd.Engine(0); e.load_weights(diee_amd.random_weights(0))
            res[name].append(e.conv_bench(G, 108, 150)[2]); e.close()
    print(G, "  ".join(f"sleep {k}: " + " ".join(f"{t:.1f}" for t in v) for k, v in res.items()), flush=True)
