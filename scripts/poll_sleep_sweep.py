"""cluster tower: s_sleep between two polls of the input tile (DIEE_LIB=die-e_amd/libdiee_ps{6,12,24}.so vs the product's 2):
forward latency (us); one process per build (run as: for v in "" 6 12 24; do DIEE_LIB=... python scripts/poll_sleep_sweep.py; done)"""
import os, sys
sys.path.insert(0, ".")
import diee_amd
tag = os.path.basename(os.environ.get("DIEE_LIB", "libdiee.so"))
for G, v in ((8, 201), (32, 201), (64, 202), (128, 204)):
    res = []
    for rep in range(4):
        e = diee_amd.Engine(0); e.load_weights(diee_amd.random_weights(0))
        res.append(e.conv_bench(G, v, 300)[2]); e.close()
    print(tag, G, " ".join(f"{t:.1f}" for t in res), flush=True)
