"""cluster tower forward latency (us) at small batches for the build given as DIEE_LIB (or the product build), and bit-identity
of its outputs with the per-layer kernels"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diee_amd
rng = np.random.default_rng(11)
n = 260
s = np.zeros(n, dtype=diee_amd.BG_STATE)
s["pts"] = rng.integers(-4, 5, size=(n, 24)); s["bar"] = rng.integers(0, 3, size=(n, 2)); s["off"] = rng.integers(0, 6, size=(n, 2))
s["roll"] = rng.integers(1, 7, size=(n, 2)); s["player"] = rng.choice([-1, 1], size=n); s["second"] = rng.integers(0, 2, size=n)
w = diee_amd.random_weights(0)
os.environ["DIEE_TOWER_CL"] = "none"; ref = diee_amd.Engine(0); ref.load_weights(w)
os.environ.pop("DIEE_TOWER_CL"); cl = diee_amd.Engine(0); cl.load_weights(w)
ok = True
for G in (1, 8, 32, 33, 64, 100, 128, 200, 256):
    p0, v0 = ref.forward_t(s[:G]); p1, v1 = cl.forward_t(s[:G])
    same = bool((p0 == p1).all() and (v0 == v1).all()); ok &= same
    rep = all((cl.forward_t(s[:G])[0] == p1).all() for _ in range(10)); ok &= rep
print(os.environ.get("DIEE_LIB", "product build"), "bit-identical to the per-layer kernels and repeatable:", ok)
print("  " + "  ".join(f"G{G}: " + "/".join(f"{ref.conv_bench(G, v, 200)[2]:.1f}" for v in vs) for G, vs in ((8, (201,)), (32, (201,)), (64, (202,)), (128, (204,)), (256, (208,)))), "us (cluster variants 201/202/204/208)")
