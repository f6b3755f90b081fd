"""Cluster tower (k_tower_cl) vs per-layer kernels: bit-identity and forward latency at small batches."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diee_amd

def random_states(n, seed):
    """random (not necessarily reachable) boards: enough for the network, no oracle needed"""
    rng = np.random.default_rng(seed)
    s = np.zeros(n, dtype=diee_amd.BG_STATE)
    s["pts"] = rng.integers(-4, 5, size=(n, 24)); s["bar"] = rng.integers(0, 3, size=(n, 2)); s["off"] = rng.integers(0, 6, size=(n, 2))
    s["roll"] = rng.integers(1, 7, size=(n, 2)); s["player"] = rng.choice([-1, 1], size=n); s["second"] = rng.integers(0, 2, size=n)
    return s

states = random_states(260, 11)
w = diee_amd.random_weights(0)

def engine(cl):
    os.environ["DIEE_TOWER_CL"] = cl
    e = diee_amd.Engine(0)
    e.load_weights(w)
    return e

ref = engine("none")
cl = engine("32:1,64:2,128:4")
cl4 = engine("128:4")
cl8 = engine("256:8")
for G in (81, 100, 129, 177, 200, 256):
    p0, v0 = ref.forward_t(states[:G]); p8, v8 = cl8.forward_t(states[:G])
    print(f"G {G:4d}  8 boards x 4-way split == per-layer: {bool((p0 == p8).all() and (v0 == v8).all())}  max|dp| {np.abs(p0 - p8).max():.2e}", flush=True)
for G in (1, 2, 3, 5, 16, 17, 33, 64, 65, 100, 128, 129):
    p0, v0 = ref.forward_t(states[:G])
    t = time.time(); p1, v1 = cl.forward_t(states[:G]); dt = time.time() - t
    p2, v2 = cl4.forward_t(states[:G])
    print(f"G {G:4d}  cluster==per-layer: policy {bool((p0 == p1).all())} value {bool((v0 == v1).all())}  "
          f"max|dp| {np.abs(p0 - p1).max():.2e}  GT4: max|dp| {np.abs(p0 - p2).max():.2e}  ({dt * 1e3:.1f} ms)", flush=True)
# repeatability (races would show as run-to-run differences)
for G in (7, 64, 128):
    a = cl.forward_t(states[:G])[0]
    same = all((cl.forward_t(states[:G])[0] == a).all() for _ in range(20))
    print(f"G {G}: 20 repeated forwards identical: {same}", flush=True)
for G in (16, 32, 64, 96, 128, 160, 200, 256):
    row = [f"G {G:4d}"]
    for v in (6, 103, 202, 204, 208):
        try:
            us = ref.conv_bench(G, v, 100)
            row.append(f"v{v} fwd {us[2]:7.1f} us")
        except Exception as ex:
            row.append(f"v{v} failed {ex}")
    print("  ".join(row), flush=True)
