"""helpers shared by the development scripts"""
import numpy as np
import diee_amd


def random_states(n, seed):
    """random (not necessarily reachable) boards: enough for the network, no oracle needed"""
    rng = np.random.default_rng(seed)
    s = np.zeros(n, dtype=diee_amd.BG_STATE)
    s["pts"] = rng.integers(-4, 5, size=(n, 24)); s["bar"] = rng.integers(0, 3, size=(n, 2)); s["off"] = rng.integers(0, 6, size=(n, 2))
    s["roll"] = rng.integers(1, 7, size=(n, 2)); s["player"] = rng.choice([-1, 1], size=n); s["second"] = rng.integers(0, 2, size=n)
    return s
