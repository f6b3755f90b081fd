import importlib, os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
az = importlib.import_module("die-e_amd.alphazero")
import torch
import diee_amd
os.environ.update({"DIEE_TRAIN": "engine", "DIEE_TRAIN_GRAPH": "1"})
rng = np.random.default_rng(0)
def mk(n):
    ps = rng.random((n, 1352), dtype=np.float32); ps /= ps.sum(1, keepdims=True)
    return {"state": rng.integers(-3, 4, size=(n, 144)).astype(np.float32), "ps": ps, "outcome": rng.choice([-1, 1], size=n).astype(np.int8)}
a = az.AlphaZero(None, az.AlphaZeroConfig(1.25, 1, 1, 1, 256, 1024), diee_amd.MctsConfig.default(100), az.OptimizerParams(1e-4, 1e-3),
                 blob=diee_amd.random_weights(0), train_device="cuda", quiet=True)
for n in (3, 30, 3, 12):
    l = a.train(mk(256 * n)); print(n, np.round(l[:3], 3), flush=True)
