"""wall clock of the FIRST AlphaZero.train call of a process (MIOpen solver search for the three small convolutions that
stay in PyTorch, code-object loads, the eager warm-up steps, graph capture) and of the second"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
az = importlib.import_module("die-e_amd.alphazero")
import torch
import diee_amd
n = 256 * 8
rng = np.random.default_rng(0)
ps = rng.random((n, 1352), dtype=np.float32); ps /= ps.sum(1, keepdims=True)
mem = {"state": rng.integers(-3, 4, size=(n, 144)).astype(np.float32), "ps": ps, "outcome": rng.choice([-1, 1], size=n).astype(np.int8)}
a = az.AlphaZero(None, az.AlphaZeroConfig(1.25, 1, 1, 1, 256, 1024), diee_amd.MctsConfig.default(100), az.OptimizerParams(1e-4, 1e-3),
                 blob=diee_amd.random_weights(0), train_device="cuda", quiet=True)
torch.cuda.synchronize()
for k in range(3):
    t = time.time(); a.train(mem); torch.cuda.synchronize()
    print(f"MIOPEN_FIND_MODE={os.environ.get('MIOPEN_FIND_MODE', '(unset)')}: train call {k}: {time.time() - t:.2f} s for 8 steps", flush=True)
