"""AlphaZero.train on the engine backend with the HIP-graph step, in a loop (for rocprofv3 --kernel-trace --stats)"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
az = importlib.import_module("die-e_amd.alphazero")
import torch
import diee_amd
steps = 40
n = 256 * steps
rng = np.random.default_rng(0)
ps = rng.random((n, 1352), dtype=np.float32); ps /= ps.sum(1, keepdims=True)
mem = {"state": rng.integers(-3, 4, size=(n, 144)).astype(np.float32), "ps": ps, "outcome": rng.choice([-1, 1], size=n).astype(np.int8)}
a = az.AlphaZero(None, az.AlphaZeroConfig(1.25, 1, 1, 1, 256, 1024), diee_amd.MctsConfig.default(100), az.OptimizerParams(1e-4, 1e-3),
                 blob=diee_amd.random_weights(0), train_device="cuda", quiet=True)
a.train(mem)
torch.cuda.synchronize(); t = time.time()
a.train(mem)
torch.cuda.synchronize()
print(f"{(time.time() - t) / steps * 1e3:.2f} ms/step", flush=True)
