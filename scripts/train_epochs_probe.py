"""wall clock of the four train() calls (epochs) of a learn iteration over one memory of 440 k fragments, per call"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
az = importlib.import_module("die-e_amd.alphazero")
import torch
import diee_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 439619
rng = np.random.default_rng(0)
ps = rng.random((n, 1352), dtype=np.float32); ps /= ps.sum(1, keepdims=True)
mem = {"state": rng.integers(-3, 4, size=(n, 144)).astype(np.float32), "ps": ps, "outcome": rng.choice([-1, 1], size=n).astype(np.int8)}
a = az.AlphaZero(None, az.AlphaZeroConfig(1.25, 1, 1, 4, 256, 1024), diee_amd.MctsConfig.default(100), az.OptimizerParams(1e-4, 1e-3),
                 blob=diee_amd.random_weights(0), train_device="cuda", quiet=True)
torch.cuda.synchronize()
for k in range(4):
    t = time.time(); losses = a.train(mem, resident=k > 0); torch.cuda.synchronize()
    dt = time.time() - t
    print(f"epoch {k}: {dt:.2f} s, {dt / len(losses) * 1e3:.2f} ms/step over {len(losses)} steps", flush=True)
