"""AlphaZero.train (the learn loop's training leg) on synthetic fragments: ms per 256-sample step for the all-PyTorch fp32 step
(DIEE_TRAIN=fp32: the default, the reference's arithmetic) and the opt-in bf16 step on the engine's kernels (DIEE_TRAIN=bf16), eager and as a HIP graph."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
az = importlib.import_module("die-e_amd.alphazero")
import torch
import diee_amd
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n = 256 * steps
rng = np.random.default_rng(0)
ps = rng.random((n, 1352), dtype=np.float32); ps /= ps.sum(1, keepdims=True)
mem = {"state": rng.integers(-3, 4, size=(n, 144)).astype(np.float32), "ps": ps, "outcome": rng.choice([-1, 1], size=n).astype(np.int8)}
for name, env in (("fp32 (default): all-PyTorch step (MIOpen), fused Adam", {"DIEE_TRAIN": "fp32", "DIEE_TRAIN_GRAPH": "0"}),
                  ("opt-in bf16 step on the engine's tower kernels, eager", {"DIEE_TRAIN": "bf16", "DIEE_TRAIN_GRAPH": "0"}),
                  ("opt-in bf16 step on the engine's tower kernels, HIP graph", {"DIEE_TRAIN": "bf16", "DIEE_TRAIN_GRAPH": "1"})):
    os.environ.update(env)
    a = az.AlphaZero(None, az.AlphaZeroConfig(1.25, 1, 1, 1, 256, 1024), diee_amd.MctsConfig.default(100), az.OptimizerParams(1e-4, 1e-3),
                     blob=diee_amd.random_weights(0), train_device="cuda", quiet=True)
    a.train({k: v[:256 * 6] for k, v in mem.items()})          # warm-up (graph capture, MIOpen find)
    torch.cuda.synchronize(); t = time.time()
    losses = a.train(mem)
    torch.cuda.synchronize(); dt = time.time() - t
    print(f"{name:36s} {dt / steps * 1e3:7.2f} ms/step  ({steps} steps, loss {losses[0]:.4f} -> {losses[-1]:.4f})", flush=True)
    if a._graph and os.environ.get("DIEE_DEBUG"):
        g = a._graph
        print("   static ps sum", float(g["ps"].sum()), "oc abs sum", float(g["oc"].abs().sum()), "st abs sum", float(g["st"].abs().sum()), np.round(losses[:4], 4))
