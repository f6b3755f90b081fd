#!/usr/bin/env python3
"""Generates tests/golden/nn_golden.npz: fp32 network outputs of oracle/nn_ref.py (the PyTorch restatement of
src/alphazero/nnet.rs:24-34,57-133) on 48 seeded backgammon states, for the seed-0 random-init blob and for a blob
with non-trivial BatchNorm statistics.  The reference pins no network output (no test constructs a
ResNet) and cannot be built here, so this fixture pins the RESTATEMENT against silent drift: tests/test_nn_golden_cpu.py
re-derives it on the CPU, tests/test_nn_gpu.py holds the bf16 engine to it at the stated tolerance.

    python tests/golden/make_nn_golden.py        (CPU, a few seconds; needs die-e_amd/libdiee.so for the blob generator)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    torch.set_num_threads(8)
    import diee_amd
    from oracle import oracle as orc
    from oracle import nn_ref
    from nn_blobs import bn_nontrivial_blob
    orc.build()
    walk = orc.random_walk_states(2024, 6)
    states = walk[np.linspace(0, len(walk) - 1, 48).astype(int)]
    planes = orc.planes_batch(states)
    out = {"states": states.view(np.uint8).reshape(-1, 32)}
    base = diee_amd.random_weights(0)
    for name, blob in (("init", base), ("bn", bn_nontrivial_blob(base))):
        pol, val, logits = nn_ref.forward_t(nn_ref.parse(blob), planes)
        out[f"{name}_logits"] = logits.astype(np.float32)
        out[f"{name}_value"] = val.astype(np.float32)
        out[f"{name}_policy_max"] = pol.max(1).astype(np.float32)
        out[f"{name}_policy_argmax"] = pol.argmax(1).astype(np.int32)
    path = os.path.join(ROOT, "tests", "golden", "nn_golden.npz")
    np.savez_compressed(path, **out)
    print(path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
