"""Writes tests/golden/ot/*: archives produced by LIBTORCH'S OWN C++ serializer (oracle/ot_ref/ot_tool.cpp -- the four calls tch's
C shim makes for VarStore::save / Tensor::save / VarStore::load / Tensor::load, alphazero.rs:149-200,263-265, nnet.rs:109-118)
plus the arrays they hold (expected.npz), so that die-e_amd/ot.py is checked against them where no compiler exists.

    make -C oracle ot_tool && python tests/golden/make_ot_golden.py

Fixtures (data, small): a tic-tac-toe ResNet checkpoint (70 variables, 322 484 floats; the backgammon one is 94 MB and is
round-tripped through the tool by the test itself when the tool is built) whose variables are written in a SHUFFLED order
(tch saves a HashMap's iteration order), and the three training-data archives of a 5-record memory.  The variable names are the
ones die-e_amd/ot.py states for tch (recalled, not pinned); the container format is what this pins."""
import importlib
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
ot = importlib.import_module("die-e_amd.ot")
TOOL = os.path.join(ROOT, "oracle", "_ref", "ot_tool")
DT = {"float32": "f32", "int8": "i8", "int64": "i64"}


def write_manifest(named, stem):
    """[(name, ndarray)] -> stem.man / stem.bin (the container oracle/ot_ref/ot_tool.cpp reads)"""
    off = 0
    with open(stem + ".man", "w") as m, open(stem + ".bin", "wb") as b:
        for name, a in named:
            a = np.ascontiguousarray(a)
            m.write(f"{name} {DT[a.dtype.name]} {a.ndim} " + " ".join(str(d) for d in a.shape) + f"{' ' if a.ndim else ''}{off}\n")
            b.write(a.tobytes()); off += a.nbytes


def read_manifest(stem):
    out = []
    raw = open(stem + ".bin", "rb").read()
    for line in open(stem + ".man"):
        f = line.split()
        nd = int(f[2]); dims = [int(x) for x in f[3:3 + nd]]; off = int(f[3 + nd])
        dt = {v: k for k, v in DT.items()}[f[1]]
        n = int(np.prod(dims)) if dims else 1
        out.append((f[0], np.frombuffer(raw, dtype=dt, count=n, offset=off).reshape(dims).copy()))
    return out


def tool(mode, archive, stem):
    subprocess.check_call([TOOL, mode, archive, stem + ".man", stem + ".bin"])


def main():
    out = os.path.join(HERE, "ot")
    os.makedirs(out, exist_ok=True)
    tmp = os.path.join(out, "_tmp"); os.makedirs(tmp, exist_ok=True)
    rng = np.random.default_rng(20261004)
    blob = rng.standard_normal(ot.weights_count(ot.TICTACTOE_ARCH)).astype(np.float32)
    named = [(n, t.numpy()) for n, t in ot.blob_to_named(blob)]
    order = rng.permutation(len(named))
    write_manifest([named[i] for i in order], os.path.join(tmp, "model"))
    tool("write", os.path.join(out, "ttt_model_libtorch.ot"), os.path.join(tmp, "model"))
    ps = rng.random((5, 1352), dtype=np.float32)
    states = rng.integers(-15, 16, (5, 6, 4, 6)).astype(np.float32)
    outcomes = np.array([1, -1, 0, 1, -1], np.int8)
    for stem, a in (("ps", ps), ("states", states), ("outcomes", outcomes)):
        write_manifest([("0", a)], os.path.join(tmp, stem))
        tool("save0", os.path.join(out, stem + "_libtorch.ot"), os.path.join(tmp, stem))
    np.savez_compressed(os.path.join(out, "expected.npz"), blob=blob, ps=ps, states=states, outcomes=outcomes,
                        names=np.array([n for n, _ in named]), written_order=order)
    for f in os.listdir(tmp):
        os.remove(os.path.join(tmp, f))
    os.rmdir(tmp)
    print({f: os.path.getsize(os.path.join(out, f)) for f in sorted(os.listdir(out))})


if __name__ == "__main__":
    main()
