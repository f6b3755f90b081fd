"""libtorch `.ot` archives <-> the engine's flat weight blob / numpy arrays.

What the reference writes (src/alphazero/alphazero.rs:149-200,263-265; src/alphazero/nnet.rs:109-118):

  * models: `VarStore::save(path)` -- tch 0.13 (Cargo.toml:10, unvendored) hands the named variables to
    `torch::serialize::OutputArchive::write(name, tensor)` and `save_to(path)`: a TorchScript zip whose module holds one
    parameter per variable.  Every layer of the ResNet is created on the ROOT path (nnet.rs:62-97), so the names carry no
    layer prefix: the first of each kind is `weight` / `bias` / `running_mean` / `running_var`, every later one gets
    `__<number of variables registered so far>` appended ([unvendored, from memory]: tch `Path::var`).
  * training data: `Tensor::save(path)` = `torch::save(tensor, path)`: the same kind of archive with the single key "0"
    (`ps.ot [M,1352]`, `states.ot [M,6,4,6]`, `outcomes.ot [M]` i8).

PINNED TO LIBTORCH'S OWN SERIALIZER (round 5): `oracle/ot_ref/ot_tool.cpp` (test infrastructure) makes exactly the four
libtorch calls tch's C shim makes -- `OutputArchive::write(name, tensor)` + `save_to`, `torch::save(tensor)`,
`jit::load(..).named_parameters()`, `torch::load(tensor)` -- against the libtorch inside this image's PyTorch wheel;
`tests/test_host_cpu.py` reads archives that program wrote (small ones committed under `tests/golden/ot/`) bit-exactly and has
that program read what this module writes.  So the CONTAINER FORMAT is pinned in both directions.

STILL RECALLED, NOT PINNED: the reference ships no `.ot` file and tch is not vendored, so the variable NAMES tch generates
(`weight`, `bias`, `weight__N`) and the within-layer creation order (weight-before-bias or the reverse) cannot be checked
here.  Reading is therefore tolerant: tensors are ordered by their `__N` suffix (creation order) and matched to the
architecture by base name and shape, whichever order a layer's variables were created in and whatever order the archive lists
them in (tch saves a HashMap's iteration order).  Writing uses the order stated above.
"""
import re

import numpy as np
import torch

# (filters, blocks, actions, input planes, board cells): Backgammon (backgammon_logic.rs:74-78) and TicTacToe (tictactoe/mod.rs:20-24)
BACKGAMMON_ARCH = (256, 19, 1352, 6, 24)
TICTACTOE_ARCH = (64, 4, 9, 3, 9)
ARCHS = (BACKGAMMON_ARCH, TICTACTOE_ARCH)


def _layers(arch=BACKGAMMON_ARCH):
    """the layers in creation order (nnet.rs:62-97; ResBlock::new nnet.rs:38-45: conv1, conv2, bn1, bn2)"""
    F, BLOCKS, A, CIN, HW = arch
    L = [("conv", F, CIN), ("bn", F)]
    for _ in range(BLOCKS):
        L += [("conv", F, F), ("conv", F, F), ("bn", F), ("bn", F)]
    L += [("conv", 32, F), ("bn", 32), ("fc", A, 32 * HW), ("conv", 3, F), ("bn", 3), ("fc", 1, 3 * HW)]
    return L


def weights_count(arch):
    return sum(int(np.prod(shape)) for layer in _layers(arch) for _, shape in _layer_tensors(layer))


def arch_of_blob(blob):
    for arch in ARCHS:
        if weights_count(arch) == np.size(blob):
            return arch
    raise ValueError(f"a blob of {np.size(blob)} floats is neither game's ResNet")


def _layer_tensors(layer):
    """(blob order of the layer's tensors as (base name, shape))  -- blob order of include/diee.h"""
    kind = layer[0]
    if kind == "conv":
        return [("weight", (layer[1], layer[2], 3, 3)), ("bias", (layer[1],))]
    if kind == "fc":
        return [("weight", (layer[1], layer[2])), ("bias", (layer[1],))]
    c = layer[1]
    return [("weight", (c,)), ("bias", (c,)), ("running_mean", (c,)), ("running_var", (c,))]


def _creation_order(layer):
    """order in which tch creates a layer's variables ([unvendored, from memory]): conv2d / linear register `bias`
    before `weight` when bias is enabled; batch_norm registers weight, bias, running_mean, running_var"""
    t = _layer_tensors(layer)
    return [t[1], t[0]] if layer[0] in ("conv", "fc") else t


class _Archive(torch.nn.Module):
    pass


def _save_named(named, path):
    m = _Archive()
    for name, t in named:
        m.register_parameter(name, torch.nn.Parameter(t.contiguous(), requires_grad=False))
    torch.jit.save(torch.jit.script(m), path)


def _load_named(path):
    m = torch.jit.load(path, map_location="cpu")
    out = [(n, p.detach()) for n, p in m.named_parameters()] + [(n, b.detach()) for n, b in m.named_buffers()]
    if not out:
        raise ValueError(f"{path}: no tensors in the archive")
    return out


# --------------------------------------------------------------------------- models
def blob_to_named(blob):
    """flat fp32 blob (include/diee.h order; either game's network, told by its size) -> [(VarStore name, tensor)] in creation order"""
    blob = np.ascontiguousarray(blob, dtype=np.float32)
    named, seen, off = [], set(), 0
    for layer in _layers(arch_of_blob(blob)):
        parts = {}
        for base, shape in _layer_tensors(layer):
            n = int(np.prod(shape))
            parts[base] = torch.from_numpy(blob[off:off + n].reshape(shape).copy()); off += n
        for base, _ in _creation_order(layer):
            name = base if base not in seen else f"{base}__{len(named)}"
            seen.add(base)
            named.append((name, parts[base]))
    assert off == blob.size, (off, blob.size)
    return named


def named_to_blob(named, arch=None):
    """[(name, tensor)] of a VarStore archive -> flat fp32 blob; tolerant to the within-layer creation order and to the order the
    archive lists its tensors in.  arch None: told by the number of tensors (250 backgammon, 70 tic-tac-toe)"""
    if arch is None:
        arch = next((a for a in ARCHS if sum(len(_layer_tensors(l)) for l in _layers(a)) == len(named)), BACKGAMMON_ARCH)

    def key(item):
        m = re.fullmatch(r"(.+?)__(\d+)", item[0])
        return int(m.group(2)) if m else -1          # un-suffixed names were created first (one per base name)
    entries = [(re.sub(r"__\d+$", "", n), t) for n, t in sorted(named, key=key)]
    used = [False] * len(entries)
    out = []
    cursor = 0
    for layer in _layers(arch):
        want = _layer_tensors(layer)
        got = {}
        # the layer's variables are the next len(want) unused entries in creation order; match them by name and shape
        idx = [i for i in range(cursor, len(entries)) if not used[i]][:len(want) * 2]
        for base, shape in want:
            for i in idx:
                if not used[i] and entries[i][0] == base and tuple(entries[i][1].shape) == tuple(shape):
                    got[base] = entries[i][1]; used[i] = True
                    break
            else:
                raise ValueError(f"archive does not match the ResNet: no `{base}` of shape {shape} for layer {layer}")
        while cursor < len(entries) and used[cursor]:
            cursor += 1
        for base, shape in want:
            out.append(got[base].to(torch.float32).reshape(-1))
    if not all(used):
        raise ValueError(f"{used.count(False)} tensors of the archive were not consumed")
    return torch.cat(out).numpy()


def save_model_ot(blob, path):
    """VarStore::save (alphazero.rs:263-265)"""
    _save_named(blob_to_named(blob), path)


def load_model_ot(path):
    """VarStore::load (nnet.rs:109-118, alphazero.rs:81-100) -> flat fp32 blob"""
    return named_to_blob(_load_named(path))


def load_model(path):
    """`.ot` archive or `.npy` blob, by extension"""
    return load_model_ot(path) if str(path).endswith(".ot") else np.load(path)


# --------------------------------------------------------------------------- single tensors (training data)
def save_tensor_ot(array, path):
    """Tensor::save (alphazero.rs:169-171): key "0" """
    _save_named([("0", torch.from_numpy(np.ascontiguousarray(array)))], path)


def load_tensor_ot(path):
    named = _load_named(path)
    return named[0][1].numpy()


def convert_data_dir(src, dst, to="ot"):
    """a self-play data directory (`ps`, `states`, `outcomes`) between .npy and .ot"""
    import os
    os.makedirs(dst, exist_ok=True)
    for stem, dtype in (("ps", np.float32), ("states", np.float32), ("outcomes", np.int8)):
        if to == "ot":
            save_tensor_ot(np.load(os.path.join(src, stem + ".npy")).astype(dtype), os.path.join(dst, stem + ".ot"))
        else:
            np.save(os.path.join(dst, stem + ".npy"), load_tensor_ot(os.path.join(src, stem + ".ot")).astype(dtype))
