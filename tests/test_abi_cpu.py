"""CPU checks of the drop-in boundary: libdiee.so loads, exports every symbol include/diee.h declares,
and fails loudly (no CPU fallback) when no GPU is present.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import diee_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols(name="diee.h"):
    src = open(os.path.join(ROOT, "include", name)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(diee_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(diee_amd.LIB_PATH):
        import importlib
        importlib.import_module("die-e_amd.build").build()
    L = diee_amd.load_library()
    syms = header_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(L, s), f"{s} is declared in include/diee.h but not exported by libdiee.so"
    assert sorted(diee_amd.EXPORTS) == syms, "die-e_amd.EXPORTS is out of sync with include/diee.h"
    assert b"gfx950" in L.diee_version()
    # development probes live in their own header, not in the boundary
    dev = header_symbols("diee_dev.h")
    assert sorted(diee_amd.DEV_EXPORTS) == dev and all(hasattr(L, s) for s in dev)
    assert not [s for s in syms if s.startswith(("diee_dev_", "diee_probe_"))]


def test_header_compiles_as_c_and_layouts_match_the_ctypes_mirror(tmp_path):
    """tests/abi_check.c: include/diee.h (+ diee_dev.h) compiled as C11 with _Static_asserts on the layouts a Rust
    #[repr(C)] binding relies on; every offset it prints equals the ctypes mirror's"""
    import json
    import subprocess
    exe = str(tmp_path / "abi_check")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "abi_check.c"), "-o", exe])
    doc = json.loads(subprocess.check_output([exe]).decode())
    mirrors = {"diee_stats": diee_amd.Stats, "diee_fragments": diee_amd.Fragments, "diee_batch": diee_amd.Batch,
               "diee_mcts_cfg": diee_amd.MctsConfig}
    seen = 0
    for key, off in doc.items():
        st, field = key.split(".")
        if st == "sizeof":
            if field in mirrors:
                assert C.sizeof(mirrors[field]) == off, key
            elif field == "diee_bg_state":
                assert diee_amd.BG_STATE.itemsize == off
            continue
        assert getattr(mirrors[st], field).offset == off, key
        seen += 1
    assert seen == 35 + 5 + 3 + 1                      # (+ band_flops_demanded, round 6)
    assert [n for n, _ in diee_amd.Stats._fields_] == [k.split(".")[1] for k in doc if k.startswith("diee_stats.")]



def test_rust_binding_in_integration_md_is_generated_from_the_header():
    """INTEGRATION.md's `extern "C"` block is scripts/gen_rust_ffi.py's output for today's include/diee.h: every diee_*
    prototype of the header appears in it with the same arity (counted here by a parser of its own, not the generator's)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_rust_ffi", os.path.join(ROOT, "scripts", "gen_rust_ffi.py"))
    gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = doc[doc.index(gen.BEGIN) + len(gen.BEGIN):doc.index(gen.END)]
    assert block.strip() == ("```rust\n" + gen.generate() + "```").strip(), "run `python scripts/gen_rust_ffi.py --update`"
    # independent arity count: C side = commas at depth 0 of the parameter list, Rust side = `name:` bindings
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "diee.h")).read(), flags=re.S)
    c_arity = {}
    for m in re.finditer(r"\b(diee_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", hdr):
        args = m.group(2).strip()
        c_arity[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    rust_arity = {m.group(1): len(re.findall(r"\b\w+\s*:", m.group(2))) for m in re.finditer(r"pub fn (diee_\w+)\(([^)]*)\)", block)}
    assert sorted(c_arity) == header_symbols() and len(c_arity) >= 29
    assert rust_arity == c_arity
    for st in ("DieeBgState", "DieeMctsCfg", "DieeStats", "DieeFragments", "DieeBatch", "DieeCtx"):
        assert f"pub struct {st}" in block or f"pub enum {st}" in block
    assert block.count("pub games: u64") == 1 and len(re.findall(r"^    pub \w+: (?:u64|f64),$", block, flags=re.M)) >= 27 + 1


def test_state_struct_is_32_bytes_and_matches_the_oracle(oracle):
    assert diee_amd.BG_STATE.itemsize == 32 == oracle.BG_STATE.itemsize
    assert diee_amd.BG_STATE.fields.keys() == oracle.BG_STATE.fields.keys()
    for k in diee_amd.BG_STATE.fields:
        assert diee_amd.BG_STATE.fields[k][1] == oracle.BG_STATE.fields[k][1]


def test_no_gpu_means_loud_failure_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(diee_amd.DieeError) as ei:
        diee_amd.Engine(0)
    assert ei.value.status == diee_amd.ERR_HIP
    # unsupported game id / bad args are status codes, never aborts
    L = diee_amd.load_library()
    h = C.c_void_p()
    assert L.diee_create(0, 7, C.byref(h)) == diee_amd.ERR_UNSUPPORTED           # no such game
    # tic-tac-toe (BASELINE configs[0], "CPU reference path (plumbing, no GPU)") is a host path by definition of the config:
    # its ctx needs no GPU (tests/test_ttt_cpu.py); the backgammon ctx -- the hot path -- has nothing of the kind
    assert L.diee_create(0, diee_amd.GAME_TTT, C.byref(h)) == diee_amd.OK
    L.diee_destroy(h)
    assert L.diee_create(0, 1, None) == diee_amd.ERR_ARG
    assert L.diee_load_weights(None, None, 0) == diee_amd.ERR_ARG


def test_weight_blob_layout_and_random_init_on_host():
    n = diee_amd.weights_count()
    # init 6->256, 19 blocks of two 256->256 convs, heads (~23.58 M parameters + BN stats)
    conv = lambda co, ci: co * ci * 9 + co
    expect = (conv(256, 6) + 4 * 256 + 19 * (2 * conv(256, 256) + 8 * 256)
              + conv(32, 256) + 4 * 32 + 1352 * 768 + 1352 + conv(3, 256) + 12 + 72 + 1)
    assert n == expect
    a = diee_amd.random_weights(0); b = diee_amd.random_weights(0); c = diee_amd.random_weights(1)
    assert a.size == n and (a == b).all() and (a != c).any() and np.isfinite(a).all()
    # tch defaults: conv weights U(+-1/sqrt(fan_in)), conv bias 0, BN gamma U(0,1) beta 0 mean 0 var 1
    w0 = a[:256 * 6 * 9]
    bd = 1 / np.sqrt(54)
    assert abs(w0).max() <= bd and abs(w0).max() > 0.9 * bd and abs(w0.mean()) < 0.01
    assert (a[256 * 6 * 9:256 * 6 * 9 + 256] == 0).all()
    g = a[256 * 6 * 9 + 256:256 * 6 * 9 + 512]
    assert g.min() >= 0 and g.max() <= 1 and 0.4 < g.mean() < 0.6
    assert diee_amd.load_library().diee_random_weights(1, 0, a.ctypes.data, n - 1) == diee_amd.ERR_ARG


def test_product_never_imports_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/"""
    pkg = os.path.join(ROOT, "die-e_amd")
    for dp, _, files in os.walk(pkg):
        if os.sep + "build" in dp:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "diee_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


def _build_cpp_host(tmp_path):
    import subprocess
    exe = str(tmp_path / "self_play")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "self_play.cpp"), "-L", os.path.join(ROOT, "die-e_amd"), "-ldiee",
                           "-Wl,-rpath," + os.path.join(ROOT, "die-e_amd"), "-o", exe])
    return exe


def test_cpp_host_mirror_compiles_links_and_fails_loudly_without_a_gpu(tmp_path):
    """include/diee.hpp (the compiled-language mirror of the reference's interface for the path) + examples/self_play.cpp
    against libdiee.so: what a Rust host's extern "C" binding would do, from C++"""
    import subprocess
    import torch
    exe = _build_cpp_host(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the run is tests/test_host_gpu.py's")
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 1 and b"no HIP device" in p.stderr


def test_every_option_key_is_documented_in_the_header():
    """diee_set_option's keys (the table in csrc/engine.cpp) and the list in include/diee.h's comment say the same names"""
    import re
    src = open(os.path.join(ROOT, "die-e_amd", "csrc", "engine.cpp")).read()
    table = src[src.index("const OptEntry kOptions[]"):src.index("const OptEntry* find_option")]
    keys = re.findall(r'\{"([a-z0-9_]+)",', table)
    assert len(keys) >= 25 and len(set(keys)) == len(keys)
    hdr = open(os.path.join(ROOT, "include", "diee.h")).read()
    doc = hdr[hdr.index("diee_create, as a development override"):hdr.index("diee_status diee_set_option")]
    missing = [k for k in keys if not re.search(r"\b%s\b" % k, doc)]
    assert not missing, missing
