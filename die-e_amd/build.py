"""Builds die-e_amd/libdiee.so for gfx950 with hipcc (in-tree; the .so travels to the GPU box).

    python die-e_amd/build.py [--force]          the product library: fixed compile-time switches, dispatched kernels only
    python die-e_amd/build.py --dev [--force]    die-e_amd/libdiee_dev.so with -DDIEE_DEV_BUILD: the superseded / experimental kernels and
                                                 diee_dev_conv_bench; DIEE_EXTRA_FLAGS=-DDIEE_... then overrides a switch of csrc/nn_common.h
                                                 (DIEE_OUT names another output); scripts/ load it through DIEE_LIB
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
DEV = "--dev" in sys.argv or os.environ.get("DIEE_DEV") == "1"
OUT = os.path.join(HERE, os.environ.get("DIEE_OUT", "libdiee_dev.so" if DEV else "libdiee.so"))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

EXTRA = (["-DDIEE_DEV_BUILD"] if DEV else []) + os.environ.get("DIEE_EXTRA_FLAGS", "").split()      # development (e.g. -DDIEE_TOWER_ABLATE=1: --dev only)
COMMON = EXTRA + ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function",
          f"--offload-arch={ARCH}"]


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", h) for h in ("diee.h", "diee_dev.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    objdir = os.path.join(HERE, "build" if os.path.basename(OUT) == "libdiee.so" else "build_" + os.path.basename(OUT))
    os.makedirs(objdir, exist_ok=True)
    hdr_t = max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith(".h"))
    hdr_t = max([hdr_t] + [os.path.getmtime(os.path.join(HERE, "..", "include", h)) for h in ("diee.h", "diee_dev.h")])
    objs, procs = [], []
    for src in sources():
        obj = os.path.join(objdir, src + ".o")
        objs.append(obj)
        sp = os.path.join(CSRC, src)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(sp), hdr_t):
            continue
        cmd = [HIPCC] + COMMON + ["-x", "hip", "-c", sp, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if out.strip():
            sys.stderr.write(out.decode())
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"hipcc failed on {src}\n")
    if failed:
        raise RuntimeError("build failed")
    cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", OUT] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(OUT)
