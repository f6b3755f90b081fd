"""Weight blobs for the network parity tests (layout of include/diee.h, creation order of nnet.rs:62-97)."""
import numpy as np

F, BLOCKS, A = 256, 19, 1352


def bn_slices():
    """(gamma, beta, mean, var) slices of every BatchNorm in the blob, in blob order"""
    out, off = [], 0

    def conv(cout, cin):
        nonlocal off
        off += cout * cin * 9 + cout

    def bn(c):
        nonlocal off
        out.append(tuple(slice(off + i * c, off + (i + 1) * c) for i in range(4)))
        off += 4 * c
    conv(F, 6); bn(F)
    for _ in range(BLOCKS):
        conv(F, F); conv(F, F); bn(F); bn(F)
    conv(32, F); bn(32); off += A * 768 + A
    conv(3, F); bn(3); off += 72 + 1
    return out, off


def bn_nontrivial_blob(base, seed=1234):
    """`base` with every BatchNorm given non-trivial statistics, as a trained checkpoint has them: beta, running mean
    ~ N(0, 0.1), running var ~ U(0.5, 1.5), gamma ~ U(0.5, 1.5) -- exercises the BN folding the random-init blob
    (beta 0, mean 0, var 1) leaves untested"""
    sl, total = bn_slices()
    assert total == len(base), (total, len(base))
    b = np.array(base, dtype=np.float32, copy=True)
    rng = np.random.default_rng(seed)
    for g, be, m, v in sl:
        n = g.stop - g.start
        b[g] = rng.uniform(0.5, 1.5, n).astype(np.float32)
        b[be] = rng.normal(0.0, 0.1, n).astype(np.float32)
        b[m] = rng.normal(0.0, 0.1, n).astype(np.float32)
        b[v] = rng.uniform(0.5, 1.5, n).astype(np.float32)
    return b
