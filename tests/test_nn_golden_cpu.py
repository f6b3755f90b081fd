"""The network oracle (oracle/nn_ref.py, PyTorch fp32 restatement of nnet.rs) against the committed fixture
tests/golden/nn_golden.npz (made by tests/golden/make_nn_golden.py): the restatement cannot drift silently.
Parity in the reference itself is UNPINNED for network outputs (no reference test constructs a ResNet)."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def test_nn_oracle_reproduces_the_golden_fixture(oracle):
    import diee_amd
    from oracle import nn_ref
    from nn_blobs import bn_nontrivial_blob, bn_slices
    gold = np.load(os.path.join(HERE, "golden", "nn_golden.npz"))
    states = gold["states"].view(oracle.BG_STATE).reshape(-1)
    assert len(states) == 48
    planes = oracle.planes_batch(states)
    base = diee_amd.random_weights(0)
    sl, total = bn_slices()
    assert total == base.size and len(sl) == 1 + 2 * 19 + 2
    for name, blob in (("init", base), ("bn", bn_nontrivial_blob(base))):
        pol, val, logits = nn_ref.forward_t(nn_ref.parse(blob), planes)
        # same arithmetic, possibly another BLAS / thread count: a few ulps of fp32 drift over 40 layers
        assert np.abs(logits - gold[f"{name}_logits"]).max() < 2e-4 * max(1.0, np.abs(gold[f"{name}_logits"]).max())
        assert np.abs(val - gold[f"{name}_value"]).max() < 1e-4
        assert (pol.argmax(1) == gold[f"{name}_policy_argmax"]).mean() > 0.95
    # the two blobs really differ where it matters (BN folding is exercised)
    assert np.abs(gold["bn_logits"] - gold["init_logits"]).max() > 1e-2
