import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Order of the -m gpu tier (the driver runs it with -x): the hot path's parity evidence first -- rules, search, tail, the BASELINE-size
# parity runs, the network, then errors / delivery / full sizes / ranks --, the callers either side of the path (training
# kernels, learn loop, arena, CLI) after it, and the five-minute full-size learn loop very last: a flake in an (f) row can no longer
# leave an (a) row unreached.  Files not listed keep their alphabetical place between the two groups.
_GPU_ORDER = ["test_rules_gpu", "test_search_gpu", "test_tail_gpu", "test_free_gpu", "test_parity_baseline_sizes_gpu", "test_nn_gpu", "test_ttt_gpu",
              "test_errors_gpu", "test_delivery_gpu", "test_fullsize_gpu", "test_dist_gpu"]
_GPU_LAST = ["test_train_gpu", "test_host_gpu"]
_VERY_LAST = ["test_learn_loop_config5_full_size"]


def pytest_collection_modifyitems(session, config, items):
    def key(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        if item.name.split("[")[0] in _VERY_LAST:
            return (3, 0)
        if mod in _GPU_ORDER:
            return (0, _GPU_ORDER.index(mod))
        if mod in _GPU_LAST:
            return (2, _GPU_LAST.index(mod))
        return (1, 0)
    items.sort(key=key)                                    # stable: the order inside a file (and of unlisted files) stays


@pytest.fixture(scope="session")
def oracle():
    """the CPU oracle (test infrastructure): builds oracle/libdiee_oracle.so on first use"""
    from oracle import oracle as orc
    orc.build()
    orc.lib()
    return orc
