import os
import sys

import numpy as np
import pytest
import torch  # noqa: F401  (first: torch bundles its own HIP runtime, which must be the one the process loads)

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """The two 50M-variable cases (config #5 at its stated size: ~100 GB of host memory, minutes of generation)
    run last: with -x, a box that cannot hold them must not hide the rows behind them."""
    heavy = [it for it in items if "test_config5_gpu.py" in it.nodeid or "lr50m" in it.nodeid]
    if heavy:
        items[:] = [it for it in items if it not in heavy] + heavy


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


def graph_from(npz, tag):
    """(weight, variable, factor, fmap, domain_mask, edges) stored by tools/make_goldens.py"""
    return (npz[tag + "_in_weight"], npz[tag + "_in_variable"], npz[tag + "_in_factor"],
            npz[tag + "_in_fmap"], npz[tag + "_in_domain_mask"], int(npz[tag + "_in_edges"]))
