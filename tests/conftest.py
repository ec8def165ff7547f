import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib_built():
    """Make sure libea_mi355x.so exists (hipcc cross-compiles without a GPU)."""
    from easyanimate_amd import build
    return build.build(verbose=False)


@pytest.fixture(scope="session")
def rccl_world1():
    """Results of tests/_rccl_world1.py (one spawned process, one RCCL initialisation, both world-of-one checks)."""
    import socket
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _rccl_world1
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    import torch
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rccl_world1.run, args=(port, ret), nprocs=1, join=True)
    out = dict(ret)
    for k in ("sp", "vae"):
        if k + "_error" in out:
            raise AssertionError(out[k + "_error"])
    return out
