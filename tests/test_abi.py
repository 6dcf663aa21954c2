"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/*.h declares, and the Python bindings cover exactly that set.  No compute calls."""
import glob
import os
import re
import subprocess

import pytest

from ir2rgb_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = open(h).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(ir2rgb_[a-z0-9_]+)\s*\(", text))
    return names


@pytest.fixture(scope="module")
def built_lib():
    from ir2rgb_amd import build
    return build.build()


def test_header_declares_something():
    assert len(declared_symbols()) >= 9


def test_library_exports_every_declared_symbol(built_lib):
    out = subprocess.check_output(["nm", "-D", "--defined-only", built_lib], text=True)
    exported = set(re.findall(r"\bT (ir2rgb_[a-z0-9_]+)", out))
    missing = declared_symbols() - exported
    assert not missing, f"declared in include/ but not exported: {sorted(missing)}"


def test_bindings_cover_the_header(built_lib):
    assert set(_lib.PROTOTYPES) == declared_symbols()
    handle = _lib.lib()  # dlopen + prototype binding; raises if a symbol is absent
    assert handle.ir2rgb_version().startswith(b"ir2rgb_hip")


def test_out_shape_matches_oracle(built_lib):
    from ir2rgb_amd.ext import correlation_cuda
    from oracle import ops
    for args in [(256, 64, 128, 20, 1, 20, 1, 2), (8, 24, 32, 21, 3, 20, 2, 3), (4, 7, 9, 4, 1, 4, 1, 1)]:
        assert correlation_cuda.out_shape(*args) == ops.correlation_out_shape(*args)


def test_cpu_tensors_are_rejected_loudly(built_lib):
    import torch
    from ir2rgb_amd.flownet2_pytorch.networks.channelnorm_package.channelnorm import ChannelNorm
    from ir2rgb_amd.flownet2_pytorch.networks.correlation_package.correlation import Correlation
    from ir2rgb_amd.flownet2_pytorch.networks.resample2d_package.resample2d import Resample2d
    with pytest.raises(ValueError, match="no CPU fallback"):
        ChannelNorm()(torch.zeros(1, 3, 4, 4))
    with pytest.raises(ValueError, match="no CPU fallback"):
        Resample2d()(torch.zeros(1, 3, 4, 4), torch.zeros(1, 2, 4, 4))
    with pytest.raises(ValueError, match="no CPU fallback"):
        Correlation(20, 1, 20, 1, 2, 1)(torch.zeros(1, 3, 4, 4), torch.zeros(1, 3, 4, 4))


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU/eager fallback"):
        _lib.lib()


def test_lds_dma_pipelines_wait_for_their_reads_before_each_barrier(built_lib):
    """Every kernel that stages operands by LDS-DMA restages a ring stage one barrier after its last fragment read; that is
    only safe if the reading wave has waited for those reads BEFORE the barrier (a raw s_barrier is no fence, and hipcc
    sinks the wait below it when it can).  tools/check_lds_war.py walks the gfx950 code of the built objects: no barrier may
    be reachable with LDS reads outstanding.  (Round 3: this was the cause of the two-stream first-forward corruption.)"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_lds_war
    res = check_lds_war.check()
    assert sum(k for _, k, _ in res) >= 100, "the scan did not find the LDS-DMA kernels"
    bad = {name: hits[:3] for _, _, f in res for name, hits in f.items()}
    assert not bad, f"LDS reads outstanding at a barrier: {bad}"
