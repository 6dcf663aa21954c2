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


def test_fastcall_bindings_cover_the_prototypes_and_agree_with_ctypes(built_lib):
    """ir2rgb_amd/fastbind.py: the CPython fastcall bindings are generated from _lib.PROTOTYPES (which the test above holds
    against the header), built next to the library, current (signature of the prototypes they were generated from), and
    hand back what the ctypes functions do -- checked on host-only entry points and on argument validation that returns
    before any launch (no GPU here)."""
    import ctypes
    import torch
    from ir2rgb_amd import conv as C, fastbind
    assert os.path.exists(fastbind.so_path())
    lib = _lib.lib()
    fast, handle = lib.fast_module, lib.ctypes_handle
    assert fast is not None and fast.SIGNATURE == fastbind.generate()[1]
    covered = set(fastbind.wrappable())
    assert covered == {n for n in _lib.PROTOTYPES if hasattr(fast, n)}
    # everything but the three entry points that return strings or write through int* is covered
    assert set(_lib.PROTOTYPES) - covered == {"ir2rgb_version", "ir2rgb_conv2d_kernel_name", "ir2rgb_correlation_out_shape",
                                              "ir2rgb_conv2d_pack_batch_build"}
    for name in covered:
        assert getattr(lib, name) is getattr(fast, name)
    d = C.make_desc((1, 1024, 32, 64), 1024, 3, 1, 1, C.PAD_REFLECT, torch.bfloat16)
    assert d is C.make_desc((1, 1024, 32, 64), 1024, 3, 1, 1, C.PAD_REFLECT, torch.bfloat16) and d._addr == ctypes.addressof(d)
    for fn in ("ir2rgb_conv2d_stats_rows", "ir2rgb_conv2d_fwd_workspace_bytes", "ir2rgb_conv2d_wgrad_workspace_elems",
               "ir2rgb_conv2d_packed_weight_elems"):
        want = getattr(handle, fn)(ctypes.byref(d))
        assert getattr(fast, fn)(d) == want                        # descriptor object (_addr)
        assert getattr(fast, fn)(d._addr) == want                  # plain address
        assert getattr(fast, fn)(ctypes.byref(d)) == want          # ctypes reference (slow path)
    assert fast.ir2rgb_bn_bwd_blocks(2048, 1024) == handle.ir2rgb_bn_bwd_blocks(2048, 1024) > 0
    assert fast.ir2rgb_head_finish_bwd_rows(1, 64, 128) == handle.ir2rgb_head_finish_bwd_rows(1, 64, 128)
    t = torch.zeros(64)
    p = t.data_ptr()
    # invalid arguments are refused by the library before anything is launched: same code through both bindings
    args = (p, p, p, p, p, p, p, p, p, p, 100, 63, 0, 1, None)
    assert fast.ir2rgb_bn_bwd(*args) == handle.ir2rgb_bn_bwd(*args) == -1
    assert fast.ir2rgb_bn_bwd(ctypes.c_void_p(p), *args[1:]) == -1
    fargs = (None, 0, 0, 1, None, None, None, None, None, 0.1, 1e-5, None, None, None, None, 1, 0, None)
    assert fast.ir2rgb_bn_finalize_ex(*fargs) == handle.ir2rgb_bn_finalize_ex(*fargs) == -1
    with pytest.raises(TypeError, match="takes 15 arguments"):
        fast.ir2rgb_bn_bwd(p, p)
    with pytest.raises(TypeError):
        fast.ir2rgb_bn_bwd("x", *args[1:])
    with pytest.raises(OverflowError):
        fast.ir2rgb_bn_bwd(*args[:11], 1 << 40, 0, 1, None)


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
