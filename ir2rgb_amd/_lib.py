"""ctypes binding of libir2rgb_hip.so -- the only way the Python side reaches the kernels.

The library is loaded lazily on first use and the load FAILS LOUDLY when the shared object is
missing (no CPU or eager fallback exists in this package by design).  Prototypes mirror
include/ir2rgb_hip.h one to one; tests/test_abi.py checks that every symbol the header declares
is exported and bound here.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libir2rgb_hip.so")

c_int, c_long, c_float, c_void_p = ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_void_p
_pint = ctypes.POINTER(ctypes.c_int)
P = c_void_p  # device pointer

class ConvDesc(ctypes.Structure):
    """ir2rgb_conv_desc of include/ir2rgb_hip.h."""
    _fields_ = [(n, c_int) for n in ("N", "Hin", "Win", "Cin", "Hout", "Wout", "Cout", "kh", "kw", "stride_h", "stride_w",
                                     "pad_h", "pad_w", "pad_mode", "transposed", "dtype", "act", "out_f32",
                                     "ldx", "ci_off", "ldy", "co_off", "stats_per_sample")]


_pdesc = ctypes.POINTER(ConvDesc)

LOSS_MAX_ITEMS = 32


class LossItem(ctypes.Structure):
    """ir2rgb_loss_item of include/ir2rgb_hip.h."""
    _fields_ = [("a", c_void_p), ("b", c_void_p), ("ga", c_void_p), ("mask", c_void_p), ("n", c_long), ("hw", c_long),
                ("chw", c_long), ("weight", c_float), ("target", c_float), ("kind", c_int), ("slot", c_int)]


_pitem = ctypes.POINTER(LossItem)


class PackJob(ctypes.Structure):
    """ir2rgb_pack_job of include/ir2rgb_hip.h."""
    _fields_ = [("desc", ConvDesc), ("w", c_void_p), ("wpacked", c_void_p), ("adjoint", c_int), ("reserved", c_int)]


_pjob = ctypes.POINTER(PackJob)

# name -> (restype, argtypes)
PROTOTYPES = {
    "ir2rgb_version": (ctypes.c_char_p, []),
    "ir2rgb_correlation_out_shape": (c_int, [c_int] * 8 + [_pint] * 3),
    "ir2rgb_correlation_fwd": (c_int, [P, P, P] + [c_int] * 9 + [P]),
    "ir2rgb_gather_f32": (c_int, [P, P, P, c_long, P]),
    "ir2rgb_avgpool3s2": (c_int, [P, P, c_long, c_int, c_int, c_int, P]),
    "ir2rgb_correlation_nhwc_half": (c_int, [P, c_int, c_int, P, c_int, c_int, P, c_int, c_int, c_int, c_float] + [c_int] * 5 + [P]),
    "ir2rgb_correlation_bwd": (c_int, [P, P, P, P, P] + [c_int] * 9 + [P]),
    "ir2rgb_resample2d_fwd": (c_int, [P, P, P] + [c_int] * 5 + [P]),
    "ir2rgb_resample2d_bwd": (c_int, [P, P, P, P, P] + [c_int] * 5 + [P]),
    "ir2rgb_channelnorm_fwd": (c_int, [P, P] + [c_int] * 5 + [P]),
    "ir2rgb_channelnorm_bwd": (c_int, [P, P, P, P] + [c_int] * 5 + [P]),
    "ir2rgb_warp_diff_norm_fwd": (c_int, [P] * 6 + [c_int] * 4 + [P]),
    "ir2rgb_conv2d_packed_weight_elems": (c_long, [_pdesc]),
    "ir2rgb_conv2d_stats_rows": (c_int, [_pdesc]),
    "ir2rgb_conv2d_pack_weight": (c_int, [_pdesc, P, P, P]),
    "ir2rgb_conv2d_pack_weight_adjoint": (c_int, [_pdesc, P, P, P]),
    "ir2rgb_conv2d_pack_batch_table_bytes": (c_long, [_pjob, c_int]),
    "ir2rgb_conv2d_pack_batch_build": (c_int, [_pjob, c_int, P, c_long, _pint]),
    "ir2rgb_conv2d_pack_batch_run": (c_int, [P, c_int, c_int, c_int, P]),
    "ir2rgb_conv2d_fwd": (c_int, [_pdesc, P, P, P, P, P, P]),
    "ir2rgb_conv2d_fwd_workspace_bytes": (c_long, [_pdesc]),
    "ir2rgb_conv2d_fwd_ws": (c_int, [_pdesc, P, P, P, P, P, P, c_long, P]),
    "ir2rgb_conv2d_kernel_name": (ctypes.c_char_p, [_pdesc]),
    "ir2rgb_bn_finalize": (c_int, [P, c_int, c_int, c_long, P, P, P, P, c_float, c_float, P, P, P, P, c_int, P]),
    "ir2rgb_bn_finalize_ex": (c_int, [P, c_int, c_int, c_long, P, P, P, P, P, c_float, c_float, P, P, P, P, c_int, c_int, P]),
    "ir2rgb_bn_finalize_apply": (c_int, [P, c_int, c_int, c_long, P, P, P, P, P, c_float, c_float, P, P, P, P, c_int,
                                         P, P, P, P, c_long, c_int, c_int, P]),
    "ir2rgb_bn_apply": (c_int, [P, P, P, P, P, P, c_long, c_int, c_int, c_int, P]),
    "ir2rgb_nchw_f32_to_nhwc_half": (c_int, [P, P] + [c_int] * 5 + [P]),
    "ir2rgb_nhwc_half_to_nchw_f32": (c_int, [P, P] + [c_int] * 5 + [P]),
    "ir2rgb_xexpand": (c_int, [P, P] + [c_int] * 10 + [P]),
    "ir2rgb_xexpand_cx": (c_int, [P, P] + [c_int] * 11 + [P]),
    "ir2rgb_nchw_f32_to_nhwc_half_slice": (c_int, [P, P] + [c_int] * 8 + [P]),
    "ir2rgb_flow_upsample_slice": (c_int, [P, P, P, P] + [c_int] * 6 + [P]),
    "ir2rgb_head_finish": (c_int, [P, P, P] + [c_int] * 7 + [ctypes.c_uint, c_float, P]),
    "ir2rgb_warp_blend_fwd": (c_int, [P] * 6 + [c_int] * 4 + [P]),
    "ir2rgb_bn_bwd_blocks": (c_int, [c_long, c_int]),
    "ir2rgb_bn_bwd": (c_int, [P] * 10 + [c_long, c_int, c_int, c_int, P]),
    "ir2rgb_thin_grad_expand": (c_int, [P, P, P, P] + [c_int] * 5 + [P]),
    "ir2rgb_fold_reflect": (c_int, [P, P] + [c_int] * 7 + [P]),
    "ir2rgb_head_finish_bwd": (c_int, [P, P, P, P, P] + [c_int] * 7 + [ctypes.c_uint, c_float, c_int, P]),
    "ir2rgb_head_finish_bwd_rows": (c_int, [c_int, c_int, c_int]),
    "ir2rgb_warp_blend_bwd": (c_int, [P] * 8 + [c_int] * 4 + [P]),
    "ir2rgb_xexpand_bwd": (c_int, [P, P] + [c_int] * 10 + [P]),
    "ir2rgb_conv2d_wgrad_workspace_elems": (c_long, [_pdesc]),
    "ir2rgb_conv2d_wgrad_acc_workspace_elems": (c_long, [_pdesc]),
    "ir2rgb_conv2d_wgrad": (c_int, [_pdesc, P, P, P, P, P]),
    "ir2rgb_conv2d_wgrad_acc": (c_int, [_pdesc, P, P, P, P, P]),
    "ir2rgb_loss_partial_elems": (c_int, []),
    "ir2rgb_loss_multi_fwd": (c_int, [_pitem, c_int, c_int, P, P, P]),
    "ir2rgb_loss_multi_bwd": (c_int, [_pitem, c_int, c_int, P, P]),
    "ir2rgb_adam_chunk_elems": (c_int, []),
    "ir2rgb_adam_step": (c_int, [P, P, c_int, c_float, c_float, c_float, c_float, c_int, P]),
}

_lib = None


class Ir2rgbError(RuntimeError):
    """A libir2rgb_hip.so entry point returned non-zero (mirrors the reference's AT_ERROR
    "CUDA call failed", correlation_cuda.cc:80-84)."""


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -m ir2rgb_amd.build` or __graft_entry__.build()). "
                "ir2rgb_amd has no CPU/eager fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        # fastcall wrappers (ir2rgb_amd/fastbind.py: ~0.5 us per call instead of ctypes' 6-9 us) for every entry point they
        # cover; the ctypes functions for the rest and when the extension is not built
        from . import fastbind
        fast = fastbind.load()
        ns = _Entry()
        ns.ctypes_handle = handle
        ns.fast_module = fast
        for name in PROTOTYPES:
            setattr(ns, name, getattr(fast, name, None) or (getattr(handle, name) if fast is not None else _tensor_shim(getattr(handle, name))))
        _lib = ns
    return _lib


def _tensor_shim(fn):
    """ctypes function that, like the fastcall wrappers, takes tensors where pointers go (IR2RGB_FASTBIND=0 / extension
    not built: the slow, reference binding)."""
    def call(*args):
        return fn(*[a.data_ptr() if hasattr(a, "data_ptr") else a for a in args])
    call.__name__ = getattr(fn, "__name__", "ir2rgb_entry")
    return call


class _Entry:
    """Namespace of the library's entry points (fastcall wrapper where one exists, else the ctypes function)."""


_ERRNAMES = {-1: "IR2RGB_EINVAL (bad size/parameter)", -2: "IR2RGB_ENOSUP (not supported)",
             -3: "IR2RGB_EALIGN (misaligned buffer)"}


def check(rc, what):
    if rc != 0:
        if rc < 0:
            if rc == -2:
                raise NotImplementedError(f"{what}: {_ERRNAMES[rc]}")
            raise ValueError(f"{what}: {_ERRNAMES.get(rc, rc)}")
        raise Ir2rgbError(f"{what}: HIP call failed (hipError_t {rc})")


_RAW_STREAM = None


def current_stream(tensor):
    """hipStream_t of torch's current stream on the tensor's device (fast path: one C call)."""
    global _RAW_STREAM
    if _RAW_STREAM is None:
        import torch
        _RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (
            lambda idx: torch.cuda.current_stream(idx).cuda_stream)
    return _RAW_STREAM(tensor.device.index)     # (an int: see ir2rgb_amd/fastbind.py)


class _NullCtx:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NULL = _NullCtx()
_GET_DEVICE = None


def on_device(tensor):
    """Context that makes the tensor's device current -- a no-op (no Python-level device switch) in the
    one-process-per-GPU deployment where it already is."""
    global _GET_DEVICE
    if _GET_DEVICE is None:
        import torch
        _GET_DEVICE = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device
    if tensor.device.index == _GET_DEVICE():
        return _NULL
    import torch
    return torch.cuda.device_of(tensor)


def require_device(*tensors, dtype=None):
    """Turn the reference's silent assumptions (CUDA device, dtype, contiguity) into errors."""
    import torch
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not isinstance(t, torch.Tensor):
            raise TypeError(f"expected a tensor, got {type(t)}")
        if not t.is_cuda:
            raise ValueError("ir2rgb_amd operators run on an AMD GPU only: got a CPU tensor "
                             "(there is no CPU fallback; use oracle/ for CPU checks)")
        if dtype is not None and t.dtype != dtype:
            raise TypeError(f"expected dtype {dtype}, got {t.dtype}")
        if not t.is_contiguous():
            raise ValueError("expected a contiguous tensor")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise ValueError(f"tensors on different devices: {dev} vs {t.device}")
    return dev
