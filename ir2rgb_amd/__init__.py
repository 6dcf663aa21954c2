"""ir2rgb_amd -- MI355X-native (gfx950) implementation of the ir2rgb vid2vid hot path.

Everything computational goes through ``lib/libir2rgb_hip.so`` (hand-written HIP, C ABI in
include/ir2rgb_hip.h).  Importing the package never touches the GPU; the library is loaded on
first use and its absence is an error (there is no CPU fallback).
"""
__version__ = "0.1.0"
