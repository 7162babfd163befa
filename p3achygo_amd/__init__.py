"""p3achygo_amd — MI355X-native policy/value-net engine for p3achygo's self-play path.

The product is the C-ABI shared library `csrc/libp3hip.so` (include/p3hip.h); this package
is the thin Python mirror of the reference's `nn::Engine` surface used by tests, bench.py
and __graft_entry__.  There is no CPU fallback: importing `engine` without the built
extension, or creating an engine without a gfx950 device, raises.
"""
from . import netspec  # noqa: F401
