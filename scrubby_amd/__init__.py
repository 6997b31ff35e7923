"""scrubby_amd — MI355X-native host-depletion backend for Scrubby's `mm2` aligner path.

The product is libscrubby_hip.so (scrubby_amd/csrc, C ABI in include/scrubby_hip.h);
this package holds its ctypes binding and the host-side mirror of the reference interface.
"""
from . import lib  # noqa: F401
