"""cbc_amd -- MI355X-native hot path of the cbc aligned-read compressor.

Only what the hot path needs lives here: csrc/ (HIP kernels + C ABI + C host packer) and thin
ctypes views of the two libraries.  The arithmetic coding runs on the GPU or not at all.
"""
from . import host  # noqa: F401
from . import gpu  # noqa: F401

__all__ = ["host", "gpu"]
