"""Process-wide default handle (one device, torch's current stream)."""
import os

from . import _lib

_default = None


def local_device():
    """cuda index for this process: FVGP_DEVICE if set, else LOCAL_RANK under torchrun, else 0."""
    return int(os.environ.get("FVGP_DEVICE", os.environ.get("LOCAL_RANK", "0")))


def default_handle():
    global _default
    if _default is None:
        _default = _lib.Handle(local_device())
    return _default
