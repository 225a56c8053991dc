"""roomnet_amd -- MI355X (gfx950) implementation of RoomNet's forward-pass inference path.

Drop-in surface of the reference (ironhide23586/RoomNet):
  ``roomnet_amd.network.RoomNet``     <- reference ``network.RoomNet``
  ``roomnet_amd.infer``               <- reference ``infer`` (``classify_im_dir``, ``CLASS_LABELS`` ...)
Execution goes through libroomnet_hip.so (C ABI in include/roomnet_hip.h); there is no CPU
fallback in the product path.
"""
from .graph import build_graph  # noqa: F401

__all__ = ["build_graph", "RoomNet", "classify_im_dir", "CLASS_LABELS"]


def __getattr__(name):
    # lazy: importing the package must not require the HIP library (oracle/tools use tf_bundle only)
    if name == "RoomNet":
        from .network import RoomNet
        return RoomNet
    if name in ("classify_im_dir", "CLASS_LABELS"):
        from . import infer
        return getattr(infer, name)
    raise AttributeError(name)
