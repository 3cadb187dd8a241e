"""MI355X-native path-tracing core for PathTracer: host-side Python access to the C ABI."""
from . import abi
from .device import Device, PtxError, load_library, make_params
from .scene import Scene

__all__ = ["abi", "Device", "PtxError", "load_library", "make_params", "Scene"]
