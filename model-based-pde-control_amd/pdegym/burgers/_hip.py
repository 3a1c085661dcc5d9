"""ctypes binding of libburgers_hip.so (C ABI: include/burgers_hip.h).  No fallback: a missing library raises."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.abspath(os.path.join(_HERE, "..", "..", "lib", "libburgers_hip.so"))

_p, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
SYMBOLS = (
    ("bg_step", [_p, _p, _p, _p, _i, _i, _i, _f, _f, _f, ctypes.c_long, _p, _p, _p]),
    ("bg_residual", [_p, _p, _p, _i, _i, _f, _f, _p]),
)
_lib = None


class BurgersHipError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BurgersHipError(f"{LIB_PATH} not found: build it (python -c 'import __graft_entry__ as g; g.build()'). "
                                  f"The Burgers stepper has no CPU fallback.")
        import torch  # noqa: F401  libburgers_hip.so must bind to torch's bundled HIP runtime (one runtime per process)
        lib = ctypes.CDLL(LIB_PATH)
        for name, args in SYMBOLS:
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = ctypes.c_int, args
        lib.bg_last_error.restype = ctypes.c_char_p
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        raise BurgersHipError(f"libburgers_hip error {rc}: {load().bg_last_error().decode(errors='replace')}")
