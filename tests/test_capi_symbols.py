"""The C-ABI libraries load without a GPU and export every function their headers declare
(no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "model-based-pde-control_amd", "lib")


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:ks|sur)_[a-z0-9_]+)\s*\(", text)))


@pytest.mark.parametrize("header,lib", [("kspde.h", "libkspde.so"), ("surrogate_hip.h", "libsurrogate_hip.so")])
def test_library_exports_every_declared_symbol(header, lib):
    path = os.path.join(LIBDIR, lib)
    if not os.path.exists(path):
        pytest.skip(f"{lib} not built (run __graft_entry__.build())")
    import torch  # noqa: F401  (its bundled HIP runtime must be the one the library binds to)
    handle = ctypes.CDLL(path)
    names = declared_functions(header)
    assert len(names) >= 6
    missing = [n for n in names if not hasattr(handle, n)]
    assert not missing, missing


def test_python_bindings_cover_the_headers():
    import kspde
    from pdecontrol.surrogates import hipops
    assert sorted(n for n, _, _ in kspde.SYMBOLS) == declared_functions("kspde.h")
    bound = sorted([n for n, _ in hipops.SYMBOLS] + ["sur_last_error"])
    assert bound == declared_functions("surrogate_hip.h")


def test_kspde_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import kspde
    with pytest.raises(kspde.KSError) as e:
        kspde.KSStepper(4, 64)
    assert "no CPU path" in str(e.value) or "HIP" in str(e.value)


def test_fused_surrogate_requires_library(monkeypatch):
    from pdecontrol.surrogates import hipops
    monkeypatch.setattr(hipops, "LIB_PATH", "/nonexistent/libsurrogate_hip.so")
    monkeypatch.setattr(hipops, "_lib", None)
    with pytest.raises(hipops.SurrogateHipError):
        hipops.load()
