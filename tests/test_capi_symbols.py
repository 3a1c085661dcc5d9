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


def test_surrogate_abi_rejects_bad_arguments_before_touching_the_device():
    """Argument validation of libsurrogate_hip.so happens on the host: negative status + a message, no HIP call."""
    lib_path = os.path.join(LIBDIR, "libsurrogate_hip.so")
    if not os.path.exists(lib_path):
        pytest.skip("libsurrogate_hip.so not built")
    from pdecontrol.surrogates import hipops
    lib = hipops.load()
    enc, chunk = hipops.EncoderParams(), hipops.ChunkParams()
    null = None
    assert lib.sur_encoder_forward(null, ctypes.byref(enc), null, 4, null, null) < 0
    assert b"sur_encoder_forward" in lib.sur_last_error()
    assert lib.sur_encoder_backward(null, ctypes.byref(enc), null, null, 4, null, 0, 1, null) < 0
    assert lib.sur_chunk_forward(null, ctypes.byref(chunk), null, null, null, null, null, 0, 0, 1, 1, null, null, null, null, null) < 0
    assert b"sur_chunk_forward" in lib.sur_last_error()
    assert lib.sur_chunk_backward(null, ctypes.byref(chunk), null, null, null, null, 0, null, null, null, null, null, null, 1, 1,
                                  1, null, null, null, null, 0, 1, null, null) < 0
    assert lib.sur_chunk_workspace_floats(ctypes.byref(chunk), 0, 4) == 0
    assert lib.sur_encoder_forward_multi(null, 3, null, null, null, null, null, 512) < 0
    assert lib.sur_encoder_backward_multi(null, 4, null, null, null, null, null, null, null, null) < 0
    assert b"sur_encoder_backward_multi" in lib.sur_last_error()
    assert lib.sur_chunks_backward(null, ctypes.byref(chunk), 9, null, null, null, null, null, 4, 4, null, 0, 16, null, null) < 0
    assert lib.sur_flush_encoder_grads(null, ctypes.byref(enc), null, 0) < 0      # no partial buffer
    assert lib.sur_flush_chunk_grads(null, ctypes.byref(chunk), null, 0) < 0
    assert lib.sur_tbptt_delta_loss(null, null, 128, 64, null, 1, 2, 64, 0.25, 0.0, 1.0, null, null, null, null, null, null, null) < 0
    assert b"sur_tbptt_delta_loss" in lib.sur_last_error()
    # geometry queries are pure host functions
    chunk.ca, chunk.cs, chunk.hq, chunk.c_mid = 4, 16, 16, 8
    sizes = [4 * 16 * 3, 16, 16 * 16 * 3] * 4 + [16 * 16 * 3, 16, 32, 32, 16 * 8 * 3, 8, 64, 64, 8 * 7, 1, 64, 64, 5, 1]
    for i, n in enumerate(sizes):
        chunk.size[i] = n
    assert lib.sur_chunk_saved_floats(ctypes.byref(chunk)) == 3840        # 3 712 floats padded to 1 KiB DMA pieces
    assert lib.sur_chunk_workspace_floats(ctypes.byref(chunk), 10, 64) == 10 * 64 * (5 * 256 + 64)
    chunk.hq = 64                                                        # N = 256: one padded copy still fits
    assert lib.sur_chunk_saved_floats(ctypes.byref(chunk)) == 14848
    chunk.hq = 8                                                         # N = 32: latent rows narrower than a GEMM tile
    assert lib.sur_chunk_saved_floats(ctypes.byref(chunk)) == 0
    enc.n, enc.c[0], enc.c[1], enc.c[2], enc.c[3] = 64, 1, 8, 16, 16
    enc.stride[0], enc.stride[1], enc.stride[2] = 2, 2, 1
    assert lib.sur_encoder_saved_floats(ctypes.byref(enc)) == 7 * (8 * 32 + 16 * 16 + 16 * 16)
    assert lib.sur_encoder_workspace_floats(ctypes.byref(enc), 10) == 10 * (8 * 32 + 16 * 16)
    # grid widths: every LayerNorm row must be 16, 32 or a multiple of 64 up to 256 values wide
    for n, ok in ((64, True), (128, True), (256, True), (32, False), (96, False), (192, False), (512, False)):
        enc.n, chunk.hq = n, n // 4
        rc = lib.sur_geometry_supported(ctypes.byref(enc), ctypes.byref(enc), ctypes.byref(chunk))
        assert (rc == 0) == ok, (n, rc, lib.sur_last_error())
        if not ok:
            assert rc == -4 and b"LayerNorm" in lib.sur_last_error()
            assert lib.sur_encoder_forward(null, ctypes.byref(enc), ctypes.c_void_p(16), 4, ctypes.c_void_p(16), null) == -4
