"""1-D spectral convolution -- the core of a Fourier-Neural-Operator layer.

The reference has no FNO (BASELINE configs[4] names one, the repository contains none); the operator is the
published one (Li et al. 2021):  ``y = irfft(W . rfft(x)[..., :modes], n=N)`` with complex weights ``W [Cin, Cout, modes]``.

* CPU tensors (and CUDA tensors after ``ops.enable_fused(False)``): ``torch.fft`` -- this is also the fp32 reference the
  kernel is tested against.
* CUDA tensors: ``libspectral_hip.so`` (include/spectral_hip.h): truncated-DFT GEMM -> complex mode mixing -> inverse-DFT
  GEMM fused in one launch per direction, one workgroup per sample, MFMA fp32.  No fallback: a missing library raises.
  **Parity unpinned** against the reference (nothing there to compare with); pinned against ``torch.fft`` in
  tests/test_fno.py.
"""
import ctypes
import os

import torch
from torch import nn

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.abspath(os.path.join(_HERE, "..", "..", "lib", "libspectral_hip.so"))
_p, _i = ctypes.c_void_p, ctypes.c_int
SYMBOLS = (
    ("spec_conv_forward", [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p]),
    ("spec_conv_backward", [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p]),
)
_lib = None


class SpectralHipError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SpectralHipError(f"{LIB_PATH} not found: build it (python -c 'import __graft_entry__ as g; g.build()'). "
                                   f"The fused spectral convolution has no fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, args in SYMBOLS:
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = ctypes.c_int, args
        lib.spec_last_error.restype = ctypes.c_char_p
        _lib = lib
    return _lib


def _check(rc):
    if rc != 0:
        raise SpectralHipError(f"libspectral_hip error {rc}: {load().spec_last_error().decode(errors='replace')}")


def _stream():
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def spectral_conv1d_reference(x, wr, wi):
    """torch.fft spelling: x [B, Cin, N], wr / wi [Cin, Cout, modes] -> [B, Cout, N]."""
    n, modes = x.shape[-1], wr.shape[-1]
    xf = torch.fft.rfft(x, dim=-1)[..., :modes]
    yf = torch.einsum("bim,iom->bom", xf, torch.complex(wr, wi))
    out = torch.zeros(x.shape[0], wr.shape[1], n // 2 + 1, dtype=yf.dtype, device=x.device)
    out[..., :modes] = yf
    return torch.fft.irfft(out, n=n, dim=-1)


class _SpectralConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, wr, wi):
        x, wr, wi = x.contiguous(), wr.contiguous(), wi.contiguous()
        b, cin, n = x.shape
        cout, modes = wr.shape[1], wr.shape[2]
        y = torch.empty((b, cout, n), device=x.device, dtype=torch.float32)
        need = any(ctx.needs_input_grad)
        xft = torch.empty((b, cin, 2, modes), device=x.device, dtype=torch.float32) if need else None
        _check(load().spec_conv_forward(_stream(), _ptr(x), _ptr(wr), _ptr(wi), b, cin, cout, n, modes, _ptr(y), _ptr(xft)))
        ctx.save_for_backward(wr, wi, xft)
        ctx.dims = (b, cin, cout, n, modes)
        return y

    @staticmethod
    def backward(ctx, dy):
        wr, wi, xft = ctx.saved_tensors
        b, cin, cout, n, modes = ctx.dims
        dy = dy.contiguous()
        dx = torch.empty((b, cin, n), device=dy.device, dtype=torch.float32)
        gy = torch.empty((b, cout, 2, modes), device=dy.device, dtype=torch.float32)
        _check(load().spec_conv_backward(_stream(), _ptr(dy), _ptr(wr), _ptr(wi), b, cin, cout, n, modes, _ptr(dx), _ptr(gy)))
        gwr = gwi = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            # contraction over the batch of two tiny tensors ([B, C, modes] each): one batched GEMM per term
            xr, xi, gr, gi = xft[:, :, 0], xft[:, :, 1], gy[:, :, 0], gy[:, :, 1]
            gwr = torch.einsum("bom,bim->iom", gr, xr) + torch.einsum("bom,bim->iom", gi, xi)
            gwi = torch.einsum("bom,bim->iom", gi, xr) - torch.einsum("bom,bim->iom", gr, xi)
        return (dx if ctx.needs_input_grad[0] else None), gwr, gwi


def spectral_conv1d(x, wr, wi):
    from pdecontrol.surrogates import ops
    if ops.use_fused(x):
        if x.dtype != torch.float32:
            raise SpectralHipError("the fused spectral convolution is fp32")
        return _SpectralConvFn.apply(x, wr, wi)
    return spectral_conv1d_reference(x, wr, wi)


class SpectralConv1d(nn.Module):
    """Fourier layer core: complex weights on the lowest ``modes`` frequencies (stored as two real tensors)."""

    def __init__(self, in_channels: int, out_channels: int, modes: int):
        super().__init__()
        self.in_channels, self.out_channels, self.modes = in_channels, out_channels, modes
        scale = 1.0 / (in_channels * out_channels)
        self.weight_real = nn.Parameter(scale * torch.rand(in_channels, out_channels, modes))
        self.weight_imag = nn.Parameter(scale * torch.rand(in_channels, out_channels, modes))

    def forward(self, x):
        return spectral_conv1d(x, self.weight_real, self.weight_imag)
