"""f4: FNO-style surrogate and its fused spectral-convolution kernel.  The reference contains no FNO:
**parity unpinned** against it.  The operator is pinned to its definition (dense DFT algebra, fp64) and the HIP kernel
to the torch.fft spelling in fp32."""
import ctypes
import math
import os
import re

import numpy as np
import pytest
import torch

from pdecontrol.architectures import BurgersFNO
from pdecontrol.surrogates import spectral
from pdecontrol.surrogates.training import PDETrainingModule

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_spelling_is_the_truncated_dft_definition():
    torch.manual_seed(0)
    b, ci, co, n, m = 3, 4, 5, 32, 6
    x = torch.randn(b, ci, n, dtype=torch.float64)
    wr, wi = torch.randn(ci, co, m, dtype=torch.float64), torch.randn(ci, co, m, dtype=torch.float64)
    pos, mode = torch.arange(n, dtype=torch.float64), torch.arange(m, dtype=torch.float64)
    th = 2 * math.pi * mode[:, None] * pos[None, :] / n
    xr, xi = x @ torch.cos(th).T, -(x @ torch.sin(th).T)
    yr = torch.einsum("bim,iom->bom", xr, wr) - torch.einsum("bim,iom->bom", xi, wi)
    yi = torch.einsum("bim,iom->bom", xr, wi) + torch.einsum("bim,iom->bom", xi, wr)
    s = torch.full((m,), 2.0 / n, dtype=torch.float64)
    s[0] = 1.0 / n
    dense = (s * yr) @ torch.cos(th) - (s * yi) @ torch.sin(th)
    np.testing.assert_allclose(spectral.spectral_conv1d_reference(x, wr, wi).numpy(), dense.numpy(), atol=1e-12)


def _module(device="cpu", n=64, width=16, modes=8, layers=2):
    torch.manual_seed(0)
    f = BurgersFNO()
    s = f.surrogate(delta=0.05, dscaling=None, tau=5, **f.model(width=width, modes=modes, layers=layers))
    m = PDETrainingModule(surrogate=s, loss=torch.nn.MSELoss(reduction="none"), tstep=0.05, delta=0.05, tau=5, tbtt=10)
    return m.to(device)


def test_fno_surrogate_through_the_training_module_on_cpu():
    import pdecontrol.architectures as arch
    assert getattr(arch, "BurgersFNO") is BurgersFNO                      # --factory BurgersFNO
    m = _module()
    g = torch.Generator().manual_seed(1)
    st, ac = torch.rand(4, 12, 1, 64, generator=g), torch.rand(4, 12, 1, 64, generator=g)
    opt = m.configure_optimizers()[0][0]
    losses = []
    for _ in range(5):
        out = m.training_step((st, ac), 0)
        opt.zero_grad()
        out["loss"].backward()
        opt.step()
        losses.append(float(out["loss"].detach()))
    assert out["outputs"].shape == (4, 12, 1, 64) and out["outdeltas"].shape == (4, 11, 1, 64)
    assert losses[-1] < losses[0]
    r = m.surrogate.rollout(states=st[:, :5], actions=ac[:, :10], times=0.05 * torch.arange(10), targets=0.05 * (torch.arange(10) + 1))
    assert r.outputs.shape == (4, 10, 1, 64) and r.hidden == ()


def test_spectral_c_abi_exports_and_rejects_bad_geometry():
    text = open(os.path.join(ROOT, "include", "spectral_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(spec_[a-z_]+)\s*\(", text)))
    lib = ctypes.CDLL(spectral.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted([n for n, _ in spectral.SYMBOLS] + ["spec_last_error"]) == declared
    lib.spec_last_error.restype = ctypes.c_char_p
    one = ctypes.c_void_p(1)
    assert lib.spec_conv_forward(None, one, one, one, 2, 16, 16, 100, 8, one, None) < 0     # N not a power of two
    assert b"power of two" in lib.spec_last_error()
    assert lib.spec_conv_forward(None, one, one, one, 2, 12, 16, 128, 8, one, None) < 0     # channels not multiples of 16
    assert lib.spec_conv_backward(None, one, one, one, 2, 16, 16, 128, 64, one, None) < 0    # modes >= N/2


@pytest.mark.gpu
@pytest.mark.parametrize("b,ci,co,n,m", [(5, 16, 16, 64, 8), (3, 32, 32, 512, 16), (2, 16, 32, 1024, 24), (7, 32, 16, 128, 32)])
def test_fused_spectral_conv_forward_backward(b, ci, co, n, m):
    from pdecontrol.surrogates import ops
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(b + n)
    x = torch.randn(b, ci, n, generator=g)
    wr, wi = torch.randn(ci, co, m, generator=g) / ci, torch.randn(ci, co, m, generator=g) / ci
    dy = torch.randn(b, co, n, generator=g)
    ref_in = [t.clone().double().requires_grad_(True) for t in (x, wr, wi)]
    y_ref = spectral.spectral_conv1d_reference(*ref_in)
    y_ref.backward(dy.double())
    got_in = [t.clone().to(dev).requires_grad_(True) for t in (x, wr, wi)]
    assert ops.use_fused(got_in[0])
    y = spectral.spectral_conv1d(*got_in)
    y.backward(dy.to(dev))
    torch.cuda.synchronize(dev)
    tol = lambda ref: 3e-5 * float(ref.abs().max())                 # fp32 sums of N (forward) / B*N (weights) terms
    np.testing.assert_allclose(y.detach().cpu().numpy(), y_ref.detach().numpy(), rtol=0, atol=tol(y_ref.detach()))
    for name, t, r in zip(("dx", "dWr", "dWi"), got_in, ref_in):
        np.testing.assert_allclose(t.grad.cpu().numpy(), r.grad.numpy(), rtol=0, atol=tol(r.grad), err_msg=name)


@pytest.mark.gpu
def test_fno_training_step_gpu_matches_cpu_and_trains():
    dev = torch.device("cuda", 0)
    cpu, gpu = _module("cpu", n=512, width=32, modes=16, layers=4), _module(dev, n=512, width=32, modes=16, layers=4)
    g = torch.Generator().manual_seed(2)
    st, ac = torch.rand(8, 12, 1, 512, generator=g) * 2 - 1, torch.rand(8, 12, 1, 512, generator=g) * 2 - 1
    ref = cpu.training_step((st, ac), 0)
    ref["loss"].backward()
    out = gpu.training_step((st.to(dev), ac.to(dev)), 0)
    out["loss"].backward()
    torch.cuda.synchronize(dev)
    rel = abs(float(out["loss"].detach()) - float(ref["loss"].detach())) / abs(float(ref["loss"].detach()))
    assert rel < 1e-5, rel
    for (k, p), q in zip(gpu.surrogate.named_parameters(), cpu.surrogate.parameters()):
        np.testing.assert_allclose(p.grad.cpu().numpy(), q.grad.numpy(), rtol=0, atol=5e-5 * max(1e-3, float(q.grad.abs().max())),
                                   err_msg=k)
    opt = gpu.configure_optimizers()[0][0]
    assert isinstance(opt, torch.optim.Adam)          # not the KS-family pack optimizer
    losses = []
    for _ in range(5):
        o = gpu.training_step((st.to(dev), ac.to(dev)), 0)
        opt.zero_grad()
        o["loss"].backward()
        opt.step()
        losses.append(float(o["loss"].detach()))
    assert losses[-1] < losses[0]


@pytest.mark.gpu
def test_fno_step_as_one_captured_graph():
    """PDETrainingModule.fused_step is architecture-agnostic: the FNO step (spectral HIP kernel + rocBLAS / torch kernels)
    captured as one hipGraph with torch's capturable Adam == the same steps run eagerly with torch.optim.Adam."""
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(4)
    st, ac = (torch.rand(4, 12, 1, 128, generator=g) * 2 - 1).to(dev), (torch.rand(4, 12, 1, 128, generator=g) * 2 - 1).to(dev)
    ref, m = _module(dev, n=128), _module(dev, n=128)
    opt = torch.optim.Adam(ref.surrogate.parameters(), lr=ref.lr)
    l_ref = []
    for _ in range(4):
        out = ref.training_step((st, ac), 0)
        opt.zero_grad(set_to_none=True)
        out["loss"].backward()
        opt.step()
        l_ref.append(float(out["loss"].detach()))
    l_mod = [float(m.fused_step((st, ac))["loss"]) for _ in range(4)]
    np.testing.assert_allclose(l_mod, l_ref, rtol=2e-5)
    assert l_mod[-1] < l_mod[0] and len(m._graphed_steps) == 1


# ---------------------------------------------------------------------------------------------------------------------
# whole-network kernels (csrc/fno.hip): one launch per model evaluation.  Parity unpinned against the reference (it has
# no FNO); pinned against the per-operator torch spelling of the same module.
# ---------------------------------------------------------------------------------------------------------------------
def test_fno_c_abi_exports():
    from pdecontrol.surrogates import fno_hip
    text = open(os.path.join(ROOT, "include", "spectral_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(fno_[a-z_]+)\s*\(", text)))
    lib = ctypes.CDLL(spectral.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(n for n, _, _ in fno_hip.SYMBOLS) == declared
    lib.fno_row_width.restype = ctypes.c_int
    assert lib.fno_row_width() >= 96 + 4 * 1056 + 1089


def _fno_pair(dev, n, scaled):
    from pdegym.common.transforms import BatchTransform, Normalize
    und = None
    if scaled:
        norm = Normalize(aggregate=True, batched=True)
        norm.mean, norm.var, norm.count = torch.full((1, 1, 1), 0.02), torch.full((1, 1, 1), 0.3), 50
        und = BatchTransform(norm)
    mods = []
    for device in ("cpu", dev):
        torch.manual_seed(0)
        f = BurgersFNO()
        s = f.surrogate(delta=0.05, dscaling=None if und is None else und.Inverse, tau=5, **f.model())
        mods.append(PDETrainingModule(surrogate=s, loss=torch.nn.MSELoss(reduction="none"), tstep=0.05, delta=0.05,
                                      undscaling=und, tau=5, tbtt=10).to(device))
    return mods


@pytest.mark.gpu
@pytest.mark.parametrize("n,scaled,B,T", [(512, False, 4, 20), (128, True, 3, 13), (64, False, 2, 7)])
def test_whole_network_kernels_training_step_vs_cpu(n, scaled, B, T):
    """training_step + backward through the whole-network kernels (teacher-forced launch + free-running chain, two TBPTT
    chunks at T = 20) against the same module on the CPU: loss <= 1e-5 relative (north_star's TBPTT tolerance), every
    parameter gradient to fp32 summation noise of its tensor's scale."""
    from conftest import check_grads
    from pdecontrol.surrogates import fno_hip
    dev = torch.device("cuda", 0)
    cpu, gpu = _fno_pair(dev, n, scaled)
    assert fno_hip.supported(gpu.surrogate.model, n)
    g = torch.Generator().manual_seed(n + T)
    st, ac = torch.rand(B, T, 1, n, generator=g) * 2 - 1, torch.rand(B, T, 1, n, generator=g) * 2 - 1
    ref = cpu.training_step((st, ac), 0)
    ref["loss"].backward()
    calls = []
    orig = fno_hip._FNOTBPTTFn.apply
    fno_hip._FNOTBPTTFn.apply = lambda *a: (calls.append(1), orig(*a))[1]
    try:
        out = gpu.training_step((st.to(dev), ac.to(dev)), 0)
    finally:
        fno_hip._FNOTBPTTFn.apply = orig
    assert len(calls) == 1, "the TBPTT pass must run on the whole-network kernels, as one autograd node"
    out["loss"].backward()
    torch.cuda.synchronize(dev)
    rel = abs(float(out["loss"].detach()) - float(ref["loss"].detach())) / abs(float(ref["loss"].detach()))
    assert rel < 1e-5, rel
    np.testing.assert_allclose(out["outputs"].detach().cpu().numpy(), ref["outputs"].detach().numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(out["outdeltas"].detach().cpu().numpy(), ref["outdeltas"].detach().numpy(), rtol=1e-4, atol=2e-5)
    check_grads(f"FNO whole-network kernels vs CPU (N={n}, scaled={scaled})",
                {k: p.grad.detach().cpu().numpy() for k, p in gpu.surrogate.named_parameters()},
                dict((k, p.grad.numpy()) for k, p in cpu.surrogate.named_parameters()).__getitem__, tol=2e-4)


@pytest.mark.gpu
def test_whole_network_rollout_inference_and_fallback_geometry():
    from pdecontrol.surrogates import fno_hip
    dev = torch.device("cuda", 0)
    cpu, gpu = _fno_pair(dev, 256, True)
    g = torch.Generator().manual_seed(9)
    st, ac = torch.rand(5, 3, 1, 256, generator=g) * 2 - 1, torch.rand(5, 7, 1, 256, generator=g) * 2 - 1
    times, targets = 0.05 * torch.arange(7), 0.05 * (torch.arange(7) + 1)
    with torch.no_grad():
        r_cpu = cpu.surrogate.rollout(states=st, actions=ac, times=times, targets=targets)
        r_gpu = gpu.surrogate.rollout(states=st.to(dev), actions=ac.to(dev), times=times, targets=targets)
    np.testing.assert_allclose(r_gpu.outputs.cpu().numpy(), r_cpu.outputs.numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(r_gpu.deltas.cpu().numpy(), r_cpu.deltas.numpy(), rtol=1e-4, atol=2e-5)
    # a geometry the whole-network kernels are not built for keeps the per-operator path (and still works)
    small = _module(dev, n=128)            # width 16, 8 modes, 2 layers
    assert not fno_hip.supported(small.surrogate.model, 128)
    out = small.training_step(((torch.rand(2, 12, 1, 128, generator=g) * 2 - 1).to(dev), (torch.rand(2, 12, 1, 128, generator=g) * 2 - 1).to(dev)), 0)
    assert torch.isfinite(out["loss"])
