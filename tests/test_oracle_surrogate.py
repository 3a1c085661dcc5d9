"""oracle/surrogate_oracle.py (functional restatement) against golden tensors from the reference."""
import numpy as np
import torch

from oracle import surrogate_oracle as so


def _grads(w):
    return {k: v.grad for k, v in w.items() if v.grad is not None}


def test_tbptt_loss_outputs_and_gradients(sur_golden):
    g = sur_golden
    w = so.load_weights(g)
    s, a = torch.from_numpy(g["b8_states"]), torch.from_numpy(g["b8_actions"])
    loss, hstep, outputs, outdeltas = so.tbptt_loss(w, s, a)
    loss.backward()
    # same torch kernels, different composition order in places -> fp32 rounding level
    np.testing.assert_allclose(loss.item(), g["b8_loss"], rtol=1e-6)
    np.testing.assert_allclose(hstep.detach().numpy(), g["b8_hsteploss"], rtol=1e-5)
    np.testing.assert_allclose(outputs.detach().numpy(), g["b8_outputs"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(outdeltas.detach().numpy(), g["b8_outdeltas"], rtol=1e-4, atol=1e-5)
    grads = _grads(w)
    assert len(grads) == len([k for k in g.files if k.startswith("b8_grad/")])
    for k, v in grads.items():
        ref = g["b8_grad/" + k]
        np.testing.assert_allclose(v.numpy(), ref, rtol=1e-3, atol=1e-5 * max(1.0, np.abs(ref).max()), err_msg=k)


def test_tbptt_with_normalize_scaling(sur_golden):
    g = sur_golden
    w = so.load_weights(g)
    mean, var, eps = 0.01, 0.5, 1e-4
    dscale = lambda d: d * np.float32(np.sqrt(np.float32(var) + np.float32(eps))) + np.float32(mean)
    undscale = lambda t: (t - np.float32(mean)) / np.float32(np.sqrt(np.float32(var) + np.float32(eps)))
    s, a = torch.from_numpy(g["b8_states"]), torch.from_numpy(g["b8_actions"])
    loss, hstep, outputs, outdeltas = so.tbptt_loss(w, s, a, dscale=dscale, undscale=undscale)
    np.testing.assert_allclose(loss.item(), g["b8n_loss"], rtol=1e-6)
    np.testing.assert_allclose(outputs.detach().numpy(), g["b8n_outputs"], rtol=1e-4, atol=1e-5)


def test_known_answer_b64(sur_golden):
    # SURVEY.md 8c known answer: B=64 loss 10.806351661682129 (inputs regenerated from the seeds)
    g = sur_golden
    w = so.load_weights(g, requires_grad=False)
    gen = torch.Generator().manual_seed(1)
    s = torch.rand(64, 20, 1, 64, generator=gen) * 2 - 1
    a = torch.rand(64, 20, 1, 64, generator=gen) * 2 - 1
    np.testing.assert_array_equal(s[:8].numpy(), g["b8_states"])  # the generator is stable
    with torch.no_grad():
        loss, hstep, _, _ = so.tbptt_loss(w, s, a)
    np.testing.assert_allclose(loss.item(), 10.806351661682129, rtol=1e-6)
    np.testing.assert_allclose(loss.item(), g["b64_loss"], rtol=1e-6)
    np.testing.assert_allclose(hstep.numpy(), g["b64_hsteploss"], rtol=1e-5)
