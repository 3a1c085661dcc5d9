"""Surrogate TBPTT step on the MI355X against the golden tensors from the reference.
Tolerance (BASELINE.json north_star): TBPTT loss within 1e-5 relative of the CPU reference (fp32)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _module(device, scaled=False, seed=0):
    from pdecontrol.architectures import KSAutoRegConvolutionalLSTM
    from pdecontrol.surrogates.training import PDETrainingModule
    from pdegym.common.transforms import BatchTransform, Normalize
    torch.manual_seed(seed)
    und = None
    if scaled:
        norm = Normalize(aggregate=True, batched=True)
        norm.mean, norm.var, norm.count = torch.full((1, 1, 1), 0.01), torch.full((1, 1, 1), 0.5), 100
        und = BatchTransform(norm)
    f = KSAutoRegConvolutionalLSTM()
    s = f.surrogate(delta=0.25, dscaling=None if und is None else und.Inverse, tau=5, **f.model())
    m = PDETrainingModule(surrogate=s, loss=torch.nn.MSELoss(reduction="none"), tstep=0.25, delta=0.25,
                          undscaling=und, tau=5, tbtt=10)
    return m.to(device)


@pytest.mark.parametrize("scaled", [False, True])
def test_training_step_parity_on_gpu(sur_golden, scaled):
    g, tag = sur_golden, ("b8n" if scaled else "b8")
    dev = torch.device("cuda", 0)
    m = _module(dev, scaled)
    s, a = torch.from_numpy(g["b8_states"]).to(dev), torch.from_numpy(g["b8_actions"]).to(dev)
    res = m.training_step((s, a), 0)
    res["loss"].backward()
    rel = abs(res["loss"].item() - g[f"{tag}_loss"]) / abs(g[f"{tag}_loss"])
    assert rel < 1e-5, rel
    np.testing.assert_allclose(res["hsteploss"].cpu().numpy(), g[f"{tag}_hsteploss"], rtol=1e-4)
    np.testing.assert_allclose(res["outputs"].cpu().numpy(), g[f"{tag}_outputs"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(res["outdeltas"].cpu().numpy(), g[f"{tag}_outdeltas"], rtol=1e-3, atol=1e-4)
    for k, p in m.surrogate.named_parameters():
        if p.grad is not None:
            ref = g[f"{tag}_grad/" + k]
            np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=1e-2, atol=2e-5 * max(1.0, np.abs(ref).max()),
                                       err_msg=k)


def test_known_answer_b64_on_gpu(sur_golden):
    dev = torch.device("cuda", 0)
    m = _module(dev)
    gen = torch.Generator().manual_seed(1)
    s = (torch.rand(64, 20, 1, 64, generator=gen) * 2 - 1).to(dev)
    a = (torch.rand(64, 20, 1, 64, generator=gen) * 2 - 1).to(dev)
    loss = m.training_step((s, a), 0)["loss"].item()
    assert abs(loss - 10.806351661682129) / 10.806351661682129 < 1e-5


def test_hip_graph_step_equals_eager_training():
    from pdecontrol.surrogates.bench_tbptt import build_module, synthetic_batch
    from pdecontrol.surrogates.graph_step import GraphedTBPTTStep
    dev = torch.device("cuda", 0)
    batch = synthetic_batch(B=16, device=dev)
    eager = build_module(dev)
    opt = torch.optim.Adam(eager.surrogate.parameters(), lr=1e-3)
    losses_e = []
    for _ in range(4):
        opt.zero_grad(set_to_none=True)
        out = eager.training_step(batch, 0)
        out["loss"].backward()
        opt.step()
        losses_e.append(out["loss"].item())
    graphed = GraphedTBPTTStep(build_module(dev), tuple(batch[0].shape))
    losses_g = []
    for i in range(4):
        res = graphed.step(*batch) if i == 0 else graphed.step()
        losses_g.append(res["loss"].item())
    np.testing.assert_allclose(losses_g, losses_e, rtol=2e-5)
    assert losses_g[-1] < losses_g[0]  # it trains
    pe = torch.cat([p.detach().reshape(-1) for p in eager.surrogate.parameters()])
    pg = torch.cat([p.detach().reshape(-1) for p in graphed.module.surrogate.parameters()])
    assert (pe - pg).abs().max().item() < 5e-4


def test_module_fused_step_replays_one_graph_per_shape():
    """PDETrainingModule.fused_step == GraphedTBPTTStep on the same module state; one graph per batch shape."""
    from pdecontrol.surrogates.bench_tbptt import build_module, synthetic_batch
    from pdecontrol.surrogates.graph_step import GraphedTBPTTStep
    dev = torch.device("cuda", 0)
    batch = synthetic_batch(B=8, device=dev)
    ref = GraphedTBPTTStep(build_module(dev), tuple(batch[0].shape))
    l_ref = [float(ref.step(*batch)["loss"].detach()) for _ in range(3)]
    m = build_module(dev)
    l_mod = [float(m.fused_step(batch)["loss"].detach()) for _ in range(3)]
    # plain torch / MIOpen kernels under the graph: backward reductions are not bit-reproducible between two captures
    np.testing.assert_allclose(l_mod, l_ref, rtol=2e-5)
    assert l_mod[-1] < l_mod[0]
    assert len(m._graphed_steps) == 1
    m.fused_step(synthetic_batch(B=4, device=dev))
    assert len(m._graphed_steps) == 2
