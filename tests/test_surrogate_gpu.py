"""Surrogate TBPTT step on the MI355X against the golden tensors from the reference.
Tolerance (BASELINE.json north_star): TBPTT loss within 1e-5 relative of the CPU reference (fp32)."""
import numpy as np
import pytest
import torch

from conftest import check_grads

pytestmark = pytest.mark.gpu


def _named_grads(m):
    return {k: p.grad.detach().cpu().numpy() for k, p in m.surrogate.named_parameters() if p.requires_grad and p.grad is not None}


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
@pytest.mark.parametrize("fused", [False, True])
def test_training_step_parity_on_gpu(sur_golden, scaled, fused):
    """Both CUDA paths (fused HIP kernels = the default; plain PyTorch-ROCm = explicit opt-out) against the reference's
    own loss / outputs / gradients."""
    from pdecontrol.surrogates import ops
    g, tag = sur_golden, ("b8n" if scaled else "b8")
    dev = torch.device("cuda", 0)
    m = _module(dev, scaled)
    s, a = torch.from_numpy(g["b8_states"]).to(dev), torch.from_numpy(g["b8_actions"]).to(dev)
    with ops.fused(fused):
        res = m.training_step((s, a), 0)
        res["loss"].backward()
    torch.cuda.synchronize(dev)
    rel = abs(res["loss"].item() - g[f"{tag}_loss"]) / abs(g[f"{tag}_loss"])
    assert rel < 1e-5, rel
    np.testing.assert_allclose(res["hsteploss"].cpu().numpy(), g[f"{tag}_hsteploss"], rtol=1e-4)
    np.testing.assert_allclose(res["outputs"].cpu().numpy(), g[f"{tag}_outputs"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(res["outdeltas"].cpu().numpy(), g[f"{tag}_outdeltas"], rtol=1e-3, atol=1e-4)
    check_grads(f"n64 {tag} training_step fused={fused}", _named_grads(m), lambda k: g[f"{tag}_grad/" + k])


@pytest.mark.parametrize("fused", [False, True])
def test_known_answer_b64_on_gpu(sur_golden, fused):
    """The benchmarked batch (B = 64, T = 20) against the reference's known answer and its gradients
    (tests/golden: b64_*), on the default fused path and on the plain path."""
    from pdecontrol.surrogates import ops
    g = sur_golden
    dev = torch.device("cuda", 0)
    m = _module(dev)
    gen = torch.Generator().manual_seed(1)
    s = (torch.rand(64, 20, 1, 64, generator=gen) * 2 - 1).to(dev)
    a = (torch.rand(64, 20, 1, 64, generator=gen) * 2 - 1).to(dev)
    with ops.fused(fused):
        assert ops.use_fused(s) == fused
        res = m.training_step((s, a), 0)
        res["loss"].backward()
    torch.cuda.synchronize(dev)
    loss = res["loss"].item()
    assert abs(loss - 10.806351661682129) / 10.806351661682129 < 1e-5
    if "b64_loss" in g.files:
        assert abs(loss - float(g["b64_loss"])) / float(g["b64_loss"]) < 1e-5
        np.testing.assert_allclose(res["hsteploss"].cpu().numpy(), g["b64_hsteploss"], rtol=1e-4)
        gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.surrogate.parameters() if p.grad is not None)).item()
        assert abs(gn - float(g["b64_grad_norm"])) / float(g["b64_grad_norm"]) < 1e-3
        check_grads(f"n64 b64 training_step fused={fused}", _named_grads(m), lambda k: g["b64_grad/" + k])


def test_pipelined_training_pass_against_reference_b64(sur_golden):
    """The captured step's hand-scheduled pass (hipops.fused_tbptt_train: no autograd, chunk c's loss rows + backward on a
    branch stream beside chunk c+1's forward) against the reference's known answer and gradients at the benchmarked batch."""
    g = sur_golden
    dev = torch.device("cuda", 0)
    m = _module(dev)
    gen = torch.Generator().manual_seed(1)
    s = (torch.rand(64, 20, 1, 64, generator=gen) * 2 - 1).to(dev)
    a = (torch.rand(64, 20, 1, 64, generator=gen) * 2 - 1).to(dev)
    res = m._pipelined_training_step((s, a))
    assert res is not None, "2 chunks of the controller's configuration must take the pipelined pass"
    torch.cuda.synchronize(dev)
    loss = res["loss"].item()
    assert abs(loss - 10.806351661682129) / 10.806351661682129 < 1e-5
    np.testing.assert_allclose(res["hsteploss"].cpu().numpy(), g["b64_hsteploss"], rtol=1e-4)
    check_grads("n64 b64 pipelined pass", _named_grads(m), lambda k: g["b64_grad/" + k])


@pytest.mark.parametrize("tbtt,T", [(10, 20), (7, 20), (6, 23)])
def test_pipelined_pass_equals_autograd_pass(tbtt, T):
    """2, 3 and 4 chunks (ragged last chunk): same loss / statistics / outputs / gradients as training_step + backward."""
    from pdecontrol.surrogates import ops
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(3)
    s = (torch.rand(16, T, 1, 64, generator=gen) * 2 - 1).to(dev)
    a = (torch.rand(16, T, 1, 64, generator=gen) * 2 - 1).to(dev)
    ref_m, pipe_m = _module(dev, scaled=True), _module(dev, scaled=True)
    ref_m.tbtt = pipe_m.tbtt = tbtt
    with ops.fused(True):
        ref = ref_m.training_step((s, a), 0)
        ref["loss"].backward()
        out = pipe_m._pipelined_training_step((s, a))
    torch.cuda.synchronize(dev)
    assert out is not None
    for key in ("loss", "hsteploss", "outputs", "outdeltas", "deltas"):
        np.testing.assert_allclose(out[key].cpu().numpy(), ref[key].detach().cpu().numpy(), rtol=1e-5, atol=1e-6, err_msg=key)
    for (k, p), q in zip(ref_m.surrogate.named_parameters(), pipe_m.surrogate.parameters()):
        if p.grad is not None:
            scale = max(1.0, p.grad.abs().max().item())
            np.testing.assert_allclose(q.grad.cpu().numpy(), p.grad.cpu().numpy(), rtol=1e-4, atol=2e-6 * scale, err_msg=k)


def test_split_graph_step_is_what_automatic_optimization_drives():
    """The default CUDA route of training_step under Lightning's automatic optimization -- forward + loss and backward +
    gradient reduction as two replayed graphs behind one autograd node (graph_step.GraphedAutogradStep) -- against the
    launch-by-launch fused step: closure order (training_step -> zero_grad(set_to_none) -> backward -> step), a scaled loss,
    gradient accumulation over two batches, a second batch shape, torch's own Adam on the handed-out gradients."""
    from pdecontrol.surrogates.bench_tbptt import build_module, synthetic_batch
    dev = torch.device("cuda", 0)
    b16, b8 = synthetic_batch(B=16, device=dev), synthetic_batch(B=8, device=dev)
    split, plain = build_module(dev), build_module(dev)
    plain.split_graphs = False
    assert split.split_graphs
    opts = [m.configure_optimizers()[0][0] for m in (split, plain)]
    losses = [[], []]
    for k in range(6):
        batch = b8 if k in (2, 3) else b16
        for i, (m, o) in enumerate(zip((split, plain), opts)):
            out = m.training_step(batch, 0)
            o.zero_grad(set_to_none=True)
            (out["loss"] * (0.5 if k == 4 else 1.0)).backward()
            o.step()
            losses[i].append(out["loss"].item())
    assert len(split.__dict__["_split_steps"]) == 2 and "_split_steps" not in plain.__dict__
    np.testing.assert_allclose(losses[0], losses[1], rtol=2e-5)
    for (name, p), q in zip(split.surrogate.named_parameters(), plain.surrogate.parameters()):
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=1e-4, atol=2e-6, err_msg=name)
    # accumulation: two backward passes onto the same gradients, no zero_grad in between; then torch's Adam on them
    for m in (split, plain):
        for p in m.surrogate.parameters():
            p.grad = None
        for batch in (b16, b16):
            m.training_step(batch, 0)["loss"].backward()
    for (name, p), q in zip(split.surrogate.named_parameters(), plain.surrogate.parameters()):
        if q.grad is None:           # H0 / C0: not trainable
            assert p.grad is None
            continue
        scale = max(1.0, q.grad.abs().max().item())
        np.testing.assert_allclose(p.grad.cpu().numpy(), q.grad.cpu().numpy(), rtol=1e-4, atol=2e-6 * scale, err_msg=name)
    # backward twice through the same loss (retain_graph): the second pass adds the same gradients again
    for m in (split, plain):
        for p in m.surrogate.parameters():
            p.grad = None
        loss = m.training_step(b16, 0)["loss"] * 0.25
        loss.backward(retain_graph=True)
        loss.backward()
    for (name, p), q in zip(split.surrogate.named_parameters(), plain.surrogate.parameters()):
        if q.grad is not None:
            scale = max(1.0, q.grad.abs().max().item())
            np.testing.assert_allclose(p.grad.cpu().numpy(), q.grad.cpu().numpy(), rtol=1e-4, atol=2e-6 * scale, err_msg=name)
    ref = {n: p.detach().clone() for n, p in split.surrogate.named_parameters()}
    adam = torch.optim.Adam(split.surrogate.parameters(), lr=1e-3)
    adam.step()
    assert any(not torch.equal(p, ref[n]) for n, p in split.surrogate.named_parameters())


def test_schedule_autotune_leaves_parameters_and_adam_state_untouched():
    """GraphedTBPTTStep(pipelined=None) captures the pipelined and the combined schedule, times both and keeps the faster
    one; the timing replays must not train: parameters, Adam moments and the step counter are put back."""
    from pdecontrol.surrogates.bench_tbptt import build_module, synthetic_batch
    from pdecontrol.surrogates.graph_step import GraphedTBPTTStep
    dev = torch.device("cuda", 0)
    batch = synthetic_batch(B=8, device=dev)
    tuned, forced = build_module(dev), build_module(dev)
    before = torch.cat([p.detach().reshape(-1) for p in tuned.surrogate.parameters()]).clone()
    g_auto = GraphedTBPTTStep(tuned, tuple(batch[0].shape))
    g_pipe = GraphedTBPTTStep(forced, tuple(batch[0].shape), pipelined=True)
    assert g_auto.autotune and set(g_auto.schedule_times_ms) == {"pipelined", "combined"} and not g_pipe.autotune
    after = torch.cat([p.detach().reshape(-1) for p in tuned.surrogate.parameters()])
    assert torch.equal(before, after)
    assert tuned.surrogate._fused_packs.adam_step_count() == 0
    la = [float(g_auto.step(*batch)["loss"]) for _ in range(3)]
    lp = [float(g_pipe.step(*batch)["loss"]) for _ in range(3)]
    np.testing.assert_allclose(la, lp, rtol=2e-5)
    assert tuned.surrogate._fused_packs.adam_step_count() == 3


def test_refitted_delta_statistics_invalidate_the_captured_graphs():
    """The controller re-fits the delta Normalize between training rounds (mbrl.py:597-602) on the object the surrogate's
    dscaling and the module's undscaling share; captured launches carry (mean, std) by value, so every graph cache must
    notice and re-capture: split-graph route, graphed route and launch-by-launch step agree before and after."""
    from pdecontrol.surrogates.bench_tbptt import build_module, synthetic_batch
    dev = torch.device("cuda", 0)
    batch = synthetic_batch(B=8, device=dev)
    split, graphed, plain = build_module(dev), build_module(dev), build_module(dev)
    plain.split_graphs = False

    def losses():
        out = []
        for m in (split, plain):
            for p in m.surrogate.parameters():
                p.grad = None
            res = m.training_step(batch, 0)
            res["loss"].backward()
            out.append((res["loss"].item(), torch.cat([p.grad.reshape(-1) for p in m.surrogate.parameters() if p.grad is not None])))
        out.append((graphed.fused_step(batch, lr=0.0)["loss"].item(), None))     # lr = 0: parameters stay equal
        return out

    first = losses()
    assert abs(first[0][0] - first[1][0]) < 2e-5 * abs(first[1][0]) and abs(first[2][0] - first[1][0]) < 2e-5 * abs(first[1][0])
    for m in (split, graphed, plain):
        norm = m.undscaling.transform
        norm.reset()
        norm.update(torch.linspace(-3.0, 5.0, 64).reshape(64, 1, 1))
    second = losses()
    assert abs(second[1][0] - first[1][0]) > 1e-2 * abs(first[1][0]), "the new statistics must change the loss"
    assert abs(second[0][0] - second[1][0]) < 2e-5 * abs(second[1][0]), (second[0][0], second[1][0])
    assert abs(second[2][0] - second[1][0]) < 2e-5 * abs(second[1][0]), (second[2][0], second[1][0])
    scale = second[1][1].abs().max().item()
    assert (second[0][1] - second[1][1]).abs().max().item() < 1e-4 * scale


@pytest.mark.parametrize("fused", [False, True])
def test_hip_graph_step_equals_eager_training(fused):
    """Graph replay == eager training in pytorch-lightning's closure order (training_step -> zero_grad(set_to_none)
    -> backward -> step): the fused flush must resolve param.grad when it runs, not during the forward pass."""
    from pdecontrol.surrogates import ops
    from pdecontrol.surrogates.bench_tbptt import build_module, synthetic_batch
    from pdecontrol.surrogates.graph_step import GraphedTBPTTStep
    dev = torch.device("cuda", 0)
    batch = synthetic_batch(B=16, device=dev)
    with ops.fused(fused):
        eager = build_module(dev)
        opt = eager.configure_optimizers()[0][0]
        losses_e = []
        for _ in range(4):
            out = eager.training_step(batch, 0)
            opt.zero_grad(set_to_none=True)
            out["loss"].backward()
            assert all(p.grad is not None for p in eager.surrogate.parameters() if p.requires_grad)
            opt.step()
            losses_e.append(out["loss"].item())
        graphed = GraphedTBPTTStep(build_module(dev), tuple(batch[0].shape))
        assert graphed.adam_in_flush == fused
        losses_g = []
        for i in range(4):
            res = graphed.step(*batch) if i == 0 else graphed.step()
            losses_g.append(res["loss"].item())
    np.testing.assert_allclose(losses_g, losses_e, rtol=2e-5)
    assert losses_g[-1] < losses_g[0]  # it trains
    pe = torch.cat([p.detach().reshape(-1) for p in eager.surrogate.parameters()])
    pg = torch.cat([p.detach().reshape(-1) for p in graphed.module.surrogate.parameters()])
    assert (pe - pg).abs().max().item() < 5e-4


@pytest.mark.parametrize("fused", [False, True])
def test_module_fused_step_replays_one_graph_per_shape(fused):
    """PDETrainingModule.fused_step == GraphedTBPTTStep on the same module state; one graph per batch shape."""
    from pdecontrol.surrogates import ops
    from pdecontrol.surrogates.bench_tbptt import build_module, synthetic_batch
    from pdecontrol.surrogates.graph_step import GraphedTBPTTStep
    dev = torch.device("cuda", 0)
    batch = synthetic_batch(B=8, device=dev)
    with ops.fused(fused):
        ref = GraphedTBPTTStep(build_module(dev), tuple(batch[0].shape))
        l_ref = [float(ref.step(*batch)["loss"]) for _ in range(3)]
        m = build_module(dev)
        l_mod = [float(m.fused_step(batch)["loss"]) for _ in range(3)]
        # plain torch / MIOpen kernels under the graph: backward reductions are not bit-reproducible between two captures
        np.testing.assert_allclose(l_mod, l_ref, rtol=2e-5)
        assert l_mod[-1] < l_mod[0]
        assert len(m._graphed_steps) == 1
        m.fused_step(synthetic_batch(B=4, device=dev))
        assert len(m._graphed_steps) == 2


@pytest.mark.parametrize("fused", [False, True])
def test_shape_switch_keeps_one_adam_state(fused):
    """B = 8 -> B = 4 -> B = 8 through fused_step (two captured graphs, ONE optimizer) against eager torch.optim.Adam
    over the same batch sequence: a ragged batch must not restart the moments or the bias correction."""
    from pdecontrol.surrogates import ops
    from pdecontrol.surrogates.bench_tbptt import build_module, synthetic_batch
    dev = torch.device("cuda", 0)
    big = synthetic_batch(B=8, device=dev)
    small = tuple(t[:4].contiguous() for t in big)
    seq = [big, small, big, small, big]
    with ops.fused(fused):
        ref = build_module(dev)
        opt = torch.optim.Adam(ref.surrogate.parameters(), lr=ref.lr)
        l_ref = []
        for batch in seq:
            out = ref.training_step(batch, 0)
            opt.zero_grad(set_to_none=True)
            out["loss"].backward()
            opt.step()
            l_ref.append(float(out["loss"].detach()))
        m = build_module(dev)
        l_mod = [float(m.fused_step(batch)["loss"]) for batch in seq]
        torch.cuda.synchronize(dev)
        assert len(m._graphed_steps) == 2
        np.testing.assert_allclose(l_mod, l_ref, rtol=3e-5)
        for (name, p), q in zip(m.surrogate.named_parameters(), ref.surrogate.parameters()):
            # Adam moves a parameter by up to lr per step whatever the gradient's magnitude: a near-zero gradient whose
            # rounding differs between two runs (the plain path's MIOpen reductions are not bit-reproducible) shows up as
            # a few 1e-6 after five steps; a restarted optimizer state would show up as ~1e-3
            np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=1e-4, atol=2e-5, err_msg=name)
        if fused:
            assert m.surrogate._fused_packs.adam_step_count() == len(seq)


def test_graphed_lr_is_a_device_scalar_the_scheduler_can_move():
    """lr = 0 through the device scalar freezes the parameters of an already captured step; restoring it trains again."""
    from pdecontrol.surrogates.bench_tbptt import build_module, synthetic_batch
    dev = torch.device("cuda", 0)
    batch = synthetic_batch(B=8, device=dev)
    m = build_module(dev)
    flat = lambda: torch.cat([p.detach().reshape(-1) for p in m.surrogate.parameters()]).clone()
    m.fused_step(batch)
    p1 = flat()
    m.fused_step(batch, lr=0.0)
    torch.cuda.synchronize(dev)
    assert torch.equal(flat(), p1)
    m.fused_step(batch, lr=1e-3)
    torch.cuda.synchronize(dev)
    assert not torch.equal(flat(), p1)


def test_longer_sequence_after_capture_keeps_the_captured_graph_valid():
    """A captured T = 20 step, then an eager backward with T = 45 (grows the partial-gradient buffers; five chunks run
    on side streams), then the T = 20 graph again: the graph's baked buffer addresses must still be alive and clean --
    its result must equal a fresh module that never saw the long batch."""
    from pdecontrol.surrogates.bench_tbptt import build_module, synthetic_batch
    dev = torch.device("cuda", 0)
    batch = synthetic_batch(B=8, device=dev)
    long_batch = synthetic_batch(B=8, T=45, device=dev)
    a, b = build_module(dev), build_module(dev)
    la = [float(a.fused_step(batch)["loss"])]
    lb = [float(b.fused_step(batch)["loss"])]
    out = a._eager_training_step(long_batch, 0)       # validation-style extra pass on the same surrogate
    out["loss"].backward()
    for p in a.surrogate.parameters():
        p.grad = None
    import gc
    gc.collect()
    torch.cuda.synchronize(dev)
    torch.cuda.empty_cache()
    junk = [torch.full((1 << 20,), float("nan"), device=dev) for _ in range(8)]   # reuse freed blocks, if any
    for _ in range(3):
        la.append(float(a.fused_step(batch)["loss"]))
        lb.append(float(b.fused_step(batch)["loss"]))
    torch.cuda.synchronize(dev)
    del junk
    assert la == lb
    for p, q in zip(a.surrogate.parameters(), b.surrogate.parameters()):
        assert torch.equal(p, q)


def test_lightning_route_reaches_the_graphed_step():
    """graphed=True: the (shim) Lightning loop only calls training_step, which replays the captured graph; same
    parameters as driving fused_step by hand, metrics logged, StepLR acting through the device learning rate."""
    from pdecontrol._compat.lightning import IS_SHIM, pl
    from pdecontrol.surrogates.bench_tbptt import build_module, synthetic_batch
    if not IS_SHIM:
        pytest.skip("real pytorch-lightning present: covered by its own manual-optimization loop")
    dev = torch.device("cuda", 0)
    batch = synthetic_batch(B=8, device=dev)
    ref = build_module(dev)
    for _ in range(3):
        ref.fused_step(batch)
    m = build_module(dev)
    m.graphed, m.automatic_optimization = True, False
    tr = pl.Trainer(max_steps=3, max_epochs=1)
    tr.fit(m, train_dataloaders=[batch] * 4)
    torch.cuda.synchronize(dev)
    assert tr.global_step == 3 and len(m._graphed_steps) == 1
    assert float(tr.callback_metrics["Train Loss"]) == float(ref._last_graphed_step.result["loss"])
    for p, q in zip(m.surrogate.parameters(), ref.surrogate.parameters()):
        assert torch.equal(p, q)


# ---- N = 256 (BASELINE configs[2]): against tensors produced by the reference's own building blocks at N = 256 ----
@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("scaled", [False, True])
def test_n256_training_step_against_reference_fixture(fused, scaled):
    from pdecontrol.surrogates import ops
    from test_surrogate_host import N256_GOLDEN, build_n256, n256_batch
    g, tag = np.load(N256_GOLDEN), ("b4n" if scaled else "b4")
    dev = torch.device("cuda", 0)
    m = build_n256(scaled, dev)
    s, a = (t.to(dev) for t in n256_batch(4))
    with ops.fused(fused):
        res = m.training_step((s, a), 0)
        res["loss"].backward()
    torch.cuda.synchronize(dev)
    rel = abs(res["loss"].item() - float(g[f"{tag}_loss"])) / float(g[f"{tag}_loss"])
    assert rel < 1e-5, rel
    np.testing.assert_allclose(res["hsteploss"].cpu().numpy(), g[f"{tag}_hsteploss"], rtol=1e-4)
    np.testing.assert_allclose(res["outputs"].cpu().numpy(), g[f"{tag}_outputs"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(res["outdeltas"].cpu().numpy(), g[f"{tag}_outdeltas"], rtol=1e-3, atol=1e-4)
    check_grads(f"n256 {tag} training_step fused={fused}", _named_grads(m), lambda k: g[f"{tag}_grad/" + k])


def test_n256_benchmarked_batch_b64_fused():
    """The bench's N = 256 TBPTT leg (B = 64, T = 20, Normalize scaling) on the fused kernels: loss within 1e-5 relative
    of the reference classes' CPU result, gradients within fp32 summation-order noise."""
    from pdecontrol.surrogates import ops
    from test_surrogate_host import N256_GOLDEN, build_n256, n256_batch
    g = np.load(N256_GOLDEN)
    dev = torch.device("cuda", 0)
    m = build_n256(True, dev)
    s, a = (t.to(dev) for t in n256_batch(64))
    assert ops.use_fused(s)
    res = m.training_step((s, a), 0)
    res["loss"].backward()
    torch.cuda.synchronize(dev)
    rel = abs(res["loss"].item() - float(g["b64n_loss"])) / float(g["b64n_loss"])
    assert rel < 1e-5, rel
    np.testing.assert_allclose(res["hsteploss"].cpu().numpy(), g["b64n_hsteploss"], rtol=1e-4)
    check_grads("n256 b64n training_step fused", _named_grads(m), lambda k: g["b64n_grad/" + k])


def test_n256_pipelined_pass_b64_against_reference_fixture():
    """The captured step's pipelined pass at the bench's TBPTT headline (N = 256, B = 64, T = 20, Normalize scaling) against
    the reference classes' CPU loss / per-step loss / gradients."""
    from test_surrogate_host import N256_GOLDEN, build_n256, n256_batch
    g = np.load(N256_GOLDEN)
    dev = torch.device("cuda", 0)
    m = build_n256(True, dev)
    s, a = (t.to(dev) for t in n256_batch(64))
    res = m._pipelined_training_step((s, a))
    assert res is not None
    torch.cuda.synchronize(dev)
    rel = abs(res["loss"].item() - float(g["b64n_loss"])) / float(g["b64n_loss"])
    assert rel < 1e-5, rel
    np.testing.assert_allclose(res["hsteploss"].cpu().numpy(), g["b64n_hsteploss"], rtol=1e-4)
    check_grads("n256 b64n pipelined pass", _named_grads(m), lambda k: g["b64n_grad/" + k])


def test_validation_and_test_step_on_gpu_against_reference_module():
    """validation_step / test_step with the module on the GPU (fused rollout) and env.rhs on the HIP hook (one batched call
    for all samples) against the arrays the reference's module produced on the CPU."""
    from pdegym.kuramoto import KuramotoSivashinskyEnv
    from test_surrogate_host import build_eval_module, check_eval_steps
    dev = torch.device("cuda", 0)
    module, g = build_eval_module(KuramotoSivashinskyEnv(), dev)
    check_eval_steps(module, g, dev, rtol=2e-4, atol=2e-5)
