"""Surrogate mirror (pdecontrol.*) on CPU against golden tensors captured from the reference.
The CPU path is plain torch, i.e. the fp32 torch reference the GPU kernels are compared to."""
import sys
import os

import numpy as np
import pytest
import torch

import pdecontrol.architectures as architectures
from pdecontrol.architectures import KSAutoRegConvolutionalLSTM, KSAutoRegConvolutionalLSTMN
from pdecontrol.mbrl.types import ModelRollout
from pdecontrol.surrogates.surrogate import AutoRegPDESurrogate, PDEEnsemble, action_and_target_indices
from pdecontrol.surrogates.training import PDETrainingModule
from pdegym.common.transforms import BatchTransform, Normalize


def build(dscaling=None, undscaling=None, factory_cls=KSAutoRegConvolutionalLSTM, **model_kw):
    torch.manual_seed(0)
    factory = factory_cls()
    surrogate = factory.surrogate(delta=0.25, dscaling=dscaling, tau=5, **factory.model(**model_kw))
    module = PDETrainingModule(surrogate=surrogate, loss=torch.nn.MSELoss(reduction="none"), tstep=0.25, delta=0.25,
                               undscaling=undscaling, tau=5, tbtt=10)
    return surrogate, module


def test_registry_and_factory_api():
    f = getattr(architectures, "KSAutoRegConvolutionalLSTM")()  # script.py:91 style lookup
    d = f.defaults
    assert set(d.keys()) == {"model", "surrogate", "training", "trainer", "curriculum"} and d.model == {}
    model = f.model(N=64, L=22.0, cfg_steps=250)  # scenario kwargs are swallowed
    assert set(model) == {"state_encoder", "state_decoder", "action_encoder", "transition_model"}
    s = f.surrogate(delta=0.25, dscaling=None, tau=5, N=64, **model)
    assert isinstance(s, AutoRegPDESurrogate)


def test_state_dict_keys_init_and_param_count(sur_golden):
    surrogate, _ = build()
    sd = surrogate.state_dict()
    keys = [k[3:] for k in sur_golden.files if k.startswith("sd/")]
    assert list(sd.keys()) == keys
    for k in keys:  # same construction order -> same RNG stream -> identical initial weights
        np.testing.assert_array_equal(sd[k].numpy(), sur_golden["sd/" + k], err_msg=k)
    assert sum(p.numel() for p in surrogate.parameters() if p.requires_grad) == int(sur_golden["n_trainable"]) == 9739
    surrogate.load_state_dict({k: torch.from_numpy(sur_golden["sd/" + k]) for k in keys})  # checkpoints load


def test_training_step_matches_reference_bitwise(sur_golden):
    g = sur_golden
    surrogate, module = build()
    batch = (torch.from_numpy(g["b8_states"]), torch.from_numpy(g["b8_actions"]))
    res = module.training_step(batch, 0)
    assert set(res) == {"loss", "hsteploss", "outputs", "actions", "states", "outdeltas", "deltas"}
    res["loss"].backward()
    assert res["loss"].item() == g["b8_loss"]
    for key in ("hsteploss", "outputs", "outdeltas", "deltas"):
        np.testing.assert_array_equal(res[key].numpy(), g["b8_" + key], err_msg=key)
    assert res["outputs"].shape == (8, 20, 1, 64) and res["outdeltas"].shape == (8, 19, 1, 64)
    for k, p in surrogate.named_parameters():
        if p.grad is not None:
            np.testing.assert_array_equal(p.grad.numpy(), g["b8_grad/" + k], err_msg=k)
    assert "Train Loss" in module.logged
    # one Adam step (configure_optimizers) and the loss again
    opt = module.configure_optimizers()[0][0]
    opt.step()
    opt.zero_grad()
    assert module.training_step(batch, 1)["loss"].item() == g["b8_loss_after_adam"]


def test_training_step_with_normalize_scaling(sur_golden):
    g = sur_golden
    norm = Normalize(aggregate=True, batched=True)
    norm.mean, norm.var, norm.count = torch.full((1, 1, 1), 0.01), torch.full((1, 1, 1), 0.5), 100
    und = BatchTransform(norm)
    surrogate, module = build(dscaling=und.Inverse, undscaling=und)
    res = module.training_step((torch.from_numpy(g["b8_states"]), torch.from_numpy(g["b8_actions"])), 0)
    res["loss"].backward()
    assert res["loss"].item() == g["b8n_loss"]
    for key in ("hsteploss", "outputs", "outdeltas", "deltas"):
        np.testing.assert_array_equal(res[key].numpy(), g["b8n_" + key], err_msg=key)
    for k, p in surrogate.named_parameters():
        if p.grad is not None:
            np.testing.assert_array_equal(p.grad.numpy(), g["b8n_grad/" + k], err_msg=k)


def test_known_answer_b64(sur_golden):
    surrogate, module = build()
    gen = torch.Generator().manual_seed(1)
    s = torch.rand(64, 20, 1, 64, generator=gen) * 2 - 1
    a = torch.rand(64, 20, 1, 64, generator=gen) * 2 - 1
    res = module.training_step((s, a), 0)
    res["loss"].backward()
    assert res["loss"].item() == 10.806351661682129 == float(sur_golden["b64_loss"])
    gn = torch.sqrt(sum((p.grad ** 2).sum() for p in surrogate.parameters() if p.grad is not None)).item()
    np.testing.assert_allclose(gn, 1.3024945259094238, rtol=1e-6)
    np.testing.assert_array_equal(res["hsteploss"].numpy(), sur_golden["b64_hsteploss"])


def test_rollout_api_hidden_carry_and_inlatents(sur_golden):
    g = sur_golden
    surrogate, _ = build()
    s, a = torch.from_numpy(g["b8_states"]), torch.from_numpy(g["b8_actions"])
    with torch.no_grad():
        times, targets = 0.25 * torch.arange(10), 0.25 * (torch.arange(10) + 1)
        r1 = surrogate.rollout(states=s[:, :5], actions=a[:, :10], times=times, targets=targets, hidden=None)
        r2 = surrogate.rollout(states=r1.outputs[:, -1, None], actions=a[:, 10:], times=times, targets=targets,
                               hidden=r1.hidden)
    assert isinstance(r1, ModelRollout) and isinstance(r1.hidden, tuple) and len(list(r1)) == 5
    for name, got in (("outputs", r1.outputs), ("deltas", r1.deltas), ("inlatents", r1.inlatents),
                      ("outlatents", r1.outlatents), ("H", r1.hidden[0]), ("C", r1.hidden[1])):
        np.testing.assert_array_equal(got.numpy(), g["ro1_" + name], err_msg=name)
    np.testing.assert_array_equal(r2.outputs.numpy(), g["ro2_outputs"])
    np.testing.assert_array_equal(r2.hidden[0].numpy(), g["ro2_H"])
    surrogate.reencode_predictions = False  # cheaper mode: same outputs, no inlatents
    with torch.no_grad():
        r3 = surrogate.rollout(states=s[:, :5], actions=a[:, :10], times=times, targets=targets, hidden=None)
    assert r3.inlatents is None
    np.testing.assert_array_equal(r3.outputs.numpy(), g["ro1_outputs"])


@pytest.mark.parametrize("name", ["train10", "train15", "scalar", "warm5", "coarse"])
def test_integer_index_paths_bit_exact(sur_golden, name):
    g = sur_golden
    aidx, tidx = action_and_target_indices(g[f"idx_{name}_times"], g[f"idx_{name}_targets"], 0.25)
    np.testing.assert_array_equal(aidx.numpy(), g[f"idx_{name}_aidx"])
    np.testing.assert_array_equal(tidx.numpy(), g[f"idx_{name}_tidx"])


def test_validation_step_and_ensemble():
    surrogate, module = build()
    gen = torch.Generator().manual_seed(3)
    s = torch.rand(4, 12, 1, 64, generator=gen)
    a = torch.rand(4, 12, 1, 64, generator=gen)
    with torch.no_grad():
        res = module.validation_step((s, a), 0)
    assert res["outputs"].shape == (4, 12, 1, 64) and "Val. Loss" in module.logged
    torch.testing.assert_close(res["outputs"][:, 0], s[:, 0])
    _, m2 = build()
    ens = PDEEnsemble([module, m2], num_elites=1)
    ens.update_elites([0.7, 0.2])
    assert ens.elite_idx == [1]
    np.random.seed(0)
    with torch.no_grad():
        out = ens.rollout(s[:, :5], a[:, :5], 0.25 * torch.arange(5), torch.tensor(1.25))
        ref = m2.surrogate.rollout(s[:, :5], a[:, :5], 0.25 * torch.arange(5), torch.tensor(1.25))
    torch.testing.assert_close(out.outputs, ref.outputs)  # only elite #1 can be chosen
    assert len(out.hidden) == 2


def test_size_parametrised_factory_n256():
    surrogate, module = build(factory_cls=KSAutoRegConvolutionalLSTMN, N=256)
    gen = torch.Generator().manual_seed(3)
    s = torch.rand(2, 20, 1, 256, generator=gen)
    a = torch.rand(2, 20, 1, 256, generator=gen)
    res = module.training_step((s, a), 0)
    res["loss"].backward()
    assert res["outputs"].shape == (2, 20, 1, 256) and torch.isfinite(res["loss"])
    s64, _ = build(factory_cls=KSAutoRegConvolutionalLSTMN, N=64)
    ref, _ = build()
    for (k1, v1), (k2, v2) in zip(s64.state_dict().items(), ref.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)


def test_tbtt_must_exceed_tau():
    torch.manual_seed(0)
    f = KSAutoRegConvolutionalLSTM()
    s = f.surrogate(delta=0.25, dscaling=None, tau=5, **f.model())
    with pytest.raises(AssertionError):
        PDETrainingModule(surrogate=s, loss=torch.nn.MSELoss(reduction="none"), tstep=0.25, delta=0.25, tau=5, tbtt=5)


def test_fused_path_env_switch(monkeypatch):
    """The fused kernels are the default CUDA path; PDECONTROL_FUSED=0 opts out at import; CPU tensors never take it."""
    import importlib
    from pdecontrol.surrogates import ops
    monkeypatch.setenv("PDECONTROL_FUSED", "0")
    try:
        importlib.reload(ops)
        assert not ops.fused_enabled()
    finally:
        monkeypatch.delenv("PDECONTROL_FUSED")
        importlib.reload(ops)
    assert ops.fused_enabled()
    assert not ops.use_fused(torch.zeros(2))      # CPU tensor: plain torch path, no library load
    with ops.fused(False):
        assert not ops.fused_enabled()
    assert ops.fused_enabled()


def test_shim_trainer_follows_lightning_closure_order_and_manual_optimization():
    """The in-repo Trainer drives training_step -> zero_grad -> backward -> step (pytorch-lightning's closure order) and,
    for automatic_optimization = False, leaves everything to training_step."""
    from pdecontrol._compat.lightning import IS_SHIM, pl
    if not IS_SHIM:
        pytest.skip("real pytorch-lightning present")
    torch.manual_seed(0)
    f = KSAutoRegConvolutionalLSTM()
    s = f.surrogate(delta=0.25, dscaling=None, tau=5, **f.model())
    m = PDETrainingModule(surrogate=s, loss=torch.nn.MSELoss(reduction="none"), tstep=0.25, delta=0.25, tau=5, tbtt=10)
    gen = torch.Generator().manual_seed(3)
    batch = (torch.rand(2, 12, 1, 64, generator=gen), torch.rand(2, 12, 1, 64, generator=gen))
    before = torch.cat([p.detach().reshape(-1) for p in s.parameters()]).clone()
    tr = pl.Trainer(max_steps=2, max_epochs=1)
    tr.fit(m, train_dataloaders=[batch, batch, batch])
    assert tr.global_step == 2 and "Train Loss" in tr.callback_metrics
    after = torch.cat([p.detach().reshape(-1) for p in s.parameters()])
    assert not torch.equal(before, after)
    assert PDETrainingModule(surrogate=s, loss=torch.nn.MSELoss(reduction="none"), tstep=0.25, delta=0.25, tau=5, tbtt=10,
                             graphed=True).automatic_optimization is False


# ---- N = 256 (BASELINE configs[2] / [3]): KSAutoRegConvolutionalLSTMN against the reference's own building blocks ----
N256_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "surrogate_n256_golden.npz")


def build_n256(scaled, device="cpu"):
    """Same construction as oracle/gen_golden.py::surrogate_n256_fixtures, from this repo's classes."""
    from pdecontrol.architectures import KSAutoRegConvolutionalLSTMN
    from pdegym.common.transforms import BatchTransform, Normalize
    torch.manual_seed(0)
    f = KSAutoRegConvolutionalLSTMN()
    model = f.model(N=256)
    und = None
    if scaled:
        norm = Normalize(aggregate=True, batched=True)
        norm.mean, norm.var, norm.count = torch.full((1, 1, 1), 0.01), torch.full((1, 1, 1), 0.5), 100
        und = BatchTransform(norm)
    s = f.surrogate(delta=0.25, dscaling=None if und is None else und.Inverse, tau=5, **model)
    m = PDETrainingModule(surrogate=s, loss=torch.nn.MSELoss(reduction="none"), tstep=0.25, delta=0.25, undscaling=und,
                          tau=5, tbtt=10)
    return m.to(device)


def n256_batch(B):
    g = torch.Generator().manual_seed(1)
    s = torch.rand(64, 20, 1, 256, generator=g) * 2 - 1
    a = torch.rand(64, 20, 1, 256, generator=g) * 2 - 1
    return s[:B].clone(), a[:B].clone()


def test_n256_factory_matches_reference_building_blocks():
    """state_dict (same seed -> same initial weights), parameter count, loss / outputs / gradients of a TBPTT step at
    N = 256, identity and Normalize scaling, against tensors produced by the reference's ConvNet / ResidualBlock /
    CNNLSTMTransitionModel / AutoRegPDESurrogate / PDETrainingModule classes (models/cnn.py:73-173,
    transition.py:229-296) instantiated at N = 256."""
    g = np.load(N256_GOLDEN)
    m = build_n256(False)
    sd = m.surrogate.state_dict()
    keys = [k[3:] for k in g.files if k.startswith("sd/")]
    assert sorted(sd.keys()) == sorted(keys)
    for k in keys:
        np.testing.assert_array_equal(sd[k].numpy(), g["sd/" + k], err_msg=k)
    assert sum(p.numel() for p in m.surrogate.parameters() if p.requires_grad) == int(g["n_trainable"])
    s4, a4 = n256_batch(4)
    for scaled, tag in ((False, "b4"), (True, "b4n")):
        m = build_n256(scaled)
        res = m.training_step((s4, a4), 0)
        res["loss"].backward()
        assert res["loss"].item() == float(g[f"{tag}_loss"])
        np.testing.assert_array_equal(res["outputs"].numpy(), g[f"{tag}_outputs"])
        np.testing.assert_array_equal(res["outdeltas"].numpy(), g[f"{tag}_outdeltas"])
        for k, p in m.surrogate.named_parameters():
            if p.grad is not None:
                np.testing.assert_allclose(p.grad.numpy(), g[f"{tag}_grad/" + k], rtol=1e-5, atol=1e-7, err_msg=k)


# ---- validation_step / test_step (pdecontrol/surrogates/training.py:132-271) against the reference's module -------------
EVAL_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "evalstep_golden.npz")


def build_eval_module(env, device="cpu"):
    """The offline-evaluation setup (pdecontrol/surrogates/evaluation/evaluate.py:86-164), as in
    oracle/gen_golden.py::evalstep_fixtures, from this repo's classes."""
    from pdegym.common import transforms as T
    g = np.load(EVAL_GOLDEN)
    oscaling = T.Normalize(aggregate=True, batched=True)
    forcing = T.BatchTransform(env.forcing)
    pdescaling = T.Normalize(aggregate=True, batched=True)
    oscaling.update(g["raw_states"])
    pdescaling.update(forcing(g["raw_actions"]))
    stransf = T.SampleTransform(oscaling, T.Operation([forcing, pdescaling]))
    torch.manual_seed(0)
    f = KSAutoRegConvolutionalLSTM()
    sur = f.surrogate(delta=0.25, dscaling=None, tau=5, **f.model())
    module = PDETrainingModule(surrogate=sur, loss=torch.nn.MSELoss(reduction="none"), tstep=0.25, delta=0.25, env=env,
                               stransf=stransf, tau=5, tbtt=10)
    return module.to(device), g


def check_eval_steps(module, g, device="cpu", rtol=1e-6, atol=1e-7):
    s, a = torch.from_numpy(g["states"]).to(device), torch.from_numpy(g["actions"]).to(device)
    with torch.no_grad():
        val = module.validation_step((s, a), 0)
        tst = module.test_step((s, a), 0)
    for k, v in val.items():
        np.testing.assert_allclose(np.asarray(v.detach().cpu().numpy()), g["val_" + k], rtol=rtol, atol=atol, err_msg="val " + k)
    assert sorted("test_" + k for k in tst) == sorted(k for k in g.files if k.startswith("test_"))
    for k, v in tst.items():
        np.testing.assert_allclose(np.asarray(v), g["test_" + k], rtol=rtol, atol=atol, err_msg="test " + k)


def test_validation_and_test_step_match_reference_module():
    """Every array validation_step / test_step return (losses per horizon step, scaled errors, reward errors, errors of
    the three spatial derivatives from env.rhs) against the reference module's, env.rhs on the oracle-backed stepper."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _oracle_stepper import OracleStepper
    from pdegym.kuramoto import KuramotoSivashinskyEnv
    module, g = build_eval_module(KuramotoSivashinskyEnv(_stepper_cls=OracleStepper))
    check_eval_steps(module, g)


def test_scaling_signature_notices_refitted_delta_statistics():
    """Graph caches compare ``hipops.scaling_signature`` (host logic, no GPU): the controller's ``update_delta_transform``
    (mbrl.py:597-602: reset() + update() on the shared Normalize) must change it, an untouched transform must not."""
    from pdecontrol.surrogates import hipops
    from pdecontrol.surrogates.bench_tbptt import build_module
    m = build_module("cpu")
    a = hipops.scaling_signature(m.surrogate, m.undscaling)
    assert hipops.same_signature(a, hipops.scaling_signature(m.surrogate, m.undscaling))
    norm = m.undscaling.transform
    norm.reset()
    norm.update(torch.linspace(-1.0, 2.0, 32).reshape(32, 1, 1))
    b = hipops.scaling_signature(m.surrogate, m.undscaling)
    assert not hipops.same_signature(a, b)
    assert hipops.same_signature(b, hipops.scaling_signature(m.surrogate, m.undscaling))
    # the surrogate's dscaling wraps the same Normalize object: both entries changed
    assert a[0][0] is a[1][0] and b[0][1] is not a[0][1]
