"""f4: viscous-Burgers path.  The reference has no Burgers env; its discretisation is BurgersPhyPDELoss
(pdecontrol/surrogates/phyloss/phyloss.py:36-86).  tests/golden/burgers_golden.npz holds residual() / phyevolve() of
THAT class (oracle/gen_golden.py::burgers_fixtures): the oracle and the HIP kernel are pinned to it at fp32 rounding
(torch's convolution sums its taps in an order we do not restate: a few ulp of the field scale).  Everything the env adds
around the step is "parity unpinned" and checked through properties (order of accuracy, energy decay, determinism)."""
import os
import sys

import numpy as np
import pytest

from oracle import burgers_oracle as bo

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "burgers_golden.npz")
TAGS = ("n512", "n128")


def _close(a, b, ulps=8):
    scale = np.abs(b).max()
    np.testing.assert_allclose(a, b, rtol=0, atol=ulps * np.finfo(np.float32).eps * scale)


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_against_reference_class(tag):
    g = np.load(GOLDEN)
    dx, dt, nu, _ = g[f"{tag}_params"]
    u = g[f"{tag}_u"]
    _close(bo.residual(u, dx, nu), g[f"{tag}_residual"])
    _close(bo.evolve(u, dx, dt, nu), g[f"{tag}_evolve"])
    c = u.copy()
    for _ in range(10):
        c = bo.evolve(c, dx, dt, nu)
    _close(c, g[f"{tag}_evolve10"])


def test_stencil_orders_of_accuracy():
    """grad is 2nd order, laplace 4th order on sin(3x) (fp64 so that truncation, not rounding, is what is measured)."""
    eg, el = [], []
    for N in (32, 64, 128):
        L = 2 * np.pi
        x = np.linspace(0, L, N, endpoint=False)
        u = np.sin(3 * x)
        eg.append(np.abs(bo.grad(u, L / N, np.float64) - 3 * np.cos(3 * x)).max())
        el.append(np.abs(bo.laplace(u, L / N, np.float64) + 9 * np.sin(3 * x)).max())
    assert all(3.7 < a / b < 4.3 for a, b in zip(eg, eg[1:]))          # halving dx: error / 4
    assert all(14.0 < a / b < 18.0 for a, b in zip(el, el[1:]))        # halving dx: error / 16


def test_c_abi_exports_and_python_binding():
    import ctypes
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "burgers_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(bg_[a-z_]+)\s*\(", text)))
    lib = ctypes.CDLL(os.path.join(root, "model-based-pde-control_amd", "lib", "libburgers_hip.so"))
    for name in declared:
        assert hasattr(lib, name), name
    from pdegym.burgers import _hip
    assert sorted([n for n, _ in _hip.SYMBOLS] + ["bg_last_error"]) == declared
    lib.bg_last_error.restype = ctypes.c_char_p
    assert lib.bg_step(None, None, None, None, 0, 0, 0, ctypes.c_float(1), ctypes.c_float(1), ctypes.c_float(1), 0, None, None, None) < 0
    assert b"bad argument" in lib.bg_last_error()


def test_env_registration_without_gpu():
    import pdegym  # noqa: F401
    from pdegym._gym import gym
    env = gym.make("BurgersEnv-v0", new_step_api=True)
    assert env.unwrapped.N == 512 and env.unwrapped.max_episode_steps == 200
    assert env.observation_space.shape == (1, 512) and env.action_space.shape == (1, 4)


# ------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def vec():
    from pdegym.burgers import make_vec
    return make_vec


@pytest.mark.gpu
@pytest.mark.parametrize("tag", TAGS)
def test_kernel_against_reference_class(vec, tag):
    import torch
    g = np.load(GOLDEN)
    dx, dt, nu, L = g[f"{tag}_params"]
    u = g[f"{tag}_u"]
    env = vec(len(u), config=dict(L=float(L), N=u.shape[1], dt=float(dt), nu=float(nu)))
    _close(env.residual(u).cpu().numpy(), g[f"{tag}_residual"])
    env.u.copy_(torch.from_numpy(u))
    env.step_torch(None, n_substeps=1)
    _close(env.u.cpu().numpy(), g[f"{tag}_evolve"])
    env.u.copy_(torch.from_numpy(u))
    env.step_torch(None, n_substeps=10)
    _close(env.u.cpu().numpy(), g[f"{tag}_evolve10"], ulps=16)


@pytest.mark.gpu
@pytest.mark.parametrize("N", [64, 128, 256, 512, 1024])
def test_kernel_against_oracle_with_forcing_and_reward(vec, N):
    import torch
    E = 37                                    # not a multiple of the 4 envs per workgroup
    env = vec(E, config=dict(N=N, nu=0.02, dt=5e-4 if N < 1024 else 1e-4))
    env.reset(seed=3)
    u0 = env.u.cpu().numpy().copy()
    act = np.random.RandomState(1).uniform(-1, 1, (E, 4)).astype(np.float32)
    phi = act @ env.forcing.forcing.numpy()
    ref, ssq = bo.step(u0, phi, env.dx, env.dt, env.nu, 50)
    u, rew = env.step_torch(torch.from_numpy(act).to(env.device))
    np.testing.assert_allclose(u.cpu().numpy(), ref, rtol=0, atol=2e-5 * np.abs(ref).max())
    np.testing.assert_allclose(rew.cpu().numpy(), -(1 / N) * ssq / 50, rtol=1e-5)
    assert int(env._status.sum()) == 0


@pytest.mark.gpu
def test_env_contract_energy_decay_autoreset_and_overflow(vec):
    env = vec(5, config=dict(N=512, Tmax=0.15))            # 3 steps per episode
    obs = env.reset(seed=7)
    again = vec(5, config=dict(N=512, Tmax=0.15)).reset(seed=7)
    np.testing.assert_array_equal(obs, again)               # seeded resets are reproducible
    assert obs.shape == (5, 1, 512) and obs.dtype == np.float32
    energy = [(obs.astype(np.float64) ** 2).sum(axis=(1, 2))]
    for k in range(3):
        obs, rew, term, trunc, infos = env.step(np.zeros((5, 1, 4), np.float32))
        if k < 2:
            energy.append((obs.astype(np.float64) ** 2).sum(axis=(1, 2)))
            assert not trunc.any() and (rew <= 0).all() and rew.dtype == np.float64
    assert all((b <= a * (1 + 1e-6)).all() for a, b in zip(energy, energy[1:]))   # unforced viscous flow loses energy
    assert trunc.all() and infos["_final_observation"].all() and list(infos["step"]) == [3] * 5
    assert all(f.shape == (1, 512) for f in infos["final_observation"]) and (env.timestep == 0).all()
    wild = vec(2, config=dict(N=512, dt=0.5))               # far beyond the explicit stability limit
    wild.reset(seed=0)
    with pytest.raises(FloatingPointError):
        for _ in range(5):
            wild.step(np.ones((2, 1, 4), np.float32))
