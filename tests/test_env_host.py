"""Host logic of the gym env mirror (pdegym.kuramoto) against the golden vectors, with the
oracle-backed stepper double standing in for the GPU.  The same checks run against the real HIP
stepper in tests/test_env_gpu.py (-m gpu)."""
import numpy as np
import pytest

import pdegym  # noqa: F401
from pdegym._gym import gym
from pdegym.kuramoto import KuramotoSivashinskyEnv, make, make_vec, shard_envs
from _oracle_stepper import OracleStepper

from conftest import KS_CONFIGS


def new_env(**cfg):
    return KuramotoSivashinskyEnv(_stepper_cls=OracleStepper, **cfg)


def test_constants_spaces_scenario(ks_golden):
    env = new_env()
    L, N, dx, dt, cfg_steps, max_steps = ks_golden["default_consts"]
    assert (env.L, env.N, env.dx, env.dt, env.cfg_steps, env.max_episode_steps) == (L, N, dx, dt, cfg_steps, max_steps)
    np.testing.assert_array_equal(env.x, ks_golden["default_x"])
    assert env.x.dtype == np.float32
    assert env.action_space.shape == (1, 4) and env.action_space.dtype == np.float32
    assert env.observation_space.shape == (1, 64) and env.observation_space.dtype == np.float32
    assert set(env.scenario) == {"cfg_steps", "Ttrans", "L", "N", "dx", "Tmax", "dt", "Xi", "noise", "lmbda",
                                 "objective"}
    assert env.scenario["lmbda"] == 1.0 and env.scenario["noise"] == 0.1  # constants, as in the reference
    assert env.unwrapped is env and env.time == 0.0
    assert env.reward_func.transf.__name__ == "_l2control"  # truthy string selects l2control
    assert env._stepper is None  # constructing must not touch the device


@pytest.mark.parametrize("tag", list(KS_CONFIGS))
def test_forcing_transform_matches_reference(ks_golden, tag):
    L, N = KS_CONFIGS[tag]
    env = new_env(L=L, N=N)
    np.testing.assert_array_equal(env.forcing.forcing.numpy(), ks_golden[f"{tag}_F"])
    for a, phi in zip(ks_golden[f"{tag}_actions"], ks_golden[f"{tag}_phi"]):
        got = np.squeeze(env.forcing(a))
        assert got.dtype == np.float32
        np.testing.assert_array_equal(got, phi)
    inv = env.forcing.Inverse
    np.testing.assert_array_equal(inv.xpos.numpy(), ks_golden[f"{tag}_xpos"])
    np.testing.assert_array_equal(inv.inv_forcing.numpy(), ks_golden[f"{tag}_invF"])
    got = np.stack([inv(p[None, :]) for p in ks_golden[f"{tag}_phi"]])
    np.testing.assert_array_equal(got, ks_golden[f"{tag}_phi_inv"])


@pytest.mark.parametrize("tag", list(KS_CONFIGS))
def test_step_matches_reference(ks_golden, tag):
    L, N = KS_CONFIGS[tag]
    for n in (1, 10, 250):
        for e in range(3):
            env = new_env(L=L, N=N, cfg_steps=n)
            env.u = ks_golden[f"{tag}_traj_u0"][e]
            env.timestep = 0
            obs, rew, term, trunc, info = env.step(ks_golden[f"{tag}_actions"][e])
            assert obs.dtype == np.float64 and obs.shape == (1, N)
            np.testing.assert_array_equal(obs[0], ks_golden[f"{tag}_traj_u{n}"][e])
            np.testing.assert_allclose(rew, ks_golden[f"{tag}_traj_rew{n}"][e], rtol=1e-13)
            assert term is False and not trunc and info == {"step": 1}


def test_two_steps_and_known_answer(ks_golden):
    env = new_env()
    env.u = ks_golden["n64_traj_u0"][0]
    env.step(ks_golden["n64_actions"][0])
    obs, rew, _, _, info = env.step(ks_golden["n64_actions"][1])
    np.testing.assert_array_equal(obs[0], ks_golden["n64_two_steps_u"])
    np.testing.assert_allclose(rew, ks_golden["n64_two_steps_rew"], rtol=1e-13)
    assert info == {"step": 2}
    env = new_env()
    env.u = ks_golden["seed0_u0"]
    obs, rew, _, _, _ = env.step([[0.3, -0.7, 1.0, -1.0]])
    np.testing.assert_array_equal(obs[0], ks_golden["seed0_u250"])
    np.testing.assert_allclose(rew, -0.01171066866857959, rtol=1e-13)


def test_episode_boundary_sequence(ks_golden):
    env = new_env(Tmax=1.0, cfg_steps=50)
    assert env.max_episode_steps == 20
    env.u = ks_golden["n64_traj_u0"][1]
    env.timestep = 17
    seq = []
    for _ in range(4):
        _, _, term, trunc, info = env.step([[0.1, 0.2, -0.3, 0.4]])
        seq.append((int(term), int(trunc), info["step"]))
    np.testing.assert_array_equal(np.asarray(seq), ks_golden["episode_seq"])


def test_seeded_reset_matches_reference(ks_golden):
    env = new_env()
    obs, info = env.reset(seed=int(ks_golden["n64_reset_seed"]), return_info=True)
    assert info == {"step": 0} and obs.shape == (1, 64) and obs.dtype == np.float64
    np.testing.assert_array_equal(obs[0], ks_golden["n64_reset_u"])
    assert env.timestep == 0
    assert env.reset(seed=1).shape == (1, 64)  # return_info=False returns the bare observation


def test_overflow_raises_floating_point_error():
    env = new_env(L=22.0, N=256)
    env.u = np.random.RandomState(0).uniform(-0.4, 0.4, 256)
    with pytest.raises(FloatingPointError):
        env.step([[0.0, 0.0, 0.0, 0.0]])


def test_rhs_api(ks_golden):
    env = new_env()
    u, phi = ks_golden["n64_rhs_u"][0], ks_golden["n64_rhs_phi"][0]
    rhs, (ux, uxx, uxxxx) = env.rhs(u, phi)
    np.testing.assert_array_equal(rhs, ks_golden["n64_rhs"][0])
    np.testing.assert_array_equal(ux, ks_golden["n64_ux"][0])
    rhs2, _ = env.rhs(u[None, :], phi[None, :])  # [1, N] as training.test_step passes it
    assert rhs2.shape == (1, 64)
    np.testing.assert_array_equal(rhs2[0], rhs)


def test_reward_func_numpy_and_torch():
    import torch
    env = new_env()
    u = np.random.RandomState(1).uniform(-1, 1, (1, 64))
    r = env.reward_func(u, np.zeros(64, np.float32))
    assert isinstance(r, np.ndarray) and r.shape == ()
    np.testing.assert_allclose(r, -(u ** 2).sum() / 64, rtol=1e-14)
    rt = env.reward_func(torch.from_numpy(u).float(), torch.zeros(64))
    assert isinstance(rt, torch.Tensor)
    np.testing.assert_allclose(rt.item(), r, rtol=1e-6)


def test_make_and_registry_time_limit():
    env = make(config={"_stepper_cls": OracleStepper, "Tmax": 0.5, "cfg_steps": 50})
    assert env.unwrapped.max_episode_steps == 10
    env.unwrapped.u = np.zeros(64)
    env._elapsed_steps = 0
    trunc = False
    for i in range(10):
        _, _, _, trunc, info = env.step(np.zeros((1, 4), np.float32))
    assert trunc and info["step"] == 10
    env2 = gym.make("KuramotoSivashinskyEnv-v0", config={"_stepper_cls": OracleStepper})
    assert env2.unwrapped.N == 64


def test_product_env_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    env = KuramotoSivashinskyEnv()
    with pytest.raises(Exception) as ei:
        env.reset(seed=0)
    assert "kspde" in type(ei.value).__module__ or "HIP" in str(ei.value) or "libkspde" in str(ei.value)


# ---------------------------------------------------------------------------------------------
# batched vector env
# ---------------------------------------------------------------------------------------------
def test_vec_env_contract_and_parity_with_single_envs():
    E = 5
    cfg = {"Tmax": 0.03, "cfg_steps": 10}  # 3 steps per episode
    vec = make_vec(E, config=cfg, burn_in=False, _stepper_cls=OracleStepper)
    assert vec.num_envs == E and vec.max_episode_steps == 3
    assert vec.single_action_space.shape == (1, 4) and vec.action_space.shape == (E, 1, 4)
    assert vec.observation_space.shape == (E, 1, 64) and vec.observation_space.dtype == np.float32
    obs, info = vec.reset(seed=100, return_info=True)
    assert obs.shape == (E, 1, 64) and obs.dtype == np.float32
    np.testing.assert_array_equal(info["step"], np.zeros(E, dtype=np.int64))
    # IC of env i == what a single env draws with seed 100 + i (no burn-in in this test)
    singles = []
    for i in range(E):
        s = new_env(**cfg)
        np.random.seed(100 + i)
        s.u = np.random.uniform(-0.4, 0.4, 64)
        s.timestep = 0
        singles.append(s)
        np.testing.assert_array_equal(obs[i, 0], s.u.astype(np.float32))
    rs = np.random.RandomState(0)
    for t in range(1, 4):
        actions = rs.uniform(-1, 1, (E, 1, 4)).astype(np.float32)
        o, r, term, trunc, infos = vec.step(actions)
        assert o.dtype == np.float32 and r.dtype == np.float64 and term.dtype == bool and trunc.dtype == bool
        ref = [s.step(a) for s, a in zip(singles, actions)]
        np.testing.assert_allclose(r, [x[1] for x in ref], rtol=1e-13)
        np.testing.assert_array_equal(infos["step"], [x[4]["step"] for x in ref])
        np.testing.assert_array_equal(trunc, [x[3] for x in ref])
        assert not term.any()
        if t < 3:
            np.testing.assert_array_equal(o[:, 0], np.stack([x[0][0] for x in ref]).astype(np.float32))
            assert "final_observation" not in infos
        else:
            # autoreset: every env truncates on the same step
            assert trunc.all() and infos["_final_observation"].all()
            finals = np.stack(list(infos["final_observation"]))
            np.testing.assert_array_equal(finals[:, 0], np.stack([x[0][0] for x in ref]).astype(np.float32))
            np.testing.assert_array_equal(infos["step"], np.full(E, 3))
            assert np.abs(o).max() <= 0.4  # fresh ICs (burn_in=False)
            np.testing.assert_array_equal(vec.timestep, np.zeros(E))
    vec.close()
    assert vec.stepper.closed


def test_vec_env_partial_autoreset_and_burn_in():
    # envs with different elapsed steps truncate at different times; burn-in enabled (2 envs)
    vec = make_vec(2, config={"Tmax": 0.02, "cfg_steps": 10, "L": 22.0, "N": 64}, _stepper_cls=OracleStepper)
    assert vec.burn_in_substeps == 200000
    obs = vec.reset(seed=[3, 4])
    vec.timestep[:] = [1, 0]
    o, r, term, trunc, infos = vec.step(np.zeros((2, 1, 4), np.float32))
    np.testing.assert_array_equal(trunc, [True, False])
    assert infos["final_observation"][1] is None and infos["final_observation"][0].shape == (1, 64)
    np.testing.assert_array_equal(infos["_final_observation"], [True, False])
    np.testing.assert_array_equal(infos["step"], [2, 1])
    np.testing.assert_array_equal(vec.timestep, [0, 1])
    assert np.abs(o[0]).max() > 0.4  # env 0 was reset AND burnt in onto the attractor


def test_vec_env_seeded_reset_equals_reference_reset(ks_golden):
    vec = make_vec(1, _stepper_cls=OracleStepper)
    obs = vec.reset(seed=int(ks_golden["n64_reset_seed"]))
    np.testing.assert_array_equal(obs[0, 0], ks_golden["n64_reset_u"].astype(np.float32))


def test_vec_env_overflow_raises():
    vec = make_vec(3, config={"L": 22.0, "N": 256}, burn_in=False, _stepper_cls=OracleStepper)
    vec.reset(seed=0)
    with pytest.raises(FloatingPointError):
        vec.step(np.zeros((3, 1, 4), np.float32))


def test_shard_envs_partition():
    for E in (1, 7, 32768):
        for W in (1, 2, 3, 8):
            spans = [shard_envs(E, r, W) for r in range(W)]
            assert spans[0][0] == 0 and spans[-1][1] == E
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_batched_vector_make_override(monkeypatch):
    import pdegym as pg
    if not hasattr(gym.vector, "make"):
        monkeypatch.setattr(gym.vector, "make", None, raising=False)
    pg.install_batched_vector_make(device=0)
    vec = gym.vector.make("KuramotoSivashinskyEnv-v0", num_envs=4, burn_in=False, _stepper_cls=OracleStepper)
    assert vec.num_envs == 4 and vec.observation_space.shape == (4, 1, 64)
