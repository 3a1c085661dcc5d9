"""The gym env / batched vector env on the real HIP stepper (MI355X)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from conftest import KS_CONFIGS


@pytest.fixture(scope="module")
def api():
    import pdegym  # noqa: F401
    from pdegym.kuramoto import KuramotoSivashinskyEnv, make_vec
    return KuramotoSivashinskyEnv, make_vec


def test_single_env_step_and_reset_parity(api, ks_golden):
    Env, _ = api
    for tag in ("n64", "n256"):
        L, N = KS_CONFIGS[tag]
        # exact arithmetic: bit-identical observations
        env = Env(L=L, N=N, step_mode="exact", reset_mode="exact")
        env.u = ks_golden[f"{tag}_traj_u0"][0]
        obs, rew, term, trunc, info = env.step(ks_golden[f"{tag}_actions"][0])
        np.testing.assert_array_equal(obs[0], ks_golden[f"{tag}_traj_u250"][0])
        np.testing.assert_allclose(rew, ks_golden[f"{tag}_traj_rew250"][0], rtol=1e-13)
        assert obs.dtype == np.float64 and info == {"step": 1} and term is False and not trunc
        # default (fast) arithmetic: inside the 1e-9 per sub-step contract
        env = Env(L=L, N=N)
        assert env.step_mode == "fast" and env.reset_mode == "fast"
        env.u = ks_golden[f"{tag}_traj_u0"][0]
        obs, rew, *_ = env.step(ks_golden[f"{tag}_actions"][0])
        assert np.abs(obs[0] - ks_golden[f"{tag}_traj_u250"][0]).max() < 1e-9
        np.testing.assert_allclose(rew, ks_golden[f"{tag}_traj_rew250"][0], rtol=1e-10)
        # default reset: fast arithmetic -- another realisation of the same attractor (finite, same energy scale)
        obs = env.reset(seed=int(ks_golden[f"{tag}_reset_seed"]))
        ref = ks_golden[f"{tag}_reset_u"]
        assert np.isfinite(obs).all() and 0.5 < np.mean(obs[0] ** 2) / np.mean(ref ** 2) < 2.0
        # seeded reset = reference reset, bit for bit, with the parity switch (burn-in in exact arithmetic)
        env = Env(L=L, N=N, reset_mode="exact")
        obs, info = env.reset(seed=int(ks_golden[f"{tag}_reset_seed"]), return_info=True)
        np.testing.assert_array_equal(obs[0], ref)
        assert info == {"step": 0}
        rhs, (ux, uxx, uxxxx) = env.rhs(ks_golden[f"{tag}_rhs_u"][0], ks_golden[f"{tag}_rhs_phi"][0])
        np.testing.assert_array_equal(rhs, ks_golden[f"{tag}_rhs"][0])
        env.close()


def test_single_env_overflow_raises(api):
    Env, _ = api
    env = Env(L=22.0, N=256)
    env.u = np.random.RandomState(0).uniform(-0.4, 0.4, 256)
    with pytest.raises(FloatingPointError):
        env.step([[0.0, 0.0, 0.0, 0.0]])


def test_vec_env_matches_single_envs_and_autoresets(api, ks_golden):
    Env, make_vec = api
    E = 6
    cfg = {"Tmax": 0.5, "cfg_steps": 250}  # 2 steps per episode
    vec = make_vec(E, config=cfg, step_mode="exact", reset_mode="exact")
    assert vec.max_episode_steps == 2
    obs = vec.reset(seed=40)
    assert obs.shape == (E, 1, 64) and obs.dtype == np.float32
    singles = [Env(step_mode="exact", reset_mode="exact", **cfg) for _ in range(E)]
    for i, s in enumerate(singles):
        o = s.reset(seed=40 + i)
        np.testing.assert_array_equal(obs[i, 0], o[0].astype(np.float32))
    # the golden seeded reset, through the batched path
    vec1 = make_vec(1, reset_mode="exact")
    np.testing.assert_array_equal(vec1.reset(seed=int(ks_golden["n64_reset_seed"]))[0, 0],
                                  ks_golden["n64_reset_u"].astype(np.float32))
    rs = np.random.RandomState(0)
    a = rs.uniform(-1, 1, (E, 1, 4)).astype(np.float32)
    o, r, term, trunc, infos = vec.step(a)
    ref = [s.step(x) for s, x in zip(singles, a)]
    np.testing.assert_array_equal(o[:, 0], np.stack([x[0][0] for x in ref]).astype(np.float32))
    np.testing.assert_allclose(r, [x[1] for x in ref], rtol=1e-13)
    assert not trunc.any() and "final_observation" not in infos
    o, r, term, trunc, infos = vec.step(a)
    ref = [s.step(x) for s, x in zip(singles, a)]
    assert trunc.all() and infos["_final_observation"].all()
    np.testing.assert_array_equal(np.stack(list(infos["final_observation"]))[:, 0],
                                  np.stack([x[0][0] for x in ref]).astype(np.float32))
    np.testing.assert_array_equal(infos["step"], np.full(E, 2))
    np.testing.assert_array_equal(vec.timestep, np.zeros(E))
    assert np.isfinite(o).all() and np.abs(o).max() > 0.4  # burnt-in fresh states
    vec.close()


def test_vec_env_large_batch_runs(api):
    _, make_vec = api
    vec = make_vec(1024, burn_in=False)
    vec.reset(seed=0)
    o, r, term, trunc, infos = vec.step(np.random.RandomState(1).uniform(-1, 1, (1024, 1, 4)).astype(np.float32))
    assert o.shape == (1024, 1, 64) and np.isfinite(o).all() and (r < 0).all()
    assert vec.stepper.layout()["variant"] in ("wave64_dpp", "row16_dpp")


def test_vec_env_device_resident_step(api):
    import torch
    _, make_vec = api
    E = 64
    a_host = np.random.RandomState(2).uniform(-1, 1, (E, 1, 4)).astype(np.float32)
    ref = make_vec(E, burn_in=False)
    ref.reset(seed=7)
    o_ref, r_ref, _, _, _ = ref.step(a_host)
    dev_env = make_vec(E, burn_in=False)
    dev_env.reset(seed=7)
    obs, rew, trunc, status = dev_env.step_torch(torch.from_numpy(a_host).cuda())
    torch.cuda.synchronize()
    assert obs.is_cuda and obs.shape == (E, 1, 64) and rew.dtype == torch.float64
    np.testing.assert_array_equal(obs.cpu().numpy(), o_ref)
    np.testing.assert_allclose(rew.cpu().numpy(), r_ref, rtol=1e-14)
    assert not trunc.any() and int(status.sum()) == 0


def test_gym_make_and_vector_make_reach_the_hip_stepper(api, ks_golden, monkeypatch):
    """The calls the reference's controller makes (pdecontrol/mbrl/mbrl.py:78-86): ``gym.make(ENV_ID, new_step_api=True)`` ->
    this repo's ``make`` -> env on the HIP stepper inside TimeLimit; ``gym.vector.make(ENV_ID, num_envs=E)`` -> the batched
    env once ``install_batched_vector_make()`` (or PDEGYM_BATCHED=1) routed it.  gym itself is absent from the image: the
    registry / TimeLimit are the in-repo shim's (their gym 0.25.2 semantics are unpinned); the arithmetic is pinned --
    a seeded reset through ``gym.make`` equals the reference's reset bit for bit."""
    import pdegym
    from pdegym._gym import gym
    from pdegym.kuramoto import ENV_ID, KSBatchedVecEnv
    env = gym.make(ENV_ID, new_step_api=True, config={"step_mode": "exact", "reset_mode": "exact"})
    obs = env.reset(seed=int(ks_golden["n64_reset_seed"]))
    np.testing.assert_array_equal(np.asarray(obs)[0], ks_golden["n64_reset_u"])
    inner = env.unwrapped
    inner.u = ks_golden["n64_traj_u0"][0]
    obs, rew, term, trunc, info = env.step(ks_golden["n64_actions"][0])
    np.testing.assert_array_equal(obs[0], ks_golden["n64_traj_u250"][0])
    assert not term and not trunc and info["step"] == 1
    # the time limit: max_episode_steps = ceil(Tmax / (dt * cfg_steps))
    short = gym.make(ENV_ID, new_step_api=True, config={"Tmax": 0.5})
    short.reset(seed=3)
    flags = [short.step(np.zeros((1, 4), np.float32))[3] for _ in range(2)]
    assert flags == [False, True]
    pdegym.install_batched_vector_make(device=0)
    vec = gym.vector.make(ENV_ID, num_envs=5, config={"Tmax": 0.5}, burn_in=False)
    assert isinstance(vec, KSBatchedVecEnv) and vec.num_envs == 5 and vec.reset(seed=1).shape == (5, 1, 64)
    vec.close()
