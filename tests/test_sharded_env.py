"""KSShardedVecEnv: one controller process, one handle + stream per device, no collective (replaces
the one-subprocess-per-env AsyncVectorEnv of pdecontrol/mbrl/mbrl.py:81-86).  The sharded env must be indistinguishable
from ``KSBatchedVecEnv`` -- bit for bit in exact arithmetic -- whatever the split.

CPU: shards on the library's CPU twin (``devices=[-1, -1, -1]``).  GPU: two handles on the one GPU (``[0, 0]``)."""
import numpy as np
import pytest


def _episode(vec, n_steps, seed):
    obs = vec.reset(seed=seed)
    outs = [obs.copy()]
    rng = np.random.RandomState(5)
    for _ in range(n_steps):
        a = rng.uniform(-1, 1, (vec.num_envs, 1, 4)).astype(np.float32)
        vec.step_async(a)
        o, r, term, trunc, info = vec.step_wait()
        outs.append((o.copy(), r.copy(), trunc.copy(), info["step"].copy(),
                     [None if f is None else f.copy() for f in info.get("final_observation", [])]))
    return outs


def _same(a, b):
    np.testing.assert_array_equal(a[0], b[0])
    for (o1, r1, t1, s1, f1), (o2, r2, t2, s2, f2) in zip(a[1:], b[1:]):
        np.testing.assert_array_equal(o1, o2)
        np.testing.assert_array_equal(r1, r2)
        np.testing.assert_array_equal(t1, t2)
        np.testing.assert_array_equal(s1, s2)
        assert len(f1) == len(f2)
        for x, y in zip(f1, f2):
            assert (x is None) == (y is None)
            if x is not None:
                np.testing.assert_array_equal(x, y)


def _compare(devices, single_device, E, cfg, n_steps, burn_in):
    from pdegym.kuramoto import make_vec, KSShardedVecEnv
    kw = dict(config=cfg, step_mode="exact", reset_mode="exact", burn_in=burn_in)
    ref = make_vec(E, device=single_device, **kw)
    sh = make_vec(E, devices=devices, **kw)
    assert isinstance(sh, KSShardedVecEnv) and len(sh.shards) == min(len(devices), E)
    assert [hi - lo for lo, hi, _ in sh.shards] == [len(x) for x in np.array_split(np.arange(E), len(sh.shards))]
    _same(_episode(ref, n_steps, 11), _episode(sh, n_steps, 11))
    ref.close()
    sh.close()


def test_sharded_env_on_cpu_twin_matches_batched_env():
    # 7 envs over 3 shards (3 + 2 + 2), episodes of 3 steps: two autoresets inside 7 steps, short burn-in
    cfg = {"Tmax": 0.03, "cfg_steps": 10}
    from pdegym.kuramoto.kuramoto import KuramotoSivashinskyEnv
    old = KuramotoSivashinskyEnv.BURN_IN_TIME
    KuramotoSivashinskyEnv.BURN_IN_TIME = 0.5          # 50 steps x 10 sub-steps of burn-in: the test stays in seconds
    try:
        _compare([-1, -1, -1], -1, 7, cfg, 7, burn_in=True)
        _compare(["cpu"], -1, 3, cfg, 4, burn_in=False)
    finally:
        KuramotoSivashinskyEnv.BURN_IN_TIME = old


def test_vector_make_routes_to_the_sharded_env(monkeypatch):
    import pdegym
    from pdegym._gym import gym
    from pdegym.kuramoto import ENV_ID, KSShardedVecEnv
    monkeypatch.setenv("PDEGYM_DEVICES", "cpu,cpu")
    saved = getattr(gym.vector, "make", None)
    try:
        pdegym.install_batched_vector_make()
        env = gym.vector.make(ENV_ID, num_envs=4, config={"Tmax": 0.02, "cfg_steps": 5}, burn_in=False)
        assert isinstance(env, KSShardedVecEnv) and env.devices == [-1, -1]
        obs = env.reset(seed=0)
        assert obs.shape == (4, 1, 64) and obs.dtype == np.float32
        o, r, term, trunc, info = env.step(np.zeros((4, 1, 4), np.float32))
        assert o.shape == (4, 1, 64) and r.shape == (4,) and not trunc.any()
        env.close()
    finally:
        if saved is None:
            del gym.vector.make
        else:
            gym.vector.make = saved


def test_step_wait_without_async_and_double_async():
    from pdegym.kuramoto import make_vec
    env = make_vec(2, devices=[-1, -1], config={"cfg_steps": 5}, burn_in=False)
    env.reset(seed=1)
    with pytest.raises(AssertionError):
        env.step_wait()
    env.step_async(np.zeros((2, 1, 4), np.float32))
    with pytest.raises(AssertionError):
        env.step_async(np.zeros((2, 1, 4), np.float32))
    env.step_wait()
    env.close()


@pytest.mark.gpu
def test_sharded_env_two_handles_on_one_gpu_matches_batched_env():
    # full-size burn-in (200 000 sub-steps per reset, exact arithmetic) on both; 2-step episodes -> autoresets
    _compare([0, 0], 0, 6, {"Tmax": 0.5, "cfg_steps": 250}, 5, burn_in=True)
    # configs[2]-shaped block split 3 ways, no burn-in: the whole-block fetch path (> 32 rows per shard)
    _compare([0, 0, 0], 0, 600, {"L": 88.0, "N": 256, "Tmax": 0.5}, 3, burn_in=False)
