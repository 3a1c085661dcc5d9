"""pdegym.common.vec_wrappers against arrays recorded from the REFERENCE's wrappers driven through the
same scripted fake vector env (oracle/gen_golden.py::wrapper_fixtures -> tests/golden/wrappers_golden.npz)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _fake_vec_env as fk  # noqa: E402

from pdegym._gym import gym  # noqa: E402
from pdegym.common import transforms as T  # noqa: E402
from pdegym.common import vec_wrappers as W  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wrappers_golden.npz")


def build_stack(env):
    ostore = W.StoreNObsVecWrapper(env, num_steps=2)
    oscaling = T.ScaleTransform(batched=True, aggregate=True, frozen=False)
    e = W.TransformObsWrapper(ostore, oscaling, frozen=False)
    e = W.TransformObsWrapper(e, T.BatchTransform(T.SensorTransform(stride=1)))
    e = W.TransformObsWrapper(e, T.BatchTransform(T.SensorTransform(stride=2)))
    astore = W.StoreNActionsVecWrapper(e, num_steps=2)
    low = env.single_action_space.low[np.newaxis, ...] * 2.0
    high = env.single_action_space.high[np.newaxis, ...] * 2.0
    ascaling = T.ScaleTransform(bounds=(low, high), aggregate=True, frozen=True, batched=True).Inverse
    top = W.TransformActionWrapper(astore, ascaling, frozen=True)
    return top, ostore, astore, oscaling


def test_wrapper_stack_matches_reference_bitwise():
    g = np.load(GOLDEN)
    env = fk.make_fake_vec_env(gym)
    top, ostore, astore, oscaling = build_stack(env)
    obs, info = top.reset(return_info=True)
    np.testing.assert_array_equal(obs, g["reset_obs"])
    np.testing.assert_array_equal(info["step"], g["reset_step"])
    np.testing.assert_array_equal(np.asarray(top.observation_space.shape), g["obs_space_shape"])
    np.testing.assert_array_equal(top.action_space.low, g["act_low"])
    np.testing.assert_array_equal(top.action_space.high, g["act_high"])
    saw_final = False
    for k, a in enumerate(fk.scripted_actions(3, 7)):
        top.step_async(a)
        obs, rew, term, trunc, infos = top.step_wait()
        for name, got in (("obs", obs), ("rew", rew), ("trunc", trunc), ("step", infos["step"]),
                          ("ostore_obs", ostore.obs), ("ostore_mask", ostore.mask), ("ostore_finals", ostore.finals),
                          ("astore_actions", astore.actions), ("astore_mask", astore.mask),
                          ("vmin", np.asarray(oscaling.vmin)), ("vmax", np.asarray(oscaling.vmax))):
            np.testing.assert_array_equal(np.asarray(got), g[f"s{k}_{name}"], err_msg=f"step {k}: {name}")
        assert ("final_observation" in infos) == bool(g[f"s{k}_has_final"])
        if "final_observation" in infos:
            saw_final = True
            np.testing.assert_array_equal(np.asarray(list(infos["final_observation"]), dtype=np.float32), g[f"s{k}_final"])
    assert saw_final and ostore.mask.dtype == np.bool_


def test_wrappers_on_batched_ks_env_shapes():
    # the same stack on the HBM-resident batched env (oracle-backed stepper double, no burn-in)
    from _oracle_stepper import OracleStepper
    from pdegym.kuramoto import make_vec
    vec = make_vec(4, config={"Tmax": 0.02, "cfg_steps": 10}, burn_in=False, _stepper_cls=OracleStepper)
    top, ostore, astore, oscaling = build_stack(vec)
    obs = top.reset()
    assert obs.shape == (4, 1, 32) and obs.dtype == np.float32   # sensor stride 2 halves the width
    assert ostore.obs.shape == (4, 2, 1, 64) and astore.actions.shape == (4, 2, 1, 4)
    for t in range(3):
        o, r, term, trunc, infos = top.step(np.full((4, 1, 4), 0.25, np.float32))
        assert o.shape == (4, 1, 32) and np.isfinite(o).all()
    assert np.abs(o).max() <= 1.0 + 1e-6       # running min/max scaling keeps observations in [-1, 1]
    assert astore.mask[:, -1].all()
    # env-side actions are the agent's actions scaled by 2 (ScaleTransform(bounds=+-2).Inverse)
    np.testing.assert_allclose(vec.stepper.u.shape, (4, 64))


@pytest.mark.gpu
def test_six_wrapper_stack_on_batched_hip_env_against_reference_stack():
    """The controller's six-wrapper stack on the HBM-resident KSBatchedVecEnv (exact arithmetic mode, seeded resets with
    the full GPU burn-in) against arrays recorded from the REFERENCE's wrappers stacked on three instances of the
    REFERENCE's KuramotoSivashinskyEnv (oracle/gen_golden.py::wrapper_ks_fixtures).  Observations, stored histories
    and the running min / max of the observation scaling are bit-exact (the state is bit-identical and everything
    after it is fp32 elementwise arithmetic); rewards to 1e-12 (the reference sums its per-sub-step terms in another
    order)."""
    from pdegym.kuramoto import make_vec
    g = np.load(os.path.join(os.path.dirname(GOLDEN), "wrappers_ks_golden.npz"))
    vec = make_vec(3, step_mode="exact", reset_mode="exact")
    top, ostore, astore, oscaling = build_stack(vec)
    obs, info = top.reset(return_info=True, seed=int(g["seed"]))
    np.testing.assert_array_equal(obs, g["reset_obs"])
    np.testing.assert_array_equal(info["step"], g["reset_step"])
    np.testing.assert_array_equal(np.asarray(top.observation_space.shape), g["obs_space_shape"])
    for k, a in enumerate(g["actions"]):
        top.step_async(a)
        obs, rew, term, trunc, infos = top.step_wait()
        for name, got in (("obs", obs), ("trunc", trunc), ("step", infos["step"]),
                          ("ostore_obs", ostore.obs), ("ostore_mask", ostore.mask), ("ostore_finals", ostore.finals),
                          ("astore_actions", astore.actions), ("astore_mask", astore.mask),
                          ("vmin", np.asarray(oscaling.vmin)), ("vmax", np.asarray(oscaling.vmax))):
            np.testing.assert_array_equal(np.asarray(got), g[f"s{k}_{name}"], err_msg=f"step {k}: {name}")
        np.testing.assert_allclose(rew, g[f"s{k}_rew"], rtol=1e-12, err_msg=f"step {k}: reward")
        assert "final_observation" not in infos
