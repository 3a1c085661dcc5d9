"""f2: WorldVecEnv (surrogate-backed imagined rollouts) against arrays recorded from the reference's
WorldVecEnv in the same scripted scenario (oracle/gen_golden.py::world_fixtures)."""
import os
import sys
import types

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _world_scenario as sc  # noqa: E402
from _oracle_stepper import OracleStepper  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "world_golden.npz")


def namespace():
    import pdegym  # noqa: F401
    from pdecontrol.architectures import KSAutoRegConvolutionalLSTM
    from pdecontrol.mbrl.replay import ExperienceReplay
    from pdecontrol.mbrl.types import Sample
    from pdecontrol.mbrl.world.world import WorldVecEnv
    from pdecontrol.surrogates.common import dataset as ds
    from pdecontrol.surrogates.surrogate import PDEEnsemble
    from pdecontrol.surrogates.training import PDETrainingModule
    from pdegym.common import transforms as T
    from pdegym.kuramoto import KuramotoSivashinskyEnv
    env_cls = lambda: KuramotoSivashinskyEnv(_stepper_cls=OracleStepper)  # never stepped: forcing / reward only
    return types.SimpleNamespace(Env=env_cls, T=T, Replay=ExperienceReplay, ds=ds, Sample=Sample,
                                 factory_cls=KSAutoRegConvolutionalLSTM, TrainingModule=PDETrainingModule,
                                 Ensemble=PDEEnsemble, WorldVecEnv=WorldVecEnv)


def test_world_env_matches_reference_on_cpu():
    g = np.load(GOLDEN)
    rec = sc.run(namespace(), device="cpu")
    assert sorted(rec) == sorted(g.files)
    for k in g.files:
        # same torch CPU kernels, same op order: bit-identical
        np.testing.assert_array_equal(np.asarray(rec[k]), g[k], err_msg=k)


@pytest.mark.gpu
def test_world_env_on_gpu_within_tolerance():
    """Plain PyTorch-ROCm kernels (explicit opt-out of the fused path)."""
    from pdecontrol.surrogates import ops
    g = np.load(GOLDEN)
    with ops.fused(False):
        rec = sc.run(namespace(), device=torch.device("cuda", 0))
    for k in g.files:
        a, b = np.asarray(rec[k]), g[k]
        if a.dtype.kind == "f":
            np.testing.assert_allclose(a, b, rtol=2e-3, atol=2e-4, err_msg=k)
        else:
            np.testing.assert_array_equal(a, b, err_msg=k)


@pytest.mark.gpu
def test_world_env_fused_kernels_and_batched_reward():
    from pdecontrol.surrogates import ops
    g = np.load(GOLDEN)
    try:
        ops.enable_fused(True)
        rec = sc.run(namespace(), device=torch.device("cuda", 0))
    finally:
        ops.reset_fused()
    for k in g.files:
        a, b = np.asarray(rec[k]), g[k]
        if a.dtype.kind == "f":
            np.testing.assert_allclose(a, b, rtol=2e-3, atol=2e-4, err_msg=k)
        else:
            np.testing.assert_array_equal(a, b, err_msg=k)
