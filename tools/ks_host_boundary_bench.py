#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry points (never bench.py's `value`): ks_step_actions with
numpy in / numpy out, one synchronisation per call.

``sharded``: the vector-env API instead -- ``KSBatchedVecEnv.step`` on one handle against ``KSShardedVecEnv.step`` with the
same envs on TWO (and four) handles of the SAME GPU: what the split entry (ks_step_begin on every shard, ks_step_end on
one host thread per shard) costs or gains on the host side.  One GPU: not a scaling number."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "model-based-pde-control_amd"))
import kspde  # noqa: E402
if os.environ.get("KSPDE_LIB"):      # A/B of library builds
    kspde.LIB_PATH = os.path.abspath(os.environ["KSPDE_LIB"])
from bench import forcing_matrix  # noqa: E402

out = {}
if "sharded" in sys.argv[1:]:
    from pdegym.kuramoto import make_vec
    for name, (E, N, L) in {"c2": (1024, 64, 22.0), "c3": (4096, 256, 88.0)}.items():
        acts = np.random.RandomState(0).uniform(-1, 1, (30, E, 1, 4)).astype(np.float32)
        res = {}
        for label, kw in (("one_handle", dict(device=0)), ("two_handles_one_gpu", dict(devices=[0, 0])),
                          ("four_handles_one_gpu", dict(devices=[0, 0, 0, 0]))):
            env = make_vec(E, config=dict(L=L, N=N), burn_in=False, **kw)
            env.reset(seed=0)
            for i in range(5):
                env.step(acts[i])
            t0 = time.perf_counter()
            for i in range(5, 30):
                env.step(acts[i])
            res[label] = {"ms_per_env_step": (time.perf_counter() - t0) / 25 * 1e3}
            env.close()
        out[name] = res
    print(json.dumps(out))
    sys.exit(0)
for name, (E, N, L) in {"c2": (1024, 64, 22.0), "c3": (4096, 256, 88.0)}.items():
    s = kspde.KSStepper(E, N, L)
    s.set_forcing(forcing_matrix(L, N))
    rs = np.random.RandomState(0)
    s.set_state(rs.uniform(-0.4, 0.4, (E, N)))
    s.step(None, 1000, want_obs=False)
    acts = rs.uniform(-1, 1, (30, E, 4)).astype(np.float32)
    for i in range(5):
        s.step_actions(acts[i], 250)
    t0 = time.perf_counter()
    for i in range(5, 30):
        obs, ssq, st = s.step_actions(acts[i], 250)
    dt = (time.perf_counter() - t0) / 25
    out[name] = {"envs": E, "N": N, "ms_per_step_host_boundary": dt * 1e3, "substeps_per_s": E * 250 / dt}
print(json.dumps(out))
