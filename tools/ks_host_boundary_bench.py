#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry points (never bench.py's `value`): ks_step_actions with
numpy in / numpy out, one synchronisation per call."""
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
