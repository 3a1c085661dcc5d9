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
out = {}
if "sharded" in sys.argv[1:]:
    # one fresh process per configuration (a real run builds ONE vector env; a process that has built and closed other
    # handle sets before measures the runtime's stream / queue history: four handles at C2 read 2-3 ms that way, 0.44 ms fresh).
    # This parent never touches the GPU.
    import re
    import subprocess
    for name, (E, N, L) in {"c2": (1024, 64, 22.0), "c3": (4096, 256, 88.0)}.items():
        res = {}
        for label, handles in (("one_handle", 1), ("two_handles_one_gpu", 2), ("four_handles_one_gpu", 4)):
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sharded_trace_run.py"), str(E), str(N), str(L), str(handles)],
                               capture_output=True, text=True, timeout=600)
            m = re.search(r"ms per step ([0-9.]+)\s+\(step_async ([0-9.]+), step_wait ([0-9.]+)\)", r.stdout)
            res[label] = ({"ms_per_env_step": float(m.group(1)), "step_async_ms": float(m.group(2)), "step_wait_ms": float(m.group(3))}
                          if m else {"error": (r.stderr or r.stdout)[-300:]})
        out[name] = res
    print(json.dumps(out))
    sys.exit(0)
import kspde  # noqa: E402
if os.environ.get("KSPDE_LIB"):      # A/B of library builds
    kspde.LIB_PATH = os.path.abspath(os.environ["KSPDE_LIB"])
from bench import forcing_matrix  # noqa: E402

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
