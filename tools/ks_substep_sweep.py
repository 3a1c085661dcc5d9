#!/usr/bin/env python3
"""Launch time of the fused stepper against the number of sub-steps per launch: T(n) = a + b n.  ``a`` is what a launch
costs beyond its sub-steps (dispatch, prologue / epilogue, whatever the clocks do when a kernel starts); ``b`` the
steady-state time per sub-step.  Back-to-back launches on one stream, HIP events around the batch (GPU box only).

usage: tools/ks_substep_sweep.py [c2|c3] [fast|exact]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "model-based-pde-control_amd"))
import torch  # noqa: E402
import kspde  # noqa: E402
import bench  # noqa: E402

name = next((a for a in sys.argv[1:] if a in bench.WORKLOADS), "c3")
mode = next((a for a in sys.argv[1:] if a in ("fast", "exact")), "fast")
E, N, L, _ = bench.WORKLOADS[name]
dev = torch.device("cuda", 0)
s = kspde.KSStepper(E, N, L, 1e-3, device=0, mode=mode)
stream = torch.cuda.Stream(device=dev)
s.set_stream(stream.cuda_stream)
s.set_forcing(bench.forcing_matrix(L, N))
s.set_state(np.random.RandomState(0).uniform(-0.4, 0.4, (E, N)))
s.step(None, 1000, want_obs=False)
acts = torch.from_numpy(np.random.RandomState(1).uniform(-1, 1, (E, 4)).astype(np.float32)).to(dev)
obs = torch.empty((E, N), dtype=torch.float32, device=dev)
ssq = torch.empty(E, dtype=torch.float64, device=dev)
st = torch.zeros(E, dtype=torch.int32, device=dev)
args = dict(d_actions=acts.data_ptr(), d_obs=obs.data_ptr(), d_ssq=ssq.data_ptr(), d_status=st.data_ptr())
rows = []
for n in (1, 5, 25, 50, 125, 250, 500, 1000, 2500, 10000):
    reps = max(3, min(200, int(60000 / n)))
    with torch.cuda.stream(stream):
        for _ in range(3):
            s.step_device(n_substeps=n, **args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            s.step_device(n_substeps=n, **args)
        e1.record(stream)
    torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / reps
    rows.append((n, ms))
    print(f"{name} {mode} n_substeps={n:6d}: {ms * 1e3:10.1f} us per launch, {ms * 1e3 / n:8.3f} us per sub-step, "
          f"HBM-model frac {20.0 * N * E * n / (ms * 1e-3) / 8e12:.3f}", flush=True)
assert int(st.sum()) == 0
n_, t_ = np.array([r[0] for r in rows], float), np.array([r[1] for r in rows], float)
b, a = np.polyfit(n_[3:], t_[3:], 1)
print(f"fit over n >= {int(n_[3])}: T(n) = {a * 1e3:.1f} us + {b * 1e3:.4f} us x n   (250 sub-steps: {a * 1e3 + 250 * b * 1e3:.1f} us, "
      f"of which fixed {100 * a / (a + 250 * b):.1f} %)")
