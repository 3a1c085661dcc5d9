#!/usr/bin/env python3
"""Times every kernel variant / block size on the BASELINE configs (GPU box only)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "model-based-pde-control_amd"))
import kspde  # noqa: E402

CASES = [(1024, 64, 22.0), (4096, 256, 88.0), (4096, 64, 22.0), (16384, 64, 22.0), (32768, 256, 88.0)]
VARIANTS = ["row16_dpp", "row16_bperm", "wave64_dpp", "wave64_bperm", "wave64_hybrid", "wave64_hybrid1", "half32_bperm", "lds"]


def run(E, N, L, variant, block, mode, nsub=250, reps=5):
    s = kspde.KSStepper(E, N, L, mode=mode, variant=variant)
    s.set_block_size(block)
    rs = np.random.RandomState(0)
    s.set_state(rs.uniform(-0.4, 0.4, (E, N)))
    phi = rs.uniform(-0.3, 0.3, (E, N)).astype(np.float32)
    s.step(phi, 1000, want_obs=False)  # warm-up onto the attractor
    import torch
    d_phi = torch.from_numpy(phi).cuda()
    d_obs = torch.empty((E, N), dtype=torch.float32, device="cuda")
    d_ssq = torch.empty(E, dtype=torch.float64, device="cuda")
    d_st = torch.empty(E, dtype=torch.int32, device="cuda")
    args = dict(d_phi=d_phi.data_ptr(), n_substeps=nsub, d_obs=d_obs.data_ptr(), d_ssq=d_ssq.data_ptr(),
                d_status=d_st.data_ptr())
    s.step_device(**args)
    s.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        s.step_device(**args)
    s.sync()
    dt = (time.perf_counter() - t0) / reps
    lay = s.layout()
    rate = E * nsub / dt
    assert int(d_st.sum()) == 0
    return rate, dt, lay


if __name__ == "__main__":
    modes = [a for a in sys.argv[1:] if a in ("fast", "exact")] or ["fast", "exact"]
    skip_lds = "nolds" in sys.argv[1:]
    for (E, N, L) in CASES:
        for mode in modes:
            for v in VARIANTS:
                for block in (64, 256):
                    if v == "lds" and (skip_lds or (block == 64 and N > 64)):
                        continue
                    try:
                        rate, dt, lay = run(E, N, L, v, block, mode)
                    except kspde.KSError:
                        continue
                    print(f"E={E:6d} N={N:4d} {mode:5s} {v:13s} block={lay['block']:3d} grid={lay['grid']:6d} "
                          f"P={lay['points_per_lane']:2d}: {rate:.3e} sub-steps/s  step={dt*1e3:8.3f} ms  "
                          f"HBM-model frac={rate*20*N/8e12:.3f}", flush=True)
