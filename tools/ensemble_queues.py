import os, sys, time
sys.path.insert(0, "/root/repo/model-based-pde-control_amd")
import torch
from pdecontrol.surrogates.bench_tbptt import time_ensemble
dev = torch.device("cuda", 0)
for N in (64, 256):
    r = time_ensemble(dev, 3, 200, 10, 64, N)
    print("GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"), "N", N, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if k in ("ms_per_step", "value")}, flush=True)
