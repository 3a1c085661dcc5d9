import os, sys, time
sys.path.insert(0, "/root/repo/model-based-pde-control_amd")
import torch
from pdecontrol.surrogates.bench_tbptt import build_module, synthetic_batch, time_eager, time_lightning_graphed
dev = torch.device("cuda", 0)
for N in (64, 256):
    batch = synthetic_batch(B=64, N=N, device=dev)
    for split in (True, False, True, False):
        m = build_module(dev, N=N)
        m.split_graphs = split
        dt, loss = time_eager(m, batch, steps=200, warmup=10)
        print(f"N={N} split_graphs={split}: {dt*1e3:.4f} ms/step loss {loss:.5f}", flush=True)
