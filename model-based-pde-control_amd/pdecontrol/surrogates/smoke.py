"""Tiny surrogate forward + backward on cuda:0 for __graft_entry__.smoke()."""
import torch


def run():
    from pdecontrol.surrogates.bench_tbptt import build_module, synthetic_batch
    dev = torch.device("cuda", 0)
    cpu_module, gpu_module = build_module("cpu"), build_module(dev)
    s, a = synthetic_batch(B=4)
    ref = cpu_module.training_step((s, a), 0)
    ref["loss"].backward()
    out = gpu_module.training_step((s.to(dev), a.to(dev)), 0)
    out["loss"].backward()
    rel = abs(float(out["loss"]) - float(ref["loss"])) / abs(float(ref["loss"]))
    assert rel < 1e-5, rel
    gmax = max(float((pg.grad.cpu() - pc.grad).abs().max()) for pg, pc in
               zip(gpu_module.surrogate.parameters(), cpu_module.surrogate.parameters()) if pc.grad is not None)
    print(f"smoke ok: surrogate TBPTT loss rel diff GPU vs CPU {rel:.2e}, max grad diff {gmax:.2e}")
