"""Tiny surrogate TBPTT step on cuda:0 for __graft_entry__.smoke(): the fused HIP kernels (libsurrogate_hip.so, the
default CUDA path) against the same module on the CPU, eagerly and through the captured hipGraph."""
import torch


def run():
    from pdecontrol.surrogates import ops
    from pdecontrol.surrogates.bench_tbptt import build_module, synthetic_batch
    dev = torch.device("cuda", 0)
    cpu_module, gpu_module = build_module("cpu"), build_module(dev)
    s, a = synthetic_batch(B=4)
    ref = cpu_module.training_step((s, a), 0)
    ref["loss"].backward()
    batch = (s.to(dev), a.to(dev))
    assert ops.use_fused(batch[0]), "the fused HIP kernels must be the CUDA path"
    out = gpu_module.training_step(batch, 0)
    out["loss"].backward()
    torch.cuda.synchronize(dev)
    assert getattr(gpu_module.surrogate, "_fused_packs", None) is not None, "fused kernels did not run"
    rel = abs(float(out["loss"].detach()) - float(ref["loss"].detach())) / abs(float(ref["loss"].detach()))
    assert rel < 1e-5, rel
    gmax = max(float((pg.grad.cpu() - pc.grad).abs().max()) for pg, pc in
               zip(gpu_module.surrogate.parameters(), cpu_module.surrogate.parameters()) if pc.grad is not None)
    assert gmax < 1e-3, gmax
    # the same step as one replayed hipGraph (forward + backward + Adam inside the gradient-reduction launch)
    first = float(gpu_module.fused_step(batch)["loss"])
    for _ in range(3):
        last = float(gpu_module.fused_step(batch)["loss"])
    assert abs(first - float(ref["loss"].detach())) / abs(first) < 1e-5 and last < first
    print(f"smoke ok: fused surrogate TBPTT loss rel diff GPU vs CPU {rel:.2e}, max grad diff {gmax:.2e}; "
          f"graphed step trains ({first:.5f} -> {last:.5f})")
