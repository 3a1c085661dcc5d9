"""Surrogate TBPTT seqs/s: KSAutoRegConvolutionalLSTM(N), B=64, T=20, tau=5, tbtt=10, Adam lr 1e-3,
undscaling = Normalize(mean 0.01, var 0.5); synthetic U(-1,1) inputs.  Primary size N = 256 (BASELINE configs[2]: the
resized factory, pinned by tests/golden/surrogate_n256_golden.npz), N = 64 (the reference's own factory) secondary.

Legs (each = training_step + backward + Adam, B sequences per step):
  hip_graph_fused   the captured step (forward + backward + Adam inside the gradient-reduction launch): ``value``
  eager_fused       what pl.Trainer.fit drives when nothing is configured: eager ``training_step`` on the fused HIP
                    kernels -> zero_grad -> backward -> torch.optim.Adam(fused=True).step(), pytorch-lightning's order
  lightning_graphed the module under (shim) Trainer.fit with graphed=True: Lightning's loop reaches the captured step
  eager_plain / hip_graph_plain   the explicit opt-out: PyTorch-ROCm / MIOpen kernels, eager and under a hipGraph
  cpu_baseline      same module tree on torch CPU kernels (what the reference runs), >= 10 steps
"""
import time

import torch

from pdecontrol.architectures import KSAutoRegConvolutionalLSTM
from pdecontrol.surrogates.training import PDETrainingModule
from pdegym.common.transforms import BatchTransform, Normalize


def build_module(device, seed=0, N=64, **module_kwargs):
    torch.manual_seed(seed)
    norm = Normalize(aggregate=True, batched=True)
    norm.mean, norm.var, norm.count = torch.full((1, 1, 1), 0.01), torch.full((1, 1, 1), 0.5), 100
    undscaling = BatchTransform(norm)
    if N == 64:
        factory = KSAutoRegConvolutionalLSTM()
        model = factory.model()
    else:   # same architecture resized to N grid points (BASELINE configs[2]: 256)
        from pdecontrol.architectures import KSAutoRegConvolutionalLSTMN
        factory = KSAutoRegConvolutionalLSTMN()
        model = factory.model(N=N)
    surrogate = factory.surrogate(delta=0.25, dscaling=undscaling.Inverse, tau=5, **model)
    module = PDETrainingModule(surrogate=surrogate, loss=torch.nn.MSELoss(reduction="none"), tstep=0.25, delta=0.25,
                               undscaling=undscaling, tau=5, tbtt=10, **module_kwargs)
    return module.to(device)


def synthetic_batch(B=64, T=20, N=64, device="cpu"):
    g = torch.Generator().manual_seed(1)
    s = torch.rand(B, T, 1, N, generator=g) * 2 - 1
    a = torch.rand(B, T, 1, N, generator=g) * 2 - 1
    return s.to(device), a.to(device)


def time_eager(module, batch, steps, warmup):
    """pytorch-lightning's automatic-optimization closure order: training_step -> zero_grad -> backward -> step."""
    opt = module.configure_optimizers()[0][0]
    dev = batch[0].device

    def one():
        out = module.training_step(batch, 0)
        opt.zero_grad(set_to_none=True)
        out["loss"].backward()
        opt.step()
        return out
    for _ in range(warmup):
        one()
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = one()
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / steps, float(out["loss"].detach())


def first_loss(device, B, N=64):
    """Loss of the very first training_step of a freshly seeded module (no update applied)."""
    module = build_module(device, N=N)
    with torch.no_grad():
        return float(module.training_step(synthetic_batch(B=B, N=N, device=device), 0)["loss"])


def time_graphed(device, batch, steps, warmup):
    from pdecontrol.surrogates.graph_step import GraphedTBPTTStep
    graphed = GraphedTBPTTStep(build_module(device, N=batch[0].shape[-1]), tuple(batch[0].shape))
    graphed.step(*batch)
    for _ in range(warmup):
        graphed.step()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        graphed.step()
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / steps
    return {"value": batch[0].shape[0] / dt, "ms_per_step": dt * 1e3, "steps": steps,
            "loss_after_training": float(graphed.result["loss"]),
            "schedule_kept": "pipelined" if graphed.used_pipelined else "combined",
            "schedule_autotune_ms": getattr(graphed, "schedule_times_ms", None)}


def time_lightning_graphed(device, batch, steps, warmup):
    """The module as the reference's caller drives it (Trainer.fit over a dataloader), graphed=True."""
    from pdecontrol._compat.lightning import IS_SHIM, pl
    module = build_module(device, N=batch[0].shape[-1], graphed=True)
    pl.Trainer(max_steps=warmup, max_epochs=1).fit(module, train_dataloaders=[batch] * warmup)
    torch.cuda.synchronize(device)
    trainer = pl.Trainer(max_steps=steps, max_epochs=1)
    t0 = time.perf_counter()
    trainer.fit(module, train_dataloaders=[batch] * steps)
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / steps
    return {"value": batch[0].shape[0] / dt, "ms_per_step": dt * 1e3, "steps": steps,
            "trainer": "in-repo Trainer shim (pytorch-lightning is not installed)" if IS_SHIM else "pytorch-lightning"}


def time_ensemble(device, members, steps, warmup, B=64, N=64):
    """``members`` independently seeded surrogates (the reference's ensemble, mbrl.py:109) stepping side by
    side on one GPU, each on its own batch of B sequences."""
    from pdecontrol.surrogates.ensemble_step import EnsembleTBPTTStep
    modules = [build_module(device, seed=i, N=N) for i in range(members)]
    batches = []
    for i in range(members):
        g = torch.Generator().manual_seed(50 + i)
        batches.append(((torch.rand(B, 20, 1, N, generator=g) * 2 - 1).to(device),
                        (torch.rand(B, 20, 1, N, generator=g) * 2 - 1).to(device)))
    ens = EnsembleTBPTTStep(modules, tuple(batches[0][0].shape))
    ens.step(batches)
    for _ in range(warmup):
        ens.step()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        ens.step()
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / steps
    return {"members": members, "value": members * B / dt, "ms_per_step": dt * 1e3, "steps": steps,
            "losses": [float(r["loss"]) for r in ens.step()]}


def cpu_baseline(N=256, B=64, steps=10):
    """Same nn.Module tree / training_step / Adam on torch CPU kernels (what the reference runs), at 1 thread (the
    reference's regime for these tiny ops) and at up to 16 threads; >= 10 timed steps each."""
    batch = synthetic_batch(B=B, N=N)
    prev = torch.get_num_threads()
    best, tried = None, {}
    for nthreads in sorted({1, min(16, prev)}):
        torch.set_num_threads(nthreads)
        dt_cpu, _ = time_eager(build_module("cpu", N=N), batch, steps=steps, warmup=2)
        tried[str(nthreads)] = B / dt_cpu
        if best is None or dt_cpu < best[0]:
            best = (dt_cpu, nthreads)
    torch.set_num_threads(prev)
    return {"value": B / best[0], "unit": "seqs/s", "cores": best[1], "kind": "port", "ms_per_step": best[0] * 1e3,
            "by_threads": tried,
            "sample": f"same nn.Module tree + training_step + Adam on torch CPU kernels, N={N}, B={B}, {steps} steps, "
                      f"best of 1 / {min(16, prev)} threads"}


def _hbm_model_bytes(B, t_len, N):
    """HBM model per step: saved forward intermediates (3 840 floats per (step, sample) at N = 64, scaling with N) written
    once and read once, hidden / cell states and deltas / outputs written once and read once, the batch read twice,
    + the backward workspace (gate gradients 4 x 256, dh 256 per step and sample at N = 64)."""
    s = N // 64
    return 4 * B * t_len * (2 * 3840 * s + 2 * 2 * 256 * s + 2 * 2 * N + 2 * 2 * N + 2 * 5 * 256 * s)


def run_size(device, N, steps, warmup, B=64, with_plain=True, with_ensemble=True):
    from pdecontrol.surrogates import ops
    batch = synthetic_batch(B=B, N=N, device=device)
    res = {"N": N}
    res["hip_graph_fused"] = time_graphed(device, batch, steps * 4, warmup)
    dt, _ = time_eager(build_module(device, N=N), batch, steps=steps * 2, warmup=warmup)
    res["eager_fused"] = {"value": B / dt, "ms_per_step": dt * 1e3, "steps": steps * 2,
                          "what": "training_step -> zero_grad -> backward -> PackAdam.step() in Lightning's closure order: the path "
                                  "pl.Trainer.fit drives by default (reference: pdecontrol/mbrl/mbrl.py:593); forward + loss and "
                                  "backward + gradient reduction are two replayed hipGraphs behind one autograd node "
                                  "(graph_step.GraphedAutogradStep)",
                          "ratio_to_graphed": dt * 1e3 / res["hip_graph_fused"]["ms_per_step"]}
    launch_by_launch = build_module(device, N=N)
    launch_by_launch.split_graphs = False
    dt, _ = time_eager(launch_by_launch, batch, steps=steps, warmup=warmup)
    res["eager_fused_launch_by_launch"] = {"value": B / dt, "ms_per_step": dt * 1e3, "steps": steps,
                                           "what": "the same with PDECONTROL_SPLIT_GRAPHS=0: ~25 ctypes launches + two autograd "
                                                   "nodes per step from Python (host-bound)"}
    try:
        res["lightning_graphed"] = time_lightning_graphed(device, batch, steps * 2, warmup)
    except Exception as exc:
        res["lightning_graphed"] = {"error": f"{type(exc).__name__}: {exc}"}
    if with_ensemble:
        # the reference's default ensemble (3 members, script.py:60) stepped side by side in one graph
        try:
            res["ensemble"] = time_ensemble(device, 3, steps * 2, warmup, B, N)
        except Exception as exc:
            res["ensemble"] = {"error": f"{type(exc).__name__}: {exc}"}
    if with_plain:
        with ops.fused(False):
            dt, _ = time_eager(build_module(device, N=N), batch, steps=5, warmup=2)
            res["eager_plain"] = {"value": B / dt, "ms_per_step": dt * 1e3, "what": "opt-out: PyTorch-ROCm / MIOpen kernels"}
            try:
                res["hip_graph_plain"] = time_graphed(device, batch, max(steps // 2, 5), warmup)
            except Exception as exc:
                res["hip_graph_plain"] = {"error": f"{type(exc).__name__}: {exc}"}
    ms = res["hip_graph_fused"]["ms_per_step"]
    bytes_model = _hbm_model_bytes(B, 20, N)
    res["value"] = res["hip_graph_fused"]["value"]
    res["hip_graph_fused"]["schedule"] = ("pipelined = chunk 0's loss rows + backward on a side branch beside chunk 1's forward "
                                          "(hipops.fused_tbptt_train), combined = every chunk's backward in the same launches; both "
                                          "are captured and timed once per batch shape, the faster one is kept (which hardware "
                                          "queue the graph's second stream gets is the runtime's choice)")
    res["roofline"] = {"bound": "instruction issue of the two overlapped queues + a critical chain of ~17 dependent launches (not HBM, not MFMA: "
                                "profiles/r03_tbptt_n256_sq_counters.txt, DESIGN 4.4)", "kernels_per_step": 30,
                       "reference_torch_ops_per_step": "~4 000", "hbm_model_bytes_per_step": bytes_model,
                       "hbm_model_gbs": bytes_model / (ms * 1e-3) / 1e9, "hbm_frac_of_8TBs": bytes_model / (ms * 1e-3) / 8e12}
    # parity of the measured configuration: first-step loss GPU (fused) vs CPU (contract: 1e-5 relative)
    loss_gpu, loss_cpu = first_loss(device, B, N), first_loss("cpu", B, N)
    res["first_loss"] = {"gpu_fused": loss_gpu, "cpu": loss_cpu, "rel_diff_fused": abs(loss_gpu - loss_cpu) / abs(loss_cpu)}
    return res


def run(device, steps=50, warmup=5, B=64, cpu=None):
    """N = 256 is the headline size of the TBPTT leg (BASELINE configs[2]); N = 64 (the reference's factory) rides along."""
    res = {"unit": "seqs/s", "config": {"factory": "KSAutoRegConvolutionalLSTMN(N=256)", "B": B, "T": 20, "tau": 5,
                                        "tbtt": 10, "N": 256, "dtype": "f32", "optimizer": "Adam lr 1e-3",
                                        "step": "training_step + backward + Adam"}}
    res.update(run_size(device, 256, steps, warmup, B))
    if cpu is not None:
        res["cpu_baseline"] = cpu
    try:
        res["n64"] = run_size(device, 64, steps, warmup, B)
        res["n64"]["factory"] = "KSAutoRegConvolutionalLSTM (the reference's own, N = 64)"
    except Exception as exc:
        res["n64"] = {"error": f"{type(exc).__name__}: {exc}"}
    return res


def run_ddp(device, steps=100, warmup=5, B=64, N=256):
    """Data-parallel TBPTT (weak scaling: B sequences per rank): fused fwd/bwd graph -> ONE all-reduce of the
    flat gradient bucket (RCCL over xGMI) -> Adam graph.  Call on every rank of an initialised
    process group; returns (this rank's seconds per step, ranks in sync, loss, bucket bytes)."""
    import torch.distributed as dist
    from pdecontrol.surrogates.distributed import broadcast_parameters
    from pdecontrol.surrogates.graph_step import GraphedTBPTTStep
    rank = dist.get_rank()
    g = torch.Generator().manual_seed(1000 + rank)
    s = (torch.rand(B, 20, 1, N, generator=g) * 2 - 1).to(device)
    a = (torch.rand(B, 20, 1, N, generator=g) * 2 - 1).to(device)
    module = build_module(device, N=N)
    broadcast_parameters(module.surrogate)
    graphed = GraphedTBPTTStep(module, tuple(s.shape), distributed=True)
    graphed.step(s, a)
    for _ in range(warmup):
        graphed.step()
    torch.cuda.synchronize(device)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        graphed.step()
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / steps
    # every rank must hold identical parameters after identical all-reduced updates
    flat = torch.cat([p.detach().reshape(-1) for p in module.surrogate.parameters()])
    ref = flat.clone()
    dist.broadcast(ref, src=0)
    in_sync = bool(torch.equal(flat, ref))
    return dt, in_sync, float(graphed.result["loss"]), graphed.bucket.nbytes
