"""Surrogate TBPTT seqs/s: KSAutoRegConvolutionalLSTM, B=64, T=20, tau=5, tbtt=10,
N=64, Adam lr 1e-3, undscaling = Normalize(mean 0.01, var 0.5); synthetic U(-1,1) inputs."""
import time

import torch

from pdecontrol.architectures import KSAutoRegConvolutionalLSTM
from pdecontrol.surrogates.training import PDETrainingModule
from pdegym.common.transforms import BatchTransform, Normalize


def build_module(device, seed=0, N=64):
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
                               undscaling=undscaling, tau=5, tbtt=10)
    return module.to(device)


def synthetic_batch(B=64, T=20, N=64, device="cpu"):
    g = torch.Generator().manual_seed(1)
    s = torch.rand(B, T, 1, N, generator=g) * 2 - 1
    a = torch.rand(B, T, 1, N, generator=g) * 2 - 1
    return s.to(device), a.to(device)


def time_eager(module, batch, steps, warmup):
    opt = module.configure_optimizers()[0][0]
    dev = batch[0].device

    def one():
        opt.zero_grad(set_to_none=True)
        out = module.training_step(batch, 0)
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


def first_loss(device, B):
    """Loss of the very first training_step of a freshly seeded module (no update applied)."""
    module = build_module(device)
    with torch.no_grad():
        return float(module.training_step(synthetic_batch(B=B, device=device), 0)["loss"])


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
            "loss_after_training": float(graphed.result["loss"].detach())}


def time_ensemble(device, members, steps, warmup, B=64):
    """``members`` independently seeded surrogates (the reference's ensemble, mbrl.py:109) stepping side by
    side on one GPU, each on its own batch of B sequences."""
    from pdecontrol.surrogates.ensemble_step import EnsembleTBPTTStep
    modules = [build_module(device, seed=i) for i in range(members)]
    batches = []
    for i in range(members):
        g = torch.Generator().manual_seed(50 + i)
        batches.append(((torch.rand(B, 20, 1, 64, generator=g) * 2 - 1).to(device),
                        (torch.rand(B, 20, 1, 64, generator=g) * 2 - 1).to(device)))
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
            "losses": [float(r["loss"].detach()) for r in ens.step()]}


def run(device, steps=50, warmup=5, B=64, cpu_steps=3):
    batch = synthetic_batch(B=B, device=device)
    res = {"unit": "seqs/s", "config": {"factory": "KSAutoRegConvolutionalLSTM", "B": B, "T": 20, "tau": 5,
                                        "tbtt": 10, "N": 64, "dtype": "f32", "optimizer": "Adam lr 1e-3",
                                        "step": "training_step + backward + Adam"}}
    module = build_module(device)
    dt, _ = time_eager(module, batch, steps=10, warmup=2)
    res["eager"] = {"value": B / dt, "ms_per_step": dt * 1e3}

    res["hip_graph"] = time_graphed(device, batch, steps, warmup)

    # fused HIP kernels (libsurrogate_hip.so): 2 encoder launches + 1 launch per time step, each way
    from pdecontrol.surrogates import ops
    try:
        ops.enable_fused(True)
        res["hip_graph_fused"] = time_graphed(device, batch, steps * 4, warmup)
        loss_fused = first_loss(device, B)
        # BASELINE configs[2] names 256 grid points: the same factory resized (N = 256), same B / T / tau / tbtt
        try:
            res["n256"] = time_graphed(device, synthetic_batch(B=B, N=256, device=device), steps, warmup)
            res["n256"]["note"] = ("KSAutoRegConvolutionalLSTMN(N=256); the chunk backward has LDS room for one copy of a step's "
                                   "intermediates only, so its DMA is waited for at the top of each step")
        except Exception as exc:
            res["n256"] = {"error": f"{type(exc).__name__}: {exc}"}
        # the reference's default ensemble (3 members, script.py:60) stepped side by side in one graph
        res["ensemble"] = time_ensemble(device, 3, steps * 2, warmup, B)
    finally:
        ops.enable_fused(False)
    res["value"] = res["hip_graph_fused"]["value"]
    # the step is bound by launch count / sequential depth, not by bytes -- report both.
    # HBM model per step (N = 64, B = 64, T = 20): saved forward intermediates 3 840 floats per (step, sample) written
    # once and read once, hidden / cell states and deltas / outputs written once and read once, the batch read twice.
    t_len, n = 20, 64
    # + the backward workspace (gate gradients 4 x 256, dh 256 per step and sample, written and read once)
    bytes_model = 4 * B * t_len * (2 * 3840 + 2 * 2 * 256 + 2 * 2 * n + 2 * 2 * n + 2 * 5 * 256)
    ms = res["hip_graph_fused"]["ms_per_step"]
    res["roofline"] = {"bound": "latency (dependent layer phases, one workgroup per sequence)", "kernels_per_step": 21,
                       "reference_torch_ops_per_step": "~4 000", "hbm_model_bytes_per_step": bytes_model,
                       "hbm_model_gbs": bytes_model / (ms * 1e-3) / 1e9, "hbm_frac_of_8TBs": bytes_model / (ms * 1e-3) / 8e12,
                       "critical_path": "enc_fwd -> (cell chain -> decoders -> integrate) x2 with one encoder in between -> loss -> "
                                        "decoder bwd -> cell chain bwd -> cell wgrad -> enc_bwd x2 -> flush -> Adam "
                                        "(profiles/r01_tbptt_fused_v6_timeline.txt)"}

    # parity of the measured configuration: first-step loss GPU vs CPU (contract: 1e-5 relative)
    loss_gpu, loss_cpu = first_loss(device, B), first_loss("cpu", B)
    res["first_loss"] = {"gpu": loss_gpu, "gpu_fused": loss_fused, "cpu": loss_cpu,
                         "rel_diff": abs(loss_gpu - loss_cpu) / abs(loss_cpu),
                         "rel_diff_fused": abs(loss_fused - loss_cpu) / abs(loss_cpu)}

    # CPU baseline: the same nn.Module tree / training_step / Adam on torch CPU kernels (what the
    # reference runs), at 1 thread (the reference's regime for these tiny ops) and at 16 threads
    best = None
    for nthreads in (1, min(16, torch.get_num_threads())):
        torch.set_num_threads(nthreads)
        dt_cpu, _ = time_eager(build_module("cpu"), synthetic_batch(B=B), steps=cpu_steps, warmup=1)
        if best is None or dt_cpu < best[0]:
            best = (dt_cpu, nthreads)
    res["cpu_baseline"] = {"value": B / best[0], "unit": "seqs/s", "cores": best[1], "kind": "port",
                           "sample": f"same nn.Module tree + training_step + Adam on torch CPU kernels, "
                                     f"{cpu_steps} steps, best of 1 / 16 threads"}
    return res


def run_ddp(device, steps=100, warmup=5, B=64):
    """Data-parallel TBPTT (weak scaling: B sequences per rank): fused fwd/bwd graph -> ONE all-reduce of the
    flat 38 956-byte gradient bucket (RCCL over xGMI) -> Adam graph.  Call on every rank of an initialised
    process group; returns this rank's seconds per step."""
    import torch.distributed as dist
    from pdecontrol.surrogates import ops
    from pdecontrol.surrogates.distributed import broadcast_parameters
    from pdecontrol.surrogates.graph_step import GraphedTBPTTStep
    rank = dist.get_rank()
    g = torch.Generator().manual_seed(1000 + rank)
    s = (torch.rand(B, 20, 1, 64, generator=g) * 2 - 1).to(device)
    a = (torch.rand(B, 20, 1, 64, generator=g) * 2 - 1).to(device)
    try:
        ops.enable_fused(True)
        module = build_module(device)
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
    finally:
        ops.enable_fused(False)
    return dt, in_sync, float(graphed.result["loss"].detach())
