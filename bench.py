#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native KS hot path (+ surrogate TBPTT step).

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 it is launched
under torch.distributed.run with one rank per GPU.  Rank 0 prints ONE JSON line.

What one "step" is: one ``env.step`` of the whole batch = ONE launch of the fused HIP stepper that
advances every env by cfg_steps = 250 RK4 sub-steps (pdegym/kuramoto/kuramoto.py:78-98 of the
reference), actions resident in HBM, phi = actions @ F evaluated in-kernel, fp32 observation,
reward sum and overflow flags written back to HBM.  Workload at N = 1: BASELINE.json configs[1]
(KS L=22, 64 grid points, 1024 batched envs, fp64).  Weak scaling: every rank owns its own 1024
envs (envs never interact, no data-path collective); ``value`` = env-sub-steps of ALL ranks per
second of the slowest rank.

Extra objects in the JSON line:
  roofline      HBM-model roofline of SURVEY.md 8(d): algorithmic bytes = 20*N per env-sub-step
                (read u fp64, write u fp64, read phi fp32), achieved = bytes per launch / average
                launch duration measured with HIP events on the launch stream.  The kernel keeps
                the state in registers for all 250 sub-steps, so its real HBM traffic is ~250x
                smaller; the binding resource is the fp64 VALU pipe (see "fp64_valu").
  cpu_baseline  the CPU oracle (oracle/ks_oracle.c, bit-exact restatement of the reference
                stepper) timed on this box's host cores on a bounded sample of the same workload.
  tbptt         surrogate TBPTT training step (B=64, T=20, tau=5, tbtt=10), seqs/s, when the
                surrogate package is built.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "model-based-pde-control_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

WORKLOADS = {
    # name: (envs per GPU, N, L)  -- BASELINE.json configs[1] / configs[2] (L=88)
    "c2": (1024, 64, 22.0),
    "c3": (4096, 256, 88.0),
}
CFG_STEPS = 250
DT = 1e-3
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6     # MI355X fp64 vector peak (spec)
FLOPS_PER_POINT_SUBSTEP = 151  # fast-mode kernel: 4 x 34 (rhs) + 14 (RK4 combines) + 1 (reward)


def forcing_matrix(L, N, sigma=0.4, Xi=(0.0, 0.25, 0.5, 0.75)):
    """fp32 [4, N] Gaussian forcing matrix, same torch ops as the reference (transforms.py:253-260)."""
    x = torch.as_tensor(np.linspace(0.0, L - L / N, N, dtype=np.float32))
    xi = (L * torch.as_tensor(Xi, dtype=torch.float32)).reshape(-1, 1)
    F = torch.exp(-((x - xi) ** 2.0) / (2.0 * sigma ** 2))
    return (F / np.sqrt(2.0 * np.pi * sigma)).numpy()


def cpu_baseline(E, N, L, target_seconds=12.0):
    """Time the CPU oracle on a bounded sample of the same workload (all host cores, OpenMP)."""
    from oracle import ks_oracle as ko
    cores = min(os.cpu_count() or 1, 64)
    rs = np.random.RandomState(1234)
    u0 = rs.uniform(-0.4, 0.4, (E, N))
    act = np.random.RandomState(99).uniform(-1, 1, (E, 4)).astype(np.float32)
    phi = ko.phi_from_actions(act, forcing_matrix(L, N))
    ko.step(u0, phi, L / N, DT, 50, nthreads=cores)  # spin up the thread pool, warm caches
    nsub = 500
    t0 = time.perf_counter()
    ko.step(u0, phi, L / N, DT, nsub, nthreads=cores)
    probe = time.perf_counter() - t0
    nsub = int(max(250, min(CFG_STEPS * 400, nsub * target_seconds / max(probe, 1e-6))))
    t0 = time.perf_counter()
    ko.step(u0, phi, L / N, DT, nsub, nthreads=cores)
    dt = time.perf_counter() - t0
    out = {"value": E * nsub / dt, "unit": "sub-steps/s", "cores": cores, "kind": "port",
           "sample": f"oracle/ks_oracle.c (bit-exact C restatement of the reference stepper), {E} envs x {nsub} "
                     f"sub-steps, N={N}, OpenMP over envs, {dt:.1f} s"}
    # SURVEY 8(d) baseline (A): the reference's own call structure (per-env object, Python loop over sub-steps,
    # 16 scipy convolve1d + a torch round trip per sub-step), one env on one core
    try:
        n_py = 1500
        t0 = time.perf_counter()
        ko.step_scipy_structured(u0[0], phi[0].astype(np.float64), L / N, DT, n_py)
        dt_py = time.perf_counter() - t0
        out["reference_call_structure"] = {"value": n_py / dt_py, "unit": "sub-steps/s per core", "cores": 1,
                                           "sample": f"1 env x {n_py} sub-steps through scipy.ndimage.convolve1d + torch.norm, "
                                                     f"{dt_py:.1f} s; the reference runs one such process per env"}
    except Exception as exc:  # scipy / torch CPU missing must not lose the baseline
        out["reference_call_structure"] = {"error": f"{type(exc).__name__}: {exc}"}
    return out


def measured_hbm_copy_gbs(dev, mib=1024, reps=10):
    """Device-to-device copy rate of this box (read + write bytes per second), SURVEY 8(d): the measured companion of
    the nominal 8 TB/s the roofline fraction is quoted against."""
    n = mib * (1 << 20) // 4
    a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    for _ in range(2):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize(dev)
    return 2.0 * a.numel() * 4 * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def measure_secondary(kspde, local_rank, dev, name, mode, steps=20, warmup=3):
    """Same measurement as the headline one for another BASELINE config (single GPU, events on the launch stream)."""
    E, N, L = WORKLOADS[name]
    stepper = kspde.KSStepper(E, N, L, DT, device=local_rank, mode=mode)
    stream = torch.cuda.Stream(device=dev)
    stepper.set_stream(stream.cuda_stream)
    stepper.set_forcing(forcing_matrix(L, N))
    stepper.set_state(np.stack([np.random.RandomState(1234 + e).uniform(-0.4, 0.4, N) for e in range(E)]))
    for _ in range(4):
        stepper.step(None, CFG_STEPS, want_obs=False)
    acts = torch.from_numpy(np.random.RandomState(7).uniform(-1, 1, (steps + warmup, E, 4)).astype(np.float32)).to(dev)
    d_obs = torch.empty((E, N), dtype=torch.float32, device=dev)
    d_ssq = torch.empty(E, dtype=torch.float64, device=dev)
    d_st = torch.zeros(E, dtype=torch.int32, device=dev)
    with torch.cuda.stream(stream):
        ev = []
        for i in range(steps + warmup):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            stepper.step_device(d_actions=acts[i].data_ptr(), n_substeps=CFG_STEPS, d_obs=d_obs.data_ptr(),
                                d_ssq=d_ssq.data_ptr(), d_status=d_st.data_ptr())
            b.record(stream)
            ev.append((a, b))
    torch.cuda.synchronize(dev)
    assert int(d_st.sum()) == 0
    ms = float(np.mean([a.elapsed_time(b) for a, b in ev[warmup:]]))
    gbs = 20.0 * N * E * CFG_STEPS / (ms * 1e-3) / 1e9
    tf = FLOPS_PER_POINT_SUBSTEP * N * E * CFG_STEPS / (ms * 1e-3) / 1e12
    return {"workload": f"KS L={L:g} N={N}, {E} envs (BASELINE.json configs[2])", "value": E * CFG_STEPS / (ms * 1e-3),
            "unit": "sub-steps/s", "avg_launch_ms": ms, "kernel": stepper.layout(),
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                         "fp64_valu_frac": tf / FP64_PEAK_TFLOPS}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=list(WORKLOADS), default="c2")
    ap.add_argument("--mode", choices=["fast", "exact"], default="fast")
    ap.add_argument("--variant", default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tbptt", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the HIP stepper has no CPU fallback")
    # one rank per GPU; ranks beyond the visible GPU count share devices (only used to rehearse the
    # multi-rank code path on a 1-GPU box, with BENCH_DIST_BACKEND=gloo)
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dist = None
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend)
    n_gpus = world if world > 1 else args.gpus
    if world == 1 and args.gpus != 1:
        sys.exit("for --gpus N > 1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py ...")

    import kspde
    E, N, L = WORKLOADS[args.workload]
    K, W = args.steps, args.warmup
    dev = torch.device("cuda", local_rank)

    # ---- synthetic inputs: IC ~ U(-0.4, 0.4) per env seed, short warm-up onto the
    #      attractor, actions ~ U(-1, 1) redrawn every step, all resident in HBM before timing
    stepper = kspde.KSStepper(E, N, L, DT, device=local_rank, mode=args.mode, variant=args.variant)
    stream = torch.cuda.Stream(device=dev)
    stepper.set_stream(stream.cuda_stream)
    stepper.set_forcing(forcing_matrix(L, N))
    base = rank * E
    u0 = np.stack([np.random.RandomState(1234 + base + e).uniform(-0.4, 0.4, N) for e in range(E)])
    stepper.set_state(u0)
    for _ in range(4):  # 1000 sub-steps onto the attractor, as 250-sub-step launches like the timed ones
        stepper.step(None, CFG_STEPS, want_obs=False)
    acts = torch.from_numpy(np.random.RandomState(99 + rank).uniform(-1, 1, (K + W, E, 4)).astype(np.float32)).to(dev)
    d_obs = torch.empty((E, N), dtype=torch.float32, device=dev)
    d_ssq = torch.empty(E, dtype=torch.float64, device=dev)
    d_st = torch.zeros(E, dtype=torch.int32, device=dev)
    st_acc = torch.zeros(E, dtype=torch.int32, device=dev)
    torch.cuda.synchronize(dev)

    def one_step(i):
        stepper.step_device(d_actions=acts[i].data_ptr(), n_substeps=CFG_STEPS, d_obs=d_obs.data_ptr(),
                            d_ssq=d_ssq.data_ptr(), d_status=d_st.data_ptr())

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    with torch.cuda.stream(stream):
        for i in range(W):
            one_step(i)
        barrier()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
        t0 = time.perf_counter()
        for i in range(K):
            ev[i][0].record(stream)
            one_step(W + i)
            ev[i][1].record(stream)
        barrier()
        elapsed = time.perf_counter() - t0
        st_acc |= d_st
    torch.cuda.synchronize(dev)
    assert int(st_acc.sum()) == 0, "non-finite state during the benchmark"
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    lay = stepper.layout()
    traffic = None
    pmc_file = os.path.join(ROOT, "profiles", "ks_pmc_traffic.json")
    if os.path.exists(pmc_file):
        try:
            traffic = json.load(open(pmc_file)).get(args.workload, {}).get("hbm_bytes_per_launch")
        except (OSError, ValueError):
            traffic = None
    valu_issue = None
    try:   # SQ counters of this command (tools/prof_sq.sh): how much of the waves' time the VALU is issuing
        sq = json.load(open(os.path.join(ROOT, "profiles", "ks_sq_counters.json"))).get(args.workload)
        if sq:
            valu_issue = {"frac_of_wave_cycles": sq["fractions_of_wave_cycles"]["SQ_ACTIVE_INST_VALU"],
                          "wait_frac": sq["fractions_of_wave_cycles"]["SQ_WAIT_ANY"],
                          "valu_instructions_per_point_substep": sq["valu_instructions_per_point_substep"],
                          "source": "profiles/ks_sq_counters.json (rocprofv3 --pmc, SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES)"}
    except (OSError, ValueError, KeyError):
        valu_issue = None
    total_substeps = n_gpus * E * CFG_STEPS * K
    value = total_substeps / elapsed
    alg_bytes_per_launch = 20.0 * N * E * CFG_STEPS
    achieved_gbs = alg_bytes_per_launch / (kernel_ms * 1e-3) / 1e9
    flops_per_launch = FLOPS_PER_POINT_SUBSTEP * N * E * CFG_STEPS
    out = {
        "metric": "KS sub-steps/sec (batched env)",
        "value": value,
        "unit": "sub-steps/s",
        "n_gpus": n_gpus,
        "steps": K,
        "warmup": W,
        "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"KS L={L:g} N={N} dt={DT:g} cfg_steps={CFG_STEPS}, {E} batched envs per GPU "
                        f"(BASELINE.json configs[{1 if args.workload == 'c2' else 2}]), random-action rollout",
            "envs_per_gpu": E, "grid_points": N, "mode": args.mode, "kernel": lay,
            "sharding": "envs sharded by rank, no collective" if n_gpus > 1 else "single GPU",
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_note": "HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                            "command (profiles/ks_pmc_traffic.json; FETCH_SIZE doubled per the gfx950 guide)",
            "kernel": "ks_rk4_fused", "avg_launch_ms": kernel_ms,
            "algorithmic_bytes_per_launch": alg_bytes_per_launch,
            "note": "algorithmic 20*N B per env-sub-step (SURVEY 8d); state stays in VGPRs for all 250 "
                    "sub-steps so real HBM traffic is ~1/250 of this; binding resource is fp64 VALU",
            "fp64_valu": {"achieved": flops_per_launch / (kernel_ms * 1e-3) / 1e12, "peak": FP64_PEAK_TFLOPS,
                          "unit": "TFLOP/s",
                          "frac": flops_per_launch / (kernel_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                          "flops_per_point_substep": FLOPS_PER_POINT_SUBSTEP},
            "valu_issue": valu_issue,
        },
    }
    if rank == 0 and n_gpus == 1:
        try:
            out["roofline"]["hbm_copy_measured"] = {"value": measured_hbm_copy_gbs(dev), "unit": "GB/s",
                                                    "what": "1 GiB device-to-device copy, read + write bytes"}
        except Exception as exc:
            out["roofline"]["hbm_copy_measured"] = {"error": f"{type(exc).__name__}: {exc}"}
    if rank == 0 and n_gpus == 1 and args.workload == "c2":
        # secondary workload in the same run: BASELINE configs[2] (4096 x 256, L = 88) -- not the headline value
        try:
            out["workload_c3"] = measure_secondary(kspde, local_rank, dev, "c3", args.mode)
        except Exception as exc:
            out["workload_c3"] = {"error": f"{type(exc).__name__}: {exc}"}
    if rank == 0 and n_gpus == 1:
        if not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(E, N, L)
            except Exception as exc:
                out["cpu_baseline"] = {"error": f"{type(exc).__name__}: {exc}"}
        if not args.no_tbptt:
            try:
                from pdecontrol.surrogates import bench_tbptt
                out["tbptt"] = bench_tbptt.run(device=dev)
            except Exception as exc:  # never lose the KS line to the secondary measurement
                out["tbptt"] = {"error": f"{type(exc).__name__}: {exc}"}
    if dist is not None and not args.no_tbptt:
        # data-parallel surrogate step: B = 64 sequences per rank, one flat-bucket all-reduce per step
        try:
            from pdecontrol.surrogates import bench_tbptt
            with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                dt_ddp, in_sync, loss_ddp = bench_tbptt.run_ddp(dev)
            td = torch.tensor([dt_ddp], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(td, op=dist.ReduceOp.MAX)
            out["tbptt"] = {"unit": "seqs/s", "value": n_gpus * 64 / float(td.item()), "ms_per_step": float(td.item()) * 1e3,
                            "scaling": "weak", "B_per_rank": 64, "ranks_in_sync": in_sync, "loss": loss_ddp,
                            "exchange": "one all-reduce of the flat 38 956-byte fp32 gradient bucket per step",
                            "path": "fused HIP kernels, fwd/bwd hipGraph + all-reduce + Adam hipGraph"}
        except Exception as exc:  # never lose the KS line to the secondary measurement
            out["tbptt"] = {"error": f"{type(exc).__name__}: {exc}"}
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
