#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native KS hot path (+ surrogate TBPTT step).

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 it is launched
under torch.distributed.run with one rank per GPU.  Rank 0 prints ONE JSON line.  Started DIRECTLY with
``--gpus N`` (N > 1, no WORLD_SIZE in the environment) the GPU-free parent starts the N ranks itself as fresh child
processes of ``python -m torch.distributed.run`` (never an exec, nothing has touched the GPU yet) and relays rank 0's
line; the N > 1 line carries ``ranks`` = what the process group itself reports (backend, world size, one gathered
record per rank with its device and uuid, a checked all-reduce, every rank's own step time).

What one "step" is: one ``env.step`` of the whole batch = ONE launch of the fused HIP stepper that
advances every env by cfg_steps = 250 RK4 sub-steps (pdegym/kuramoto/kuramoto.py:78-98 of the
reference), actions resident in HBM, phi = actions @ F evaluated in-kernel, fp32 observation,
reward sum and overflow flags written back to HBM.

Workload at N = 1: BASELINE.json configs[2] -- KS 256 grid points (L = 88), 4096 batched envs, fp64 --
the largest single-GPU configuration; configs[1] (1024 x 64) rides along as ``workload_c2``.  For N > 1 every rank
owns its own 4096 x 256 envs (configs[3]: 32768 x 256 over 8 GPUs; envs never interact, no data-path
collective): weak scaling, ``value`` = env-sub-steps of ALL ranks per second of the slowest rank.

Extra objects in the JSON line:
  roofline      HBM-model roofline of SURVEY.md 8(d): algorithmic bytes = 20*N per env-sub-step
                (read u fp64, write u fp64, read phi fp32), achieved = bytes per launch / average
                launch duration measured with HIP events on the launch stream.  The kernel keeps
                the state in registers for all 250 sub-steps, so its real HBM traffic is ~250x
                smaller (``traffic``, from PMC counters); the binding resource is the fp64 VALU pipe
                (``fp64_valu``, ``valu_issue``).
  cpu_baseline  (i) the CPU oracle (oracle/ks_oracle.c, bit-exact restatement of the reference stepper) on all host
                cores, and (ii) SURVEY 8(d) baseline (A): the reference's own call structure (per-env Python loop, 16
                scipy convolve1d + a torch round trip per sub-step) as P independent processes, aggregate, P stated --
                both on a bounded sample of THIS workload, timed before the GPU is touched.
  tbptt         surrogate TBPTT training step at N = 256 (B=64, T=20, tau=5, tbtt=10), seqs/s: captured graph,
                eager fused (what pl.Trainer.fit drives), plain-torch legs, CPU baseline; N = 64 secondary.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "model-based-pde-control_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

WORKLOADS = {
    # name: (envs per GPU, N, L, BASELINE.json configs index)  -- L = 88 at N = 256
    "c2": (1024, 64, 22.0, 1),
    "c3": (4096, 256, 88.0, 2),
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


def host_cores():
    """Cores this process may actually use: the affinity mask, capped by the cgroup CPU quota when there is one."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def default_reset_mode():
    try:
        from pdegym.kuramoto.batched import DEFAULT_RESET_MODE
        return DEFAULT_RESET_MODE()
    except Exception:
        return None


def kernel_source_sha():
    """Identity of the KS kernel sources: profile-derived numbers (PMC traffic, SQ counters) are only quoted while the
    kernel they were measured on is the kernel that runs."""
    h = hashlib.sha256()
    for name in ("ks_kernels.hip", "ks_internal.h"):
        with open(os.path.join(ROOT, "model-based-pde-control_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


# ---------------------------------------------------------------------------------------------------------
# CPU baselines (run BEFORE the GPU is initialised: the P worker processes are started by a GPU-free parent)
# ---------------------------------------------------------------------------------------------------------
def cpu_baseline(E, N, L, target_seconds=12.0):
    """(i) oracle/ks_oracle.c with OpenMP over envs on every host core; (ii) baseline (A) as P processes."""
    from oracle import ks_oracle as ko
    cores = min(host_cores(), 64)
    rs = np.random.RandomState(1234)
    u0 = rs.uniform(-0.4, 0.4, (E, N))
    act = np.random.RandomState(99).uniform(-1, 1, (E, 4)).astype(np.float32)
    phi = ko.phi_from_actions(act, forcing_matrix(L, N))
    ko.step(u0, phi, L / N, DT, 20, nthreads=cores)  # spin up the thread pool, warm caches
    nsub = 100
    t0 = time.perf_counter()
    ko.step(u0, phi, L / N, DT, nsub, nthreads=cores)
    probe = time.perf_counter() - t0
    nsub = int(max(50, min(CFG_STEPS * 400, nsub * target_seconds / max(probe, 1e-6))))
    t0 = time.perf_counter()
    ko.step(u0, phi, L / N, DT, nsub, nthreads=cores)
    dt = time.perf_counter() - t0
    out = {"value": E * nsub / dt, "unit": "sub-steps/s", "cores": cores, "kind": "port",
           "sample": f"oracle/ks_oracle.c (bit-exact C restatement of the reference stepper), {E} envs x {nsub} "
                     f"sub-steps, N={N}, OpenMP over envs, {dt:.1f} s"}
    try:
        out["reference_call_structure"] = reference_structure_baseline(N, L)
    except Exception as exc:  # scipy / torch CPU missing must not lose the baseline
        out["reference_call_structure"] = {"error": f"{type(exc).__name__}: {exc}"}
    try:
        out["cpu_twin"] = cpu_twin_baseline(E, N, L, cores)
    except Exception as exc:
        out["cpu_twin"] = {"error": f"{type(exc).__name__}: {exc}"}
    return out


def cpu_twin_baseline(E, N, L, cores, seconds=4.0):
    """SURVEY 8(d) baseline (B): the product's own CPU twin behind the C ABI (device = -1, csrc/ks_cpu.cpp: vectorised C++,
    envs over host threads), fast arithmetic like the GPU number, all cores and one thread.  Product code, not the oracle."""
    import kspde
    u0 = np.random.RandomState(1234).uniform(-0.4, 0.4, (E, N))
    act = np.random.RandomState(99).uniform(-1, 1, (E, 4)).astype(np.float32)
    out = {"unit": "sub-steps/s", "what": "libkspde CPU twin (device = -1), fast mode, ks_step_actions on this workload"}
    for label, threads, envs in (("all_cores", cores, E), ("one_thread", 1, max(1, E // max(cores, 1)))):
        os.environ["KSPDE_CPU_THREADS"] = str(threads)
        try:
            s = kspde.KSStepper(envs, N, L, DT, device=-1, mode="fast")
            s.set_forcing(forcing_matrix(L, N))
            s.set_state(u0[:envs])
            s.step_actions(act[:envs], 25, want_obs=False)
            t0 = time.perf_counter()
            s.step_actions(act[:envs], 50, want_obs=False)
            probe = time.perf_counter() - t0
            nsub = int(max(50, min(CFG_STEPS * 40, 50 * seconds / max(probe, 1e-6))))
            t0 = time.perf_counter()
            s.step_actions(act[:envs], nsub, want_obs=False)
            dt = time.perf_counter() - t0
            out[label] = {"value": envs * nsub / dt, "threads": threads, "envs": envs, "sub_steps": nsub, "seconds": dt}
            s.close()
        finally:
            os.environ.pop("KSPDE_CPU_THREADS", None)
    return out


def reference_structure_baseline(N, L, seconds=10.0):
    """SURVEY 8(d) baseline (A): P = host cores independent processes, each one env stepped the way the reference
    steps it (pdegym/kuramoto/kuramoto.py:78-129 through AsyncVectorEnv, pdecontrol/mbrl/mbrl.py:81-86): Python loop
    over sub-steps, 16 scipy.ndimage.convolve1d(mode="wrap") and one numpy->torch->numpy reward per sub-step.
    Aggregate sub-steps/s = sum over processes of sub-steps / the slowest process' loop time."""
    P = max(1, min(host_cores(), 64))
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "ks_oracle.py"), "--structured-worker", str(N), str(L), str(seconds)]
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
    procs = [subprocess.Popen(cmd + [str(1234 + i)], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, env=env)
             for i in range(P)]
    try:
        for p in procs:                      # every worker has imported numpy / scipy / torch and taken its first steps
            if p.stdout.readline().strip() != "ready":
                raise RuntimeError("structured-baseline worker failed to start")
        for p in procs:                      # common start
            p.stdin.write("go\n")
            p.stdin.flush()
        res = []
        for p in procs:
            line = p.communicate(timeout=seconds * 6 + 60)[0].strip().splitlines()
            if p.returncode == 0 and line:
                res.append(json.loads(line[-1]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    if not res:
        raise RuntimeError("no structured-baseline worker finished")
    total = sum(r["substeps"] for r in res)
    slowest = max(r["seconds"] for r in res)
    return {"value": total / slowest, "unit": "sub-steps/s", "processes": len(res), "cores": len(res),
            "per_process": total / slowest / len(res),
            "sample": f"{len(res)} processes x 1 env (N={N}) x ~{res[0]['substeps']} sub-steps through scipy.ndimage.convolve1d + "
                      f"torch.norm, Python loop per sub-step, released together, slowest loop {slowest:.1f} s"}


# ---------------------------------------------------------------------------------------------------------
# GPU measurement of one workload
# ---------------------------------------------------------------------------------------------------------
def profile_extras(name):
    """PMC traffic / SQ issue counters of this workload from profiles/ -- only while they describe the current kernel."""
    sha = kernel_source_sha()
    traffic, traffic_src, valu_issue = None, None, None
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "ks_pmc_traffic.json"))).get(name, {})
        if d.get("kernel_source_sha") == sha:
            traffic = d.get("hbm_bytes_per_launch")
            traffic_src = "profiles/ks_pmc_traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command " \
                          "(FETCH_SIZE doubled per the gfx950 guide), kernel sources unchanged since (sha %s)" % sha
        elif d:
            traffic_src = "profiles/ks_pmc_traffic.json is STALE (kernel sources changed since it was measured): not quoted"
    except (OSError, ValueError):
        pass
    try:
        sq = json.load(open(os.path.join(ROOT, "profiles", "ks_sq_counters.json"))).get(name)
        if sq and sq.get("kernel_source_sha") == sha:
            valu_issue = {"frac_of_wave_cycles": sq["fractions_of_wave_cycles"]["SQ_ACTIVE_INST_VALU"],
                          "wait_frac": sq["fractions_of_wave_cycles"]["SQ_WAIT_ANY"],
                          "valu_instructions_per_point_substep": sq["valu_instructions_per_point_substep"],
                          "source": "profiles/ks_sq_counters.json (rocprofv3 --pmc, SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES)"}
    except (OSError, ValueError, KeyError):
        pass
    return traffic, traffic_src, valu_issue


def roofline_of(name, E, N, kernel_ms, layout):
    alg_bytes = 20.0 * N * E * CFG_STEPS
    gbs = alg_bytes / (kernel_ms * 1e-3) / 1e9
    flops = FLOPS_PER_POINT_SUBSTEP * N * E * CFG_STEPS
    tf = flops / (kernel_ms * 1e-3) / 1e12
    traffic, traffic_src, valu_issue = profile_extras(name)
    return {
        "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
        "traffic": traffic, "traffic_note": traffic_src,
        "kernel": "ks_rk4_fused<P=%s,G=%s>" % (layout.get("points_per_lane"), layout.get("lanes_per_env")),
        "avg_launch_ms": kernel_ms, "algorithmic_bytes_per_launch": alg_bytes,
        "note": "algorithmic 20*N B per env-sub-step (SURVEY 8d) x E x 250 per launch; the state stays in VGPRs for all 250 "
                "sub-steps so real HBM traffic is ~1/250 of this; the binding resource is fp64 VALU issue",
        "fp64_valu": {"achieved": tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_PEAK_TFLOPS,
                      "flops_per_point_substep": FLOPS_PER_POINT_SUBSTEP},
        "valu_issue": valu_issue,
    }


class KSRun:
    """Stepper + synthetic inputs of one workload, resident in HBM (IC ~ U(-0.4, 0.4) per env seed, 1000
    sub-steps onto the attractor, actions ~ U(-1, 1) redrawn every step)."""

    def __init__(self, kspde, name, local_rank, dev, rank, steps, mode, variant="auto"):
        self.E, self.N, self.L, self.cfg = WORKLOADS[name]
        E, N, L = self.E, self.N, self.L
        self.name, self.dev, self.mode = name, dev, mode
        self.stepper = kspde.KSStepper(E, N, L, DT, device=local_rank, mode=mode, variant=variant)
        self.stream = torch.cuda.Stream(device=dev)
        self.stepper.set_stream(self.stream.cuda_stream)
        self.stepper.set_forcing(forcing_matrix(L, N))
        base = rank * E
        self.stepper.set_state(np.stack([np.random.RandomState(1234 + base + e).uniform(-0.4, 0.4, N) for e in range(E)]))
        for _ in range(4):  # 1000 sub-steps onto the attractor, as 250-sub-step launches like the timed ones
            self.stepper.step(None, CFG_STEPS, want_obs=False)
        self.acts = torch.from_numpy(np.random.RandomState(99 + rank).uniform(-1, 1, (steps, E, 4)).astype(np.float32)).to(dev)
        self.d_obs = torch.empty((E, N), dtype=torch.float32, device=dev)
        self.d_ssq = torch.empty(E, dtype=torch.float64, device=dev)
        self.d_st = torch.zeros(E, dtype=torch.int32, device=dev)
        self.st_acc = torch.zeros(E, dtype=torch.int32, device=dev)
        torch.cuda.synchronize(dev)

    def one_step(self, i):
        self.stepper.step_device(d_actions=self.acts[i].data_ptr(), n_substeps=CFG_STEPS, d_obs=self.d_obs.data_ptr(),
                                 d_ssq=self.d_ssq.data_ptr(), d_status=self.d_st.data_ptr())

    def timed(self, K, W, barrier):
        """W warm-up launches, then K launches bracketed by barrier(); returns (elapsed s, mean launch ms from ONE pair of
        HIP events recorded on the launch stream around the K back-to-back launches).  Round 2 recorded a pair PER launch:
        every record is a barrier packet that drains the queue before the next dispatch, which cost 5 % (C3) to 13 % (C2)
        of the very time being measured (tools/ks_substep_sweep.py: same box, same launches, no events in between)."""
        with torch.cuda.stream(self.stream):
            for i in range(W):
                self.one_step(i)
            barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record(self.stream)
            for i in range(K):
                self.one_step(W + i)
            e1.record(self.stream)
            barrier()
            elapsed = time.perf_counter() - t0
            self.st_acc |= self.d_st
        torch.cuda.synchronize(self.dev)
        assert int(self.st_acc.sum()) == 0, "non-finite state during the benchmark"
        return elapsed, e0.elapsed_time(e1) / K


def end_to_end_leg(run, K=20, W=3):
    """SURVEY 8(d) "end-to-end" figure: the same step through the HOST-buffer entry (ks_step_actions: actions [E, 4] fp32
    from pageable host memory, fp32 observations + reward sums + status back into host arrays, one synchronisation) --
    PCIe inclusive, what gym's step() costs a host-side controller.  Never ``value``."""
    st, E, N = run.stepper, run.E, run.N
    acts = np.random.RandomState(7).uniform(-1, 1, (K + W, E, 4)).astype(np.float32)
    torch.cuda.synchronize(run.dev)
    for i in range(W):
        st.step_actions(acts[i], CFG_STEPS)
    t0 = time.perf_counter()
    for i in range(K):
        _, _, status = st.step_actions(acts[W + i], CFG_STEPS)
    dt = (time.perf_counter() - t0) / K
    assert not status.any()
    gbs = 20.0 * N * E * CFG_STEPS / dt / 1e9
    return {"value": E * CFG_STEPS / dt, "unit": "sub-steps/s", "ms_per_step": dt * 1e3, "steps": K,
            "hbm_model_frac": gbs / HBM_PEAK_GBS,
            "host_bytes_per_step": {"h2d_actions": E * 16, "d2h_obs_f32": E * N * 4, "d2h_reward_status": E * 12},
            "what": "ks_step_actions: host actions in, host obs / reward sums / status out, synchronous (PCIe inclusive)"}


def episode_leg(run, reset_mode, ring=8):
    """What a real run of the reference's loop sees per episode (pdegym/kuramoto/kuramoto.py:100-116: the reset burn-in is
    800 step-equivalents, two thirds of all sub-steps): max_episode_steps = 400 steps of every env in the step arithmetic
    (fast), then the autoreset all envs take together -- fresh initial conditions uploaded + ONE launch of the 200 000
    sub-step burn-in in ``reset_mode``.  Device resident like ``value`` (seeding and drawing the ICs on the host --
    pdegym/kuramoto/mt_batch.py, bit-identical to one RandomState per env -- is outside the timed region and reported
    beside it)."""
    st, E, N, dev = run.stepper, run.E, run.N, run.dev
    steps, burn = 400, 200000
    from pdegym.kuramoto.mt_batch import BatchedMT19937
    t0 = time.perf_counter()
    mt = BatchedMT19937(E)                # the vector envs' IC generator: env e <- RandomState(seed + e), as array operations
    mt.seed_rows(np.arange(E), [4321 + e for e in range(E)])
    u0 = mt.uniform_rows(np.arange(E), -0.4, 0.4, N)
    ic_ms = (time.perf_counter() - t0) * 1e3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    with torch.cuda.stream(run.stream):
        for i in range(steps):
            run.one_step(i % ring)
        t_steps = None
        st.set_state(u0)                       # synchronises: the 400 steps are done here
        t_steps = time.perf_counter() - t0
        st.set_mode(reset_mode)
        e0.record(run.stream)
        st.step_device(n_substeps=burn, d_obs=run.d_obs.data_ptr(), d_ssq=run.d_ssq.data_ptr(), d_status=run.d_st.data_ptr())
        e1.record(run.stream)
        st.set_mode(run.mode)
    torch.cuda.synchronize(dev)
    total = time.perf_counter() - t0
    assert int(run.d_st.sum()) == 0, "non-finite state after the burn-in"
    burn_ms = e0.elapsed_time(e1)
    sub = E * (steps * CFG_STEPS + burn)
    frac = lambda substeps, seconds: 20.0 * N * substeps / seconds / 1e9 / HBM_PEAK_GBS
    return {"value": sub / total, "unit": "sub-steps/s", "seconds_per_episode": total, "hbm_model_frac": frac(sub, total),
            "steps": {"count": steps, "seconds": t_steps, "mode": run.mode},
            "reset": {"mode": reset_mode, "sub_steps": burn, "kernel_ms": burn_ms,
                      "hbm_model_frac": frac(E * burn, burn_ms * 1e-3), "ic_upload_included": True,
                      "ic_draw_on_host_ms_not_included": ic_ms}}


VALU_ISSUE_PEAK = 1024 * 2.4e9 / 4 / 1e9     # G wave-instructions/s: 256 CUs x 4 SIMDs, one wave64 VALU op per 4 cycles, 2.4 GHz


def burgers_roofline(E, N, cfg_steps, ms):
    """The Burgers stepper keeps its state in VGPRs for all sub-steps of a launch: HBM is not what bounds it (the 12*N
    streaming model would read as 2.5x the HBM peak -- not a roofline).  The binding resource is VALU issue: every wave64
    VALU instruction holds its SIMD for >= 4 cycles, so achieved = wave-instructions per second (SQ_INSTS_VALU per
    launch from profiles/burgers_sq_counters.json / this run's launch time) against 1024 SIMDs x 2.4 GHz / 4."""
    alg = 12.0 * N * E * cfg_steps
    out = {"bound": "valu_issue", "achieved": None, "peak": VALU_ISSUE_PEAK, "unit": "G wave-instructions/s", "frac": None,
           "hbm_streaming_model_gbs": alg / (ms * 1e-3) / 1e9,
           "note": "fp32 stencil with the state in registers for all 50 sub-steps of a launch: VALU-issue bound; the 12*N B "
                   "per env-sub-step streaming figure is kept as a rate only, it is not a bound for this kernel"}
    try:
        sq = json.load(open(os.path.join(ROOT, "profiles", "burgers_sq_counters.json")))["c4"]
        sha = hashlib.sha256(open(os.path.join(ROOT, "model-based-pde-control_amd", "csrc", "burgers.hip"), "rb").read()).hexdigest()[:16]
        if sq.get("kernel_source_sha") == sha:
            per_point = sq["valu_instructions_per_point_substep"]
            insts = per_point * E * N * cfg_steps / 64.0
            out["achieved"] = insts / (ms * 1e-3) / 1e9
            out["frac"] = out["achieved"] / VALU_ISSUE_PEAK
            out["valu_instructions_per_point_substep"] = per_point
            busy = sq.get("fractions_of_wave_cycles", {}).get("SQ_ACTIVE_INST_VALU")
            waves = sq.get("mean_per_launch", {}).get("SQ_WAVES")
            out["valu_busy_frac_of_wave_cycles"] = busy
            if busy and waves:
                # SQ_WAVE_CYCLES sums over the waves resident on a SIMD: x waves per SIMD = how busy each SIMD's VALU is
                out["valu_busy_per_simd"] = busy * waves / 1024.0
                out["note"] += "; packed fp32 instructions (v_pk_*_f32) occupy the pipe for two passes, which is why the " \
                               "VALU is busy nearly all cycles (valu_busy_per_simd) at this fraction of the single-pass issue peak"
            out["source"] = "profiles/burgers_sq_counters.json (rocprofv3 --pmc SQ_INSTS_VALU, kernel sources unchanged since)"
        else:
            out["source"] = "profiles/burgers_sq_counters.json is STALE (burgers.hip changed since): fraction not quoted"
    except (OSError, ValueError, KeyError):
        out["source"] = "no SQ counters on file: fraction not quoted"
    return out


def burgers_fno_leg(dev, steps=30, warmup=5):
    """BASELINE configs[4] (second PDE path; the reference ships neither a Burgers env nor an FNO, so nothing here has a
    reference-side pin beyond the discretisation, tests/test_burgers.py): the fp32 Burgers stepper at 512 grid points and
    the TBPTT training step of the FNO-style surrogate on the whole-network kernels (eager, and captured as one hipGraph)."""
    from pdegym.burgers import make_vec
    E, N = 8192, 512
    env = make_vec(E, config=dict(N=N), device=dev.index or 0)
    env.reset(seed=0)
    acts = torch.from_numpy(np.random.RandomState(5).uniform(-1, 1, (steps + warmup, E, 4)).astype(np.float32)).to(dev)
    for i in range(warmup):
        env.step_torch(acts[i])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps):
        env.step_torch(acts[warmup + i])
    e1.record()
    torch.cuda.synchronize(dev)
    assert int(env._status.sum()) == 0
    ms = e0.elapsed_time(e1) / steps
    out = {"workload": f"Burgers nu={env.nu} N={N} dt={env.dt} cfg_steps={env.cfg_steps}, {E} batched envs (BASELINE.json configs[4], "
                       f"one GPU's share), fp32",
           "value": E * env.cfg_steps / (ms * 1e-3), "unit": "sub-steps/s", "avg_launch_ms": ms,
           "roofline": burgers_roofline(E, N, env.cfg_steps, ms)}
    # FNO surrogate: eager training_step + backward + Adam, B = 64, T = 20, N = 512 (width 32, 16 modes, 4 layers)
    from pdecontrol.architectures import BurgersFNO
    from pdecontrol.surrogates.training import PDETrainingModule
    torch.manual_seed(0)
    f = BurgersFNO()
    sur = f.surrogate(delta=0.05, dscaling=None, tau=5, **f.model())
    mod = PDETrainingModule(surrogate=sur, loss=torch.nn.MSELoss(reduction="none"), tstep=0.05, delta=0.05, tau=5, tbtt=10).to(dev)
    g = torch.Generator().manual_seed(1)
    batch = ((torch.rand(64, 20, 1, N, generator=g) * 2 - 1).to(dev), (torch.rand(64, 20, 1, N, generator=g) * 2 - 1).to(dev))
    opt = mod.configure_optimizers()[0][0]

    def one():
        o = mod.training_step(batch, 0)
        opt.zero_grad(set_to_none=True)
        o["loss"].backward()
        opt.step()
    for _ in range(3):
        one()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(10):
        one()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / 10
    out["fno_tbptt"] = {"value": 64 / dt, "unit": "seqs/s", "ms_per_step": dt * 1e3,
                        "config": {"factory": "BurgersFNO", "width": 32, "modes": 16, "layers": 4, "B": 64, "T": 20, "N": N},
                        "path": "eager; whole-network HIP kernels (csrc/fno.hip): one launch per model evaluation and direction, one "
                                "autograd node per TBPTT chunk"}
    try:   # the same step captured as one hipGraph (PDETrainingModule.fused_step is architecture-agnostic)
        mod.fused_step(batch)
        for _ in range(2):
            mod.fused_step(batch)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(10):
            mod.fused_step(batch)
        torch.cuda.synchronize(dev)
        dg = (time.perf_counter() - t0) / 10
        out["fno_tbptt"]["hip_graph"] = {"value": 64 / dg, "ms_per_step": dg * 1e3}
    except Exception as exc:
        out["fno_tbptt"]["hip_graph"] = {"error": f"{type(exc).__name__}: {exc}"}
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

# ---------------------------------------------------------------------------------------------------------
# N > 1: rank processes and what they report about themselves
# ---------------------------------------------------------------------------------------------------------
def spawn_ranks(n, argv):
    """``python bench.py --gpus N`` started directly: start N fresh rank processes (torch.distributed.run, one per GPU,
    rendezvous on 127.0.0.1) from this process, which has not touched the GPU, and relay rank 0's JSON line.  On a box
    with fewer GPUs than ranks the ranks share devices over gloo -- a rehearsal of the code path, flagged as such in the
    line (``ranks.rehearsal``), never a scaling number.  Returns the launcher's exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    if "BENCH_DIST_BACKEND" not in env and torch.cuda.device_count() < n:   # device_count() does not initialise the GPU
        env["BENCH_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for raw in proc.stdout:             # rank 0 prints exactly one JSON line; anything else is passed to stderr
        txt = raw.strip()
        if txt.startswith("{") and '"metric"' in txt:
            line = txt
        elif txt:
            print(txt, file=sys.stderr)
    rc = proc.wait()
    if line is not None:
        print(line)
    elif rc == 0:
        rc = 1
    return rc


def init_process_group_checked(dist, backend, dev, world, timeout_s=600):
    """Initialise the process group and PROVE the backend with one checked all-reduce before anything is timed.  The KS path
    shards envs without a data-path collective, so when RCCL cannot be brought up on this node (driver / IPC mode / two ranks
    on one device) the measurement does not have to be lost: every rank then re-initialises over gloo, and the line says so
    (``ranks.requested_backend``, ``ranks.backend_error``) instead of there being no line.  Returns (backend in use, note)."""
    import datetime
    timeout = datetime.timedelta(seconds=timeout_s)
    try:
        dist.init_process_group(backend, timeout=timeout)
        cdev = dev if backend == "nccl" else "cpu"
        x = torch.ones(1, dtype=torch.float64, device=cdev)
        dist.all_reduce(x, op=dist.ReduceOp.SUM)
        if backend == "nccl":
            torch.cuda.synchronize(dev)
        if int(x.item()) != world:
            raise RuntimeError(f"all-reduce of ones over {world} ranks returned {x.item()}")
        return backend, None
    except Exception as exc:
        if backend == "gloo":
            raise
        note = {"requested_backend": backend, "backend_error": f"{type(exc).__name__}: {str(exc)[:400]}"}
        try:
            dist.destroy_process_group()
        except Exception:
            pass
        dist.init_process_group("gloo", timeout=timeout)
        return "gloo", note


def rank_evidence(dist, backend, dev, rank, local_rank, world, elapsed, kernel_ms, K):
    """What the process group itself says about the ranks: one gathered record per rank, a checked all-reduce on the
    backend's own device type, every rank's step time (a straggler shows).  Every rank calls this."""
    props = torch.cuda.get_device_properties(dev)
    me = {"rank": rank, "local_rank": local_rank, "device": torch.cuda.current_device(),
          "uuid": str(getattr(props, "uuid", "")), "pci_bus_id": getattr(props, "pci_bus_id", None), "name": props.name,
          "gcn_arch": getattr(props, "gcnArchName", None), "pid": os.getpid(), "host": socket.gethostname(),
          "ms_per_step": elapsed / K * 1e3, "kernel_ms": kernel_ms}
    gathered = [None] * world
    dist.all_gather_object(gathered, me)
    cdev = dev if backend == "nccl" else "cpu"
    x = torch.tensor([rank + 1.0], dtype=torch.float64, device=cdev)
    dist.all_reduce(x, op=dist.ReduceOp.SUM)
    devices = {(r["host"], r["uuid"] or r["device"]) for r in gathered}
    return {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
            "collective": "RCCL (torch 'nccl' backend on ROCm)" if backend == "nccl" else backend,
            "all_reduce_check": {"sum_of_rank_plus_1": float(x.item()), "expected": world * (world + 1) / 2.0,
                                 "ok": float(x.item()) == world * (world + 1) / 2.0, "on": str(cdev)},
            "distinct_devices": len(devices), "rehearsal": len(devices) < world,
            "ms_per_step": [r["ms_per_step"] for r in gathered], "kernel_ms": [r["kernel_ms"] for r in gathered],
            "per_rank": [{k: v for k, v in r.items() if k not in ("ms_per_step", "kernel_ms")} for r in gathered]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=list(WORKLOADS), default="c3")
    ap.add_argument("--mode", choices=["fast", "exact"], default="fast")
    ap.add_argument("--variant", default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tbptt", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-burgers", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the end_to_end / episode_inclusive legs")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_gpus = world if world > 1 else args.gpus
    if world == 1 and args.gpus > 1:
        # started directly: this process stays GPU-free and starts the ranks itself
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    E, N, L, cfg_idx = WORKLOADS[args.workload]
    K, W = args.steps, args.warmup

    # ---- CPU baselines first: nothing has touched the GPU yet, so starting worker processes is safe ----
    cpu_ks, cpu_tbptt = None, None
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        try:
            cpu_ks = cpu_baseline(E, N, L)
        except Exception as exc:
            cpu_ks = {"error": f"{type(exc).__name__}: {exc}"}
        if not args.no_tbptt:
            try:
                from pdecontrol.surrogates import bench_tbptt
                cpu_tbptt = bench_tbptt.cpu_baseline(N=256, B=64, steps=10)
            except Exception as exc:
                cpu_tbptt = {"error": f"{type(exc).__name__}: {exc}"}

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the HIP stepper has no CPU fallback")
    # one rank per GPU; ranks beyond the visible GPU count share devices (only used to rehearse the
    # multi-rank code path on a 1-GPU box, with BENCH_DIST_BACKEND=gloo)
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dist = None
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
    backend_note = None
    if world > 1:
        import torch.distributed as dist
        backend, backend_note = init_process_group_checked(dist, backend, torch.device("cuda", local_rank), world)

    import kspde
    dev = torch.device("cuda", local_rank)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    run = KSRun(kspde, args.workload, local_rank, dev, rank, K + W, args.mode, args.variant)
    elapsed, kernel_ms = run.timed(K, W, barrier)
    ranks = None
    if dist is not None:
        ranks = rank_evidence(dist, backend, dev, rank, local_rank, world, elapsed, kernel_ms, K)
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    lay = run.stepper.layout()
    total_substeps = n_gpus * E * CFG_STEPS * K
    out = {
        "metric": "KS sub-steps/sec (batched env)",
        "value": total_substeps / elapsed,
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
                        f"(BASELINE.json configs[{cfg_idx if n_gpus == 1 else 3}]"
                        f"{'' if n_gpus == 1 else f': {n_gpus * E} envs over {n_gpus} GPUs'}), random-action rollout",
            "envs_per_gpu": E, "grid_points": N, "mode": args.mode, "kernel": lay,
            "sharding": "envs sharded by rank, no collective" if n_gpus > 1 else "single GPU",
        },
        "roofline": roofline_of(args.workload, E, N, kernel_ms, lay),
    }
    if ranks is not None:
        if backend_note:
            ranks.update(backend_note)
        out["ranks"] = ranks
    if rank == 0 and n_gpus == 1:
        try:
            out["roofline"]["hbm_copy_measured"] = {"value": measured_hbm_copy_gbs(dev), "unit": "GB/s",
                                                    "what": "1 GiB device-to-device copy, read + write bytes"}
        except Exception as exc:
            out["roofline"]["hbm_copy_measured"] = {"error": f"{type(exc).__name__}: {exc}"}
        if cpu_ks is not None:
            out["cpu_baseline"] = cpu_ks
        if not args.no_extras:
            # the two figures SURVEY 8(d) asks for next to ``value``: PCIe-inclusive, and with the reset burn-in
            try:
                out["end_to_end"] = end_to_end_leg(run)
            except Exception as exc:
                out["end_to_end"] = {"error": f"{type(exc).__name__}: {exc}"}
            out["episode_inclusive"] = {
                "what": "400 env steps + the autoreset of every env (fresh ICs + 200 000-sub-step burn-in, "
                        "kuramoto.py:100-116), device resident; per reset arithmetic",
                "default_reset_mode": default_reset_mode()}
            for rm in ("fast", "exact"):
                try:
                    out["episode_inclusive"][rm + "_reset"] = episode_leg(run, rm)
                except Exception as exc:
                    out["episode_inclusive"][rm + "_reset"] = {"error": f"{type(exc).__name__}: {exc}"}
        if not args.no_secondary:
            # the other single-GPU BASELINE config in the same run -- not the headline value
            other = "c2" if args.workload == "c3" else "c3"
            try:
                E2, N2, L2, idx2 = WORKLOADS[other]
                # enough launches for the clocks to settle after the headline workload (C2 launches are 0.1 ms)
                K2, W2 = (200, 30) if other == "c2" else (30, 5)
                sec = KSRun(kspde, other, local_rank, dev, rank, K2 + W2, args.mode)
                el2, ms2 = sec.timed(K2, W2, lambda: torch.cuda.synchronize(dev))
                out["workload_" + other] = {
                    "workload": f"KS L={L2:g} N={N2}, {E2} envs (BASELINE.json configs[{idx2}])",
                    "value": E2 * CFG_STEPS * K2 / el2, "steps": K2, "unit": "sub-steps/s", "avg_launch_ms": ms2, "kernel": sec.stepper.layout(),
                    "roofline": roofline_of(other, E2, N2, ms2, sec.stepper.layout())}
                del sec
            except Exception as exc:
                out["workload_" + other] = {"error": f"{type(exc).__name__}: {exc}"}
        if not args.no_tbptt:
            try:
                from pdecontrol.surrogates import bench_tbptt
                out["tbptt"] = bench_tbptt.run(device=dev, cpu=cpu_tbptt)
            except Exception as exc:  # never lose the KS line to the secondary measurement
                out["tbptt"] = {"error": f"{type(exc).__name__}: {exc}"}
        if not args.no_burgers and not args.no_secondary:
            try:
                out["burgers_fno"] = burgers_fno_leg(dev)
            except Exception as exc:
                out["burgers_fno"] = {"error": f"{type(exc).__name__}: {exc}"}
    if dist is not None and not args.no_tbptt:
        # data-parallel surrogate step at N = 256: B = 64 sequences per rank, one flat-bucket all-reduce per step
        ddp, err = None, None
        try:
            from pdecontrol.surrogates import bench_tbptt
            with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                ddp = bench_tbptt.run_ddp(dev, N=256)
        except Exception as exc:  # never lose the KS line to the secondary measurement
            err = f"{type(exc).__name__}: {exc}"
        # every rank reaches the same collectives whatever happened above: a rank that failed must not leave the others
        # waiting in the timing reduction
        cdev = dev if backend == "nccl" else "cpu"
        ok = torch.tensor([0.0 if ddp is None else 1.0], dtype=torch.float64, device=cdev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        td = torch.tensor([0.0 if ddp is None else ddp[0]], dtype=torch.float64, device=cdev)
        dist.all_reduce(td, op=dist.ReduceOp.MAX)
        # "in sync" is every rank's comparison with rank 0's parameters: rank 0's own is trivially true
        sync = torch.tensor([1.0 if (ddp is not None and ddp[1]) else 0.0], dtype=torch.float64, device=cdev)
        dist.all_reduce(sync, op=dist.ReduceOp.MIN)
        if float(ok.item()) == 1.0:
            _dt, _, loss_ddp, nbytes = ddp
            in_sync = float(sync.item()) == 1.0
            out["tbptt"] = {"unit": "seqs/s", "value": n_gpus * 64 / float(td.item()), "ms_per_step": float(td.item()) * 1e3,
                            "scaling": "weak", "B_per_rank": 64, "N": 256, "ranks_in_sync": in_sync, "loss": loss_ddp,
                            "exchange": f"one all-reduce of the flat {nbytes}-byte fp32 gradient bucket per step",
                            "backend": dist.get_backend(),
                            "path": "fused HIP kernels, fwd/bwd hipGraph (chunks pipelined) + all-reduce + Adam hipGraph"}
        else:
            out["tbptt"] = {"error": err or "the data-parallel step failed on another rank"}
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
