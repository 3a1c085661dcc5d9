"""`python bench.py --gpus N` started directly (N > 1): the GPU-free parent starts the rank processes itself and relays
rank 0's line, which carries what the process group reports about its ranks.  The shape being replaced is the reference's
one-process-per-env data parallelism (pdecontrol/mbrl/mbrl.py:81-86).

On a one-GPU box the two ranks share the device over gloo (`ranks.rehearsal` true): this pins the code path, it is not a
scaling number."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, timeout):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def test_direct_multi_gpu_invocation_starts_its_own_ranks():
    """Without a GPU the ranks themselves refuse (no CPU fallback) -- but they were started: the parent no longer exits
    with "launch with torch.distributed.run"."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-tbptt"], 300)
    assert r.returncode != 0
    assert "needs an MI355X" in r.stderr
    assert "launch with" not in r.stderr


@pytest.mark.gpu
def test_two_ranks_started_by_bench_itself():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], 900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["steps"] == 3
    rk = out["ranks"]
    assert rk["world_size"] == 2 and rk["backend"] in ("gloo", "nccl")
    assert rk["all_reduce_check"]["ok"]
    assert [r_["rank"] for r_ in rk["per_rank"]] == [0, 1]
    assert len({r_["pid"] for r_ in rk["per_rank"]}) == 2
    assert len(rk["ms_per_step"]) == 2 and all(t > 0 for t in rk["ms_per_step"])
    # value = sub-steps of ALL ranks over the slowest rank's time
    E = out["config"]["envs_per_gpu"]
    assert out["value"] == pytest.approx(2 * E * 250 * 3 / (out["ms_per_step"] * 3e-3), rel=1e-9)
    assert out["ms_per_step"] >= max(rk["ms_per_step"]) * (1 - 1e-9)
    # the data-parallel surrogate step ran on both ranks and they ended with identical parameters
    assert out["tbptt"].get("ranks_in_sync") is True, out["tbptt"]
    import torch
    if torch.cuda.device_count() < 2:
        assert rk["rehearsal"] is True and rk["backend"] == "gloo"


@pytest.mark.gpu
def test_driver_style_launch_survives_a_backend_that_cannot_start():
    """The driver's own launch line (``python -m torch.distributed.run ... bench.py --gpus 2``, backend left at its default =
    RCCL) on a box where RCCL cannot form the group -- here: two ranks on ONE device ("Duplicate GPU detected").  The checked
    first all-reduce fails on every rank, the ranks re-initialise over gloo, the line is still printed and says what happened.
    (On a node with one GPU per rank the same code path keeps nccl; this is the only RCCL failure a one-GPU box can stage.)"""
    import socket
    import subprocess
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs the one-GPU box: with a device per rank RCCL starts")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.pop("BENCH_DIST_BACKEND", None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-tbptt"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, r.stdout[-2000:]
    rk = json.loads(lines[0])["ranks"]
    assert rk["backend"] == "gloo" and rk["requested_backend"] == "nccl" and rk["backend_error"], rk
    assert rk["all_reduce_check"]["ok"] and rk["world_size"] == 2
