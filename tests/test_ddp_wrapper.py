"""The ``--trainer '{"strategy": "ddp"}'`` route (pdecontrol/mbrl/mbrl.py:357-365): Lightning wraps the module in torch's
DistributedDataParallel.  Two gloo ranks, Lightning's closure order, three optimizer steps, against single-process training
on the global batch:

* plain path (CPU here; PDECONTROL_FUSED=0 on the GPU): gradients come out of autograd, DDP's hooks average them;
* fused path (GPU): the fused backward writes ``param.grad`` outside autograd, so the wrapper's hooks never fire -- and
  never start a reduction either, so the wrapper is inert -- while ``PackAdam.step()`` averages the pack gradients over the
  initialised process group itself.  Same result, pinned here.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "_ddp_wrapper_worker.py")


def _run(device, fused, tmp_path, world=2):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tmp_path / f"ddp_{device}_{int(fused)}.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), OMP_NUM_THREADS="2")
    env.pop("PDECONTROL_FUSED", None)
    procs = [subprocess.Popen([sys.executable, WORKER, device, "1" if fused else "0", str(out)],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              text=True) for r in range(world)]
    logs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(log[-3000:] for log in logs)
    return json.load(open(out))


def _check(rep, param_tol):
    assert rep["ranks_in_sync"], rep
    # mean of the equal shards' means == the global mean (fp32 summation order differs)
    for a, b in zip(rep["mean_shard_loss"], rep["single_process_loss"]):
        assert abs(a - b) <= 2e-5 * abs(b), rep
    assert rep["max_param_diff_vs_single_process"] <= param_tol, rep


def test_ddp_wrapped_module_plain_path_cpu(tmp_path):
    rep = _run("cpu", False, tmp_path)
    assert rep["optimizer"] == "Adam" and not rep["fused"]
    # three Adam steps of lr = 1e-3: entries whose gradient is at rounding level may step the other way (2 x 3 x lr)
    _check(rep, 2e-5)
    print("ddp wrapper, plain CPU path:", rep)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False])
def test_ddp_wrapped_module_on_one_shared_gpu(tmp_path, fused):
    rep = _run("cuda", fused, tmp_path)
    assert rep["fused"] == fused and rep["pack_adam"] == fused
    _check(rep, 2e-5)
    print("ddp wrapper, GPU, fused =", fused, rep)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump(rep, open(os.path.join(ROOT, "gpurun_out", f"ddp_wrapper_fused{int(fused)}.json"), "w"), indent=1)
    except OSError:
        pass
