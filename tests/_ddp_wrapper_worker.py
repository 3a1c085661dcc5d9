"""Worker of tests/test_ddp_wrapper.py: PDETrainingModule wrapped in torch's DistributedDataParallel the way
pytorch-lightning's ``strategy="ddp"`` wraps it (the reference switches that on from the CLI: Trainer kwargs are splatted
from ``--trainer``, pdecontrol/mbrl/mbrl.py:357-365), driven in Lightning's closure order for a few steps, against
single-process training on the global batch.  Started once per rank with RANK / WORLD_SIZE / MASTER_* set.

argv: <cpu|cuda> <fused 0|1> <out.json>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "model-based-pde-control_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from torch.nn.parallel import DistributedDataParallel  # noqa: E402

device, fused, out_path = sys.argv[1], sys.argv[2] == "1", sys.argv[3]
if not fused:
    os.environ["PDECONTROL_FUSED"] = "0"
from pdecontrol.surrogates import hipops, ops  # noqa: E402
from pdecontrol.surrogates.bench_tbptt import build_module, synthetic_batch  # noqa: E402
from pdecontrol.surrogates.distributed import shard_batch  # noqa: E402

torch.set_num_threads(2)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda", 0) if device == "cuda" else torch.device("cpu")
if dev.type == "cuda":
    torch.cuda.set_device(dev)
STEPS, B = 3, 4
full = synthetic_batch(B=B * world, device=dev)
mine = shard_batch(full, rank, world)


class LightningStyleWrapper(torch.nn.Module):
    """What pytorch_lightning.overrides.base._LightningModuleWrapperBase does while training: forward = training_step."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, batch, batch_idx):
        return self.module.training_step(batch, batch_idx)


def flat(m):
    return torch.cat([p.detach().reshape(-1) for p in m.surrogate.parameters()]).cpu()


def closure_steps(step_fn, module, opt, batch):
    losses = []
    for k in range(STEPS):
        out = step_fn(batch, k)
        opt.zero_grad(set_to_none=True)
        out["loss"].backward()
        opt.step()
        losses.append(float(out["loss"].detach()))
    return losses


# single process, global batch (every rank computes it: deterministic kernels)
ref = build_module(dev, seed=0)
ref_losses = closure_steps(ref.training_step, ref, ref.configure_optimizers()[0][0], full)

m = build_module(dev, seed=rank)                       # ranks start apart; the wrapper broadcasts rank 0's weights
ddp = DistributedDataParallel(LightningStyleWrapper(m), device_ids=[0] if dev.type == "cuda" else None)
opt = m.configure_optimizers()[0][0]
losses = closure_steps(ddp, m, opt, mine)
if dev.type == "cuda":
    torch.cuda.synchronize()
mean = torch.tensor(losses, dtype=torch.float64)
dist.all_reduce(mean)
mean /= world
mine_flat = flat(m)
ref0 = mine_flat.clone()
dist.broadcast(ref0, src=0)
same = torch.tensor([1.0 if torch.equal(ref0, mine_flat) else 0.0])
dist.all_reduce(same, op=dist.ReduceOp.MIN)
report = {"world": world, "device": device, "fused": bool(fused and dev.type == "cuda" and ops.fused_enabled()),
          "optimizer": type(opt).__name__, "pack_adam": isinstance(opt, hipops.PackAdam),
          "ranks_in_sync": bool(same.item() == 1.0),
          "max_param_diff_vs_single_process": float((mine_flat - flat(ref)).abs().max()),
          "mean_shard_loss": mean.tolist(), "single_process_loss": ref_losses}
dist.barrier()
if rank == 0:
    json.dump(report, open(out_path, "w"))
dist.destroy_process_group()
