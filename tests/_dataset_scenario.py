"""One scripted scenario over the replay / dataset / scheduler classes, run once against the
reference's modules (oracle/gen_golden.py) and once against this repo's (tests/test_dataset.py)."""
import numpy as np
import torch


def run(Replay, ds, sched, Sample):
    rec = {}
    W = 8
    # ---- replay: 2 sub-envs, episodes end at different times ------------------------------------
    rp = Replay()
    rs = np.random.RandomState(11)
    t = np.zeros(2, dtype=np.int64)
    ends = {0: (5, 12), 1: (8,)}          # env 0 ends after its 5th and 12th step, env 1 after its 8th
    count = np.zeros(2, dtype=np.int64)
    for step in range(14):
        samples = []
        for e in range(2):
            count[e] += 1
            t[e] += 1
            done = count[e] in ends[e]
            samples.append(Sample(rs.randn(1, W).astype(np.float32), rs.randn(1, 4).astype(np.float32),
                                  rs.randn(1, W).astype(np.float32), np.float32(rs.randn()), False, bool(done),
                                  np.int32(t[e])))
            if done:
                t[e] = 0
        rp.add(samples)
    rec["episodes"] = np.asarray(rp.episodes)
    rec["ntimesteps"] = np.asarray(rp.ntimesteps)
    rec["stopped"] = np.asarray(rp.stopped)
    rec["lengths"] = np.asarray([len(rp.obs[k]) for k in rp.episodes])
    flat = rp.dataset()
    rec["flat_obs"], rec["flat_steps"] = np.asarray(flat.obs), np.asarray(flat.steps)
    s0 = rp.sample(index=rp.episodes[1])
    rec["ep1_obs"], rec["ep1_trunc"] = s0.obs.numpy(), s0.truncated.numpy()
    mean, std = rp.statistics()
    rec["ret_mean"], rec["ret_std"] = np.asarray(mean), np.asarray(std)

    # ---- sub-sequence datasets ---------------------------------------------------------------------
    def dump(tag, dataset, n=None):
        n = len(dataset) if n is None else n
        rec[f"{tag}_len"] = np.asarray(len(dataset))
        items = [dataset[i] for i in range(n)]
        rec[f"{tag}_obs"] = np.stack([it.obs.numpy() for it in items]) if items else np.zeros(0)
        rec[f"{tag}_steps"] = np.stack([it.steps.numpy() for it in items]) if items else np.zeros(0)

    plain = ds.SubSeqDataset(rp.data, length=4, stride=2, bootstrapping=False)
    dump("plain", plain)
    rec["plain_index"] = np.asarray(plain.index)
    bounded = ds.SubSeqDataset(rp.data, subsamples=rp.episodes[:2], length=3, bootstrapping=False, bounds=(1, 1))
    dump("bounded", bounded)
    np.random.seed(0)
    boot = ds.SubSeqDataset(rp.data, length=4, stride=2, bootstrapping=True)
    rec["boot_mapping"] = np.asarray(boot.boots_mapping)
    rec["boot_index"] = np.asarray(boot.boots_index)
    dump("boot", boot)
    loader = ds.PDEDataLoader(plain, batch_size=3, shuffle=False, num_workers=0,
                              collate_fn=ds.PDEDataLoader.sample_collate)
    batches = list(loader)
    rec["loader_nbatches"] = np.asarray(len(batches))
    rec["loader_b0_obs"], rec["loader_b0_trunc"] = batches[0][0].numpy(), batches[0][5].numpy()
    rec["loader_last_actions"] = batches[-1][1].numpy()

    starting = ds.StartingStateDataset(rp.data, length=3)
    rec["starting_len"] = np.asarray(len(starting))
    rec["starting_lens"] = np.asarray([len(d) for d in starting.datasets])
    picks = [0, len(starting.datasets[0]), len(starting) - 1, len(starting) // 2]
    padded = ds.PDEDataLoader.padding_collate([starting[i] for i in picks])
    rec["padded_obs"], rec["padded_steps"] = padded.obs.numpy(), padded.steps.numpy()

    # ---- schedulers -----------------------------------------------------------------------------------
    lin = sched.LinearScheduler(steptype="iteration", start=2, stop=10, vmin=1, vmax=15)
    rec["linear"] = np.asarray([float(lin(iteration=i, epoch=0, step=0)) for i in range(14)])
    stp = sched.StepScheduler(steptype="epoch", steps=[2, 5], values=[1, 4, 9])
    rec["step"] = np.asarray([stp(iteration=0, epoch=e, step=0) for e in range(7)])
    fac = sched.Scheduler.factory({"scheduler": "ConstantLengthScheduler", "length": 7})
    rec["const"] = np.asarray(fac(iteration=3))
    return rec, rp
