"""f3: replay / sub-sequence datasets / loaders / schedulers against arrays recorded from the
reference's own classes (oracle/gen_golden.py::dataset_fixtures), plus the device-resident store."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _dataset_scenario as sc  # noqa: E402

from pdecontrol.mbrl.replay import ExperienceReplay  # noqa: E402
from pdecontrol.mbrl.types import Sample  # noqa: E402
from pdecontrol.surrogates.common import dataset as ds  # noqa: E402
from pdecontrol.surrogates.common import schedulers as sched  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dataset_golden.npz")


def test_scenario_matches_reference_bitwise():
    g = np.load(GOLDEN)
    rec, _ = sc.run(ExperienceReplay, ds, sched, Sample)
    assert sorted(rec) == sorted(g.files)
    for k in g.files:
        np.testing.assert_array_equal(np.asarray(rec[k]), g[k], err_msg=k)


def test_replay_resize_extend_and_edge_cases():
    _, rp = sc.run(ExperienceReplay, ds, sched, Sample)
    n = rp.ntimesteps
    other = ExperienceReplay(capacity=n)
    other.extend(rp)
    assert other.ntimesteps == n and other.nepisodes == rp.nepisodes
    first = min(rp.obs.keys())
    rp.resize(n - 1)  # dropping below capacity removes whole oldest episodes
    assert first not in rp.obs and rp.ntimesteps < n
    empty = ds.SubSeqDataset(rp.data, length=1000, bootstrapping=False)
    assert len(empty) == 0 and len(empty.boots_mapping) == 0


def _device():
    return torch.device("cuda", 0) if torch.cuda.is_available() else torch.device("cpu")


def test_device_store_batches_equal_dataset_items():
    from pdegym.common.transforms import BatchTransform, Normalize, SampleTransform
    _, rp = sc.run(ExperienceReplay, ds, sched, Sample)
    norm = Normalize(aggregate=True, batched=True)
    norm.mean, norm.var = torch.full((1, 1, 1), 0.2), torch.full((1, 1, 1), 1.5)
    stransf = SampleTransform(otransf=BatchTransform(norm))
    store = ds.DeviceSubSeqStore(rp.data, _device())
    assert store.total == rp.ntimesteps
    for bootstrapping in (False, True):
        np.random.seed(3)
        host = ds.SubSeqDataset(rp.data, length=4, stride=2, bootstrapping=bootstrapping, stransf=stransf)
        idx = list(range(len(host)))
        batch = store.batch(host, idx, stransf=stransf)
        ref = ds.PDEDataLoader.sample_collate([host[i] for i in idx])
        for name, got, want in zip(("obs", "actions", "nxtobs", "rewards", "terminated", "truncated", "steps"), batch, ref):
            assert got.shape == want.shape and got.dtype == want.dtype, name
            torch.testing.assert_close(got.cpu(), want, rtol=0, atol=0, msg=name)


@pytest.mark.gpu
def test_device_store_on_gpu():
    test_device_store_batches_equal_dataset_items()


def test_datamodule_curriculum_window():
    from pdecontrol.surrogates.common.datamodule import PDEDataModule
    _, rp = sc.run(ExperienceReplay, ds, sched, Sample)
    cur = sched.LinearScheduler(steptype="iteration", start=0, stop=4, vmin=1, vmax=3)
    dm = PDEDataModule(rp.data, train=rp.episodes, val=rp.episodes[:1], test=rp.episodes[:1], bootstrapping=False,
                       curriculum=cur, iteration=2, tau=2, target_length=2, batch_size=4)

    class T:
        current_epoch, global_step = 0, 0
    dm.trainer = T()
    b = next(iter(dm.train_dataloader()))
    assert b[0].shape == (4, 2 + 2, 1, 8)       # tau + K(iteration=2) = 2 + 2
    assert next(iter(dm.test_dataloader()))[0].shape[1] == 4
    dm.iteration = 4
    assert next(iter(dm.val_dataloader()))[0].shape[1] == 2 + 3


def test_locate_many_is_locate():
    _, rp = sc.run(ExperienceReplay, ds, sched, Sample)
    for bootstrapping, stride in ((False, 2), (False, None), (True, 1)):
        np.random.seed(5)
        d = ds.SubSeqDataset(rp.data, length=3, stride=stride, bootstrapping=bootstrapping, bounds=(1, 0))
        idx = np.arange(len(d))
        keys, starts = d.locate_many(idx)
        one_by_one = [d.locate(int(i)) for i in idx]
        assert keys == [k for k, _ in one_by_one]
        np.testing.assert_array_equal(starts, [s for _, s in one_by_one])
    assert d.locate_many([])[0] == []


def _datamodule_batches(device_data):
    from pdecontrol.surrogates.common.datamodule import PDEDataModule
    from pdegym.common.transforms import BatchTransform, Normalize, SampleTransform
    _, rp = sc.run(ExperienceReplay, ds, sched, Sample)
    norm = Normalize(aggregate=True, batched=True)
    norm.mean, norm.var = torch.full((1, 1, 1), 0.2), torch.full((1, 1, 1), 1.5)
    dm = PDEDataModule(rp.data, train=rp.episodes, val=rp.episodes[:1], test=rp.episodes[:1], bootstrapping=True,
                       stransf=SampleTransform(otransf=BatchTransform(norm)), tau=2, target_length=2, batch_size=3,
                       device_data=device_data)

    class T:
        current_epoch, global_step = 0, 0
    dm.trainer = T()
    out = []
    for make in (dm.train_dataloader, dm.val_dataloader, dm.test_dataloader):
        np.random.seed(9)                      # the bootstrap mapping is drawn when the dataset is built
        out.append([[t.cpu() for t in batch] for batch in make()])
    return out


def check_device_loader(device):
    host, dev = _datamodule_batches(None), _datamodule_batches(device)
    for h_loader, d_loader in zip(host, dev):
        assert len(h_loader) == len(d_loader) > 0
        for hb, db in zip(h_loader, d_loader):       # incl. the ragged last batch
            assert len(hb) == len(db) == 7
            for h, d in zip(hb, db):
                assert h.shape == d.shape and h.dtype == d.dtype
                torch.testing.assert_close(d, h, rtol=0, atol=0)


def test_device_batch_loader_equals_host_loader():
    """PDEDataModule(device_data=...) yields the batches the reference-style host loader yields (same windows, order,
    dtypes, transform applied), here with the store on the CPU device."""
    check_device_loader("cpu")


@pytest.mark.gpu
def test_device_batch_loader_on_gpu():
    check_device_loader("cuda:0")
