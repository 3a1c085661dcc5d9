"""Sub-sequence sampling over the episodic replay.

API / index-rule mirror of the reference's ``pdecontrol/surrogates/common/dataset.py``
(``SubSeqDataset`` :16-116, ``StartingStateDataset`` :119-160, ``PDEDataLoader`` :163-205,
``ResampleDataLoader`` :208-227).  The integer path -- how a flat item index maps to
(episode, start offset) with or without bootstrap resampling -- is ``SubSeqDataset.locate`` and is
bit-exact with the reference (pinned by tests/golden/dataset_golden.npz).

MI355X-native addition: ``DeviceSubSeqStore`` packs the replay ONCE into contiguous HBM tensors and
assembles whole batches with a single gather per field from the very same (episode, offset) pairs,
instead of ``islice`` + ``np.asarray`` + per-item transforms on the host for every item.
"""
import bisect
import math
from collections import defaultdict
from itertools import islice
from typing import List, Tuple

import numpy as np
import torch
from torch.utils.data import ConcatDataset, DataLoader, Dataset
from torch.utils.data.dataloader import default_collate

from pdecontrol.mbrl.types import Sample

_DTYPES = (np.float32, np.float32, np.float32, np.float32, np.bool_, np.bool_, np.int32)


class SubSeqDataset(Dataset):
    """All windows of ``length`` steps (hop ``stride``) inside each episode of ``subsamples``;
    with ``bootstrapping`` the i-th item is a window drawn (once, at construction, with
    ``np.random.randint``) uniformly among ALL hop-1 windows."""

    def __init__(self, data: Tuple, subsamples: List[int] = None, length: int = 1, stride: int = None,
                 bootstrapping: bool = True, bounds: Tuple[int, int] = (0, 0), stransf=None):
        super().__init__()
        self.fields = tuple(data)
        (self.obs, self.actions, self.nxtobs, self.rewards, self.terminated, self.truncated, self.steps) = self.fields
        if subsamples is None:
            subsamples = list(self.obs.keys()) if isinstance(self.obs, defaultdict) else list(np.arange(self.obs.shape[0]))
        self.subsamples, self.length, self.bootstrapping, self.stransf = subsamples, length, bootstrapping, stransf
        self.stride = length if stride is None else stride  # default: non-overlapping windows
        self.lower, self.upper = bounds
        self.index = np.cumsum(self.count_sub_seqs(self.length, self.stride))
        self.boots_index = np.cumsum(self.count_sub_seqs(self.length, 1))
        self.boots_mapping = np.random.randint(low=0, high=np.max(self.boots_index, initial=0).astype(np.int32),
                                               size=len(self))

    # -- the integer path ---------------------------------------------------------------------
    def locate(self, idx):
        """item index -> (episode key, first step of the window)."""
        assert idx < len(self)
        if self.bootstrapping:
            idx, index, stride = self.boots_mapping[idx], self.boots_index, 1
        else:
            index, stride = self.index, self.stride
        key = bisect.bisect_right(index, idx)
        offset = index[key - 1] if key - 1 >= 0 else 0
        return self.subsamples[key], (idx - offset) * stride + self.lower

    def locate_many(self, idxs):
        """``locate`` for a whole batch of item indices at once (numpy ``searchsorted(side="right")`` is ``bisect_right``):
        returns (episode keys [list], first steps [int64 array])."""
        idxs = np.asarray(idxs, dtype=np.int64)
        assert idxs.size == 0 or idxs.max() < len(self)
        if self.bootstrapping:
            idxs, index, stride = self.boots_mapping[idxs].astype(np.int64), self.boots_index, 1
        else:
            index, stride = self.index, self.stride
        index = np.asarray(index, dtype=np.int64)
        keys = np.searchsorted(index, idxs, side="right")
        offsets = np.where(keys > 0, index[np.maximum(keys - 1, 0)], 0)
        return [self.subsamples[k] for k in keys], (idxs - offsets) * stride + self.lower

    def __getitem__(self, idx):
        bidx, sidx = self.locate(idx)
        window = lambda store, dt: np.asarray(list(islice(store[bidx], sidx, sidx + self.length)), dtype=dt)
        sample = Sample(*(window(store, dt) for store, dt in zip(self.fields, _DTYPES)))
        if self.stransf:
            sample = self.stransf(sample)
        return sample.totorch()

    def __len__(self):
        return np.max(self.index, initial=0).astype(np.int32)

    def _usable(self, key):
        return len(self.obs[key]) - self.lower - self.upper

    def count_sub_seqs(self, length, stride):
        return [self.count_seq_sub_seqs(self._usable(key), length, stride) for key in self.subsamples]

    def count_seq_sub_seqs(self, nelems, length, stride):
        return max(math.floor((nelems - length) / stride) + 1, 0)

    @property
    def max_seq_length(self):
        return max(self._usable(key) for key in self.subsamples)


class StartingStateDataset(ConcatDataset):
    """Warm-up windows for imagined rollouts: full-length windows anywhere, plus the shorter windows
    (1 .. length steps) that start at the very beginning of an episode."""

    def __init__(self, data: Tuple, subsamples: List[int] = None, length: int = 1, stride: int = None,
                 bootstrapping: bool = False, bounds: Tuple[int, int] = (0, 0), stransf=None):
        full = SubSeqDataset(data=data, subsamples=subsamples, length=length, stride=stride,
                             bootstrapping=bootstrapping, bounds=bounds, stransf=stransf)
        lower, upper = bounds
        parts = [full]
        for short in range(1, length + 1):
            parts.append(SubSeqDataset(data=data, subsamples=subsamples, length=short, stride=length - short + 1,
                                       bootstrapping=bootstrapping,
                                       bounds=(lower, upper + full.max_seq_length - short), stransf=stransf))
        super().__init__(parts)


class PDEDataLoader(DataLoader):
    @staticmethod
    def sample_collate(samples):
        return default_collate([tuple(sample) for sample in samples])

    @staticmethod
    def repeat_padding(tensors, dim=0):
        """Left-pad every tensor to the longest one by repeating its first slice."""
        longest = max(t.size(dim) for t in tensors)
        padded = []
        for t in tensors:
            first = torch.index_select(t, dim=dim, index=torch.as_tensor(0))
            padded.append(torch.cat((torch.repeat_interleave(first, longest - t.size(dim), dim=dim), t), dim=dim))
        return torch.stack(padded)

    @staticmethod
    def padding_collate(samples):
        columns = zip(*(tuple(sample) for sample in samples))
        return Sample(*(PDEDataLoader.repeat_padding(list(col), dim=0) for col in columns))


class ResampleDataLoader(DataLoader):
    """Endless loader: restarts its iterator when the dataset is exhausted."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.iterator = super().__iter__()

    def __iter__(self):
        return self

    def __next__(self):
        try:
            return next(self.iterator)
        except StopIteration:
            self.iterator = super().__iter__()
            return next(self.iterator)


class DeviceSubSeqStore:
    """The replay packed into contiguous device tensors + batched window gathers.

    ``pack`` concatenates every episode of ``data`` along time into one tensor per field
    (obs ``[sum_T, ...]`` fp32 etc.) with ``starts[key]`` the offset of each episode.  ``batch`` takes
    the (episode, first-step) pairs of ``SubSeqDataset.locate`` for a whole batch and gathers
    ``[B, length, ...]`` per field with one ``index_select`` -- the same windows, in the same order,
    as iterating the dataset, but assembled in HBM.  An optional ``stransf`` (SampleTransform) is
    applied to the assembled device batch (its transforms are device-aware and batch-vectorised).
    """

    def __init__(self, data, device, keys=None):
        fields = tuple(data)
        keys = list(fields[0].keys()) if keys is None else list(keys)
        self.device = torch.device(device)
        self.starts, off = {}, 0
        for k in keys:
            self.starts[k] = off
            off += len(fields[0][k])
        self.total = off
        pack = lambda store, dt: torch.from_numpy(
            np.concatenate([np.asarray(store[k], dtype=dt).reshape(len(store[k]), *np.shape(store[k][0])) for k in keys])
        ).to(self.device)
        self.tensors = tuple(pack(store, dt) for store, dt in zip(fields, _DTYPES))

    def batch(self, dataset: SubSeqDataset, indices, stransf=None):
        keys, starts = dataset.locate_many(indices)
        first = np.asarray([self.starts[k] for k in keys], dtype=np.int64) + starts
        rows = torch.from_numpy((first[:, None] + np.arange(dataset.length)[None, :]).reshape(-1)).to(self.device)
        shape = (len(keys), dataset.length)
        out = [t.index_select(0, rows).reshape(shape + tuple(t.shape[1:])) for t in self.tensors]
        out[6] = out[6].to(torch.int32)
        sample = Sample(*out)
        return stransf(sample) if stransf is not None else sample


class DeviceBatchLoader:
    """What ``PDEDataLoader(dataset, batch_size, shuffle=False, collate_fn=PDEDataLoader.sample_collate)`` yields -- the
    loader the reference's datamodule builds (pdecontrol/surrogates/common/datamodule.py:66-72) -- assembled in HBM: the
    same windows in the same order (the index stream of an unshuffled loader is 0, 1, 2, ...; the bootstrap resampling
    lives in the dataset), one gather per field per batch from the packed replay, the dataset's ``stransf`` applied to
    the whole device batch.  Batches are lists of device tensors in ``Sample`` field order, like ``default_collate``'s."""

    def __init__(self, dataset: SubSeqDataset, store: DeviceSubSeqStore, batch_size: int, drop_last: bool = False):
        self.dataset, self.store, self.batch_size, self.drop_last = dataset, store, int(batch_size), drop_last

    def __len__(self):
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = int(len(self.dataset))
        for i0 in range(0, n, self.batch_size):
            i1 = min(i0 + self.batch_size, n)
            if self.drop_last and i1 - i0 < self.batch_size:
                return
            yield list(self.store.batch(self.dataset, np.arange(i0, i1), stransf=self.dataset.stransf))
