"""Episodic experience replay (mirror of the reference's ``pdecontrol/mbrl/replay.py:9-149``).

One deque of per-step arrays per episode and field; ``vindex`` maps a sub-environment id to the
episode it is currently writing, and moves to a fresh episode id when that env's step terminated or
truncated.  ``data`` is the 7-tuple-like ``Sample`` of the dicts that the datasets index
(pdecontrol/surrogates/common/dataset.py).
"""
from collections import defaultdict, deque
from typing import List

import numpy as np

from pdecontrol.mbrl.types import Sample

_FIELDS = ("obs", "actions", "nxtobs", "rewards", "terminated", "truncated", "steps")
_DTYPES = (np.float32, np.float32, np.float32, np.float32, np.bool_, np.bool_, np.int32)


class ExperienceReplay:
    def __init__(self, capacity: int = None):
        self.capacity = np.inf if capacity is None else capacity
        for name in _FIELDS:
            setattr(self, name, defaultdict(deque))
        self.data = Sample(*(getattr(self, name) for name in _FIELDS))
        # episode id each sub-environment currently writes to; a new sub-env gets the next free id
        self.vindex = defaultdict(self._next_episode_id)

    def _next_episode_id(self):
        return max(self.vindex.values(), default=-1) + 1

    def _stores(self):
        return [getattr(self, name) for name in _FIELDS]

    def add(self, samples: List[Sample], stransf=None):
        if stransf is not None:
            samples = [stransf(sample) for sample in samples]
        for vid, sample in enumerate(samples):
            vpos = self.vindex[vid]
            if sample is None:
                continue
            if stransf is not None:  # (sic) the reference applies the transform a second time here
                sample = stransf(sample)
            for store, value in zip(self._stores(), sample):
                store[vpos].append(value)
            if sample.terminated or sample.truncated:
                self.vindex[vid] = self._next_episode_id()
        self.resize(self.capacity)

    def extend(self, replay):
        for vid, ep in enumerate(sorted(replay.episodes)):
            vid = vid % len(replay.vindex)
            vpos = self.vindex[vid]
            for mine, theirs in zip(self._stores(), replay._stores()):
                mine[vpos].extend(theirs[ep].copy())
            if np.any(self.terminated[vpos]) or np.any(self.truncated[vpos]):
                self.vindex[vid] = self._next_episode_id()
        self.resize(self.capacity)

    def sample(self, index: int = None, stransf=None):
        index = np.random.choice(self.episodes) if index is None else index
        sample = Sample(*(np.asarray(store[index], dtype=dt) for store, dt in zip(self._stores(), _DTYPES)))
        if stransf is not None:
            sample = stransf(sample)
        return sample.totorch()

    def resize(self, size):
        """Drop the oldest episodes until at most ``size`` time steps remain."""
        self.capacity = size
        while self.ntimesteps > self.capacity:
            oldest = min(self.obs.keys())
            for store in self._stores():
                store.pop(oldest)

    def statistics(self):
        returns = [sum(self.sample(ep).tonumpy().rewards) for ep in self.stopped]
        return np.mean(returns), np.std(returns)

    def dataset(self):
        flat = lambda store: np.asarray([v for seq in store.values() for v in seq], dtype=np.float32)
        return Sample(*(flat(store) for store in self._stores()))

    @property
    def stopped(self):
        return [idx for idx in self.episodes if bool(self.truncated[idx][-1])]

    @property
    def nstopped(self):
        return len(self.stopped)

    @property
    def episodes(self):
        return list(self.obs.keys())

    @property
    def nepisodes(self):
        return len(self.episodes)

    @property
    def ntimesteps(self):
        return sum(len(seq) for seq in self.obs.values())
