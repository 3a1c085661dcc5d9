"""Initial conditions of many envs at once, bit-identical to the reference's per-env draws.

The reference seeds NumPy's legacy global generator in ``reset`` and draws ``np.random.uniform(-0.4, 0.4, N)``
(pdegym/kuramoto/kuramoto.py:101,106); gym's vector convention gives env i the seed ``seed + i``, and later autoresets of
an env continue that env's stream.  One ``np.random.RandomState`` per env reproduces that, at ~45 us per env and draw: 0.2 s
for a full reset of 4096 envs and 1.5 s for 32768 -- more than the 0.6 s the whole 200 000-sub-step burn-in takes on the
GPU.  ``BatchedMT19937`` keeps the MT19937 state of every env in one ``[E, 624]`` array and runs seeding (``init_genrand``),
the twist, the tempering and the 53-bit double conversion as NumPy array operations ACROSS envs: same bits
(tests/test_mt_batch.py compares against ``RandomState`` across twist boundaries and mixed stream positions), ~100x less
host time.

Seeds must lie in [0, 2**32) as for ``RandomState``; ``None`` (OS entropy in the reference: nothing to reproduce) draws the
32-bit seed from ``SeedSequence``.
"""
import numpy as np

_N, _M = 624, 397
_UPPER, _LOWER, _MATRIX = np.uint32(0x80000000), np.uint32(0x7FFFFFFF), np.uint32(0x9908B0DF)


def _mix(upper_src, lower_src):
    y = (upper_src & _UPPER) | (lower_src & _LOWER)
    return (y >> np.uint32(1)) ^ ((y & np.uint32(1)) * _MATRIX)


class BatchedMT19937:
    """State is word-major, ``[624, E]``: the recurrences walk the words, every step is one contiguous array operation over
    the envs."""

    def __init__(self, num_envs):
        self.E = int(num_envs)
        self.state = np.zeros((_N, self.E), dtype=np.uint32)
        self.pos = np.full(self.E, _N, dtype=np.int64)        # index of the next word; _N: a twist is due
        self.seed_rows(np.arange(self.E), [None] * self.E)

    # -- seeding --------------------------------------------------------------------------------------------------
    def seed_rows(self, ids, seeds):
        """env ids[k] <- RandomState(seeds[k]) (None: fresh entropy)."""
        ids = np.asarray(ids, dtype=np.int64)
        vals = np.zeros(len(ids), dtype=np.uint32)
        n_none = sum(s is None for s in seeds)
        entropy = iter(np.random.SeedSequence().generate_state(n_none).tolist()) if n_none else iter(())
        for k, s in enumerate(seeds):
            if s is None:
                vals[k] = next(entropy)
            elif 0 <= int(s) < 2 ** 32:
                vals[k] = int(s)
            else:
                raise ValueError("Seed must be between 0 and 2**32 - 1")
        # init_genrand: mt[0] = s; mt[i] = 1812433253 * (mt[i-1] ^ (mt[i-1] >> 30)) + i   (mod 2**32), all envs per step
        mt = np.empty((_N, len(ids)), dtype=np.uint32)
        mt[0] = vals
        prev = vals.astype(np.uint64)
        mult, mask, sh = np.uint64(1812433253), np.uint64(0xFFFFFFFF), np.uint64(30)
        for i in range(1, _N):
            prev = (mult * (prev ^ (prev >> sh)) + np.uint64(i)) & mask
            mt[i] = prev
        if len(ids) == self.E and (ids == np.arange(self.E)).all():
            self.state = mt
        else:
            self.state[:, ids] = mt
        self.pos[ids] = _N

    # -- generation -----------------------------------------------------------------------------------------------
    @staticmethod
    def _twisted(mt):
        """The next 624 words of every column of ``mt`` [624, n]."""
        new = np.empty_like(mt)
        new[:227] = mt[397:] ^ _mix(mt[:227], mt[1:228])
        new[227:454] = new[:227] ^ _mix(mt[227:454], mt[228:455])
        new[454:623] = new[227:396] ^ _mix(mt[454:623], mt[455:624])
        new[623] = new[396] ^ _mix(mt[623], new[0])
        return new

    @staticmethod
    def _temper(y):
        y = y ^ (y >> np.uint32(11))
        y = y ^ ((y << np.uint32(7)) & np.uint32(0x9D2C5680))
        y = y ^ ((y << np.uint32(15)) & np.uint32(0xEFC60000))
        return y ^ (y >> np.uint32(18))

    def _words(self, ids, count):
        """count tempered 32-bit outputs of every listed env -> [count, len(ids)] uint32 (word-major)."""
        ids = np.asarray(ids, dtype=np.int64)
        everyone = len(ids) == self.E and (ids == np.arange(self.E)).all()
        pos = self.pos[ids]
        if (pos == pos[0]).all():
            # every listed env at the same stream position (a reset of all envs, or envs that have always been reset
            # together): whole blocks of words at a time
            st = self.state if everyone else self.state[:, ids]
            p, parts, need = int(pos[0]), [], count
            while need > 0:
                if p >= _N:
                    st, p = self._twisted(st), 0
                take = min(need, _N - p)
                parts.append(st[p:p + take])
                p += take
                need -= take
            if everyone:
                self.state = st
            else:
                self.state[:, ids] = st
            self.pos[ids] = p
            return self._temper(parts[0] if len(parts) == 1 else np.concatenate(parts, axis=0))
        # mixed positions: group the envs by position (few distinct values in practice)
        out = np.empty((count, len(ids)), dtype=np.uint32)
        for value in np.unique(pos):
            sel = np.nonzero(pos == value)[0]
            out[:, sel] = self._words(ids[sel], count)
        return out

    def uniform_rows(self, ids, low, high, n):
        """[len(ids), n] float64: what ``RandomState.uniform(low, high, n)`` of every listed env returns next."""
        if len(ids) == 0:
            return np.empty((0, n), dtype=np.float64)
        w = self._words(ids, 2 * n)
        a = (w[0::2] >> np.uint32(5)).astype(np.float64)
        b = (w[1::2] >> np.uint32(6)).astype(np.float64)
        d = (a * 67108864.0 + b) / 9007199254740992.0
        return np.ascontiguousarray((low + (high - low) * d).T)
