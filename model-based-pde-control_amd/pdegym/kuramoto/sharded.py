"""KSShardedVecEnv -- ONE controller process, several GPUs.

The reference's controller is a single process that calls ``gym.vector.make(env_id, num_envs=cpus)`` once
(pdecontrol/mbrl/mbrl.py:81-86) and gets one subprocess per env.  Here the same call returns one vector env whose envs
live in contiguous blocks on the listed devices (``shard_envs``: device g owns envs [lo_g, hi_g)), one ``ks_handle`` +
stream per device.  Envs never interact, so there is no collective: per step every shard's launch is
enqueued through the split host-boundary entry (``ks_step_begin``: pinned staging, launch, copy into the pinned mirror,
no wait) BEFORE any shard is waited for, then one host thread per shard collects its block (``ks_step_end``: wait +
copy into that shard's rows of the ``[E, 1, N]`` result) -- ctypes releases the GIL, so the waits and host copies of the
shards overlap.  Autoreset and ``reset`` run the burn-in of every shard's rows the same way.

``step_async`` really starts the step here (gym's contract for it); ``step_wait`` collects.

A device may be listed more than once (``devices=[0, 0]``: two handles and streams on one GPU -- how the single-GPU test
checks this class bit for bit against ``KSBatchedVecEnv``), and ``-1`` is the library's CPU twin.
"""
from concurrent.futures import ThreadPoolExecutor
from typing import Optional, Sequence

import numpy as np

from pdegym.kuramoto.batched import KSBatchedVecEnv, shard_envs


def resolve_devices(spec):
    """``"all"`` -> every visible HIP device; ``"0,2,3"`` / a sequence -> those ordinals (``cpu`` = -1)."""
    if isinstance(spec, str):
        if spec.strip().lower() == "all":
            import torch
            n = torch.cuda.device_count()
            if n < 1:
                raise RuntimeError("PDEGYM_DEVICES=all: no HIP device is visible")
            return list(range(n))
        spec = [t for t in spec.replace(";", ",").split(",") if t.strip()]
    out = [(-1 if str(d).strip().lower() == "cpu" else int(d)) for d in spec]
    if not out:
        raise ValueError("empty device list")
    return out


class KSShardedVecEnv(KSBatchedVecEnv):
    #: observation bytes per shard from which the shards are collected on one host thread each (the host copy out of the
    #: pinned mirror is what the threads parallelise)
    THREADS_FROM_BYTES = 512 * 1024

    def __init__(self, num_envs: int, config: Optional[dict] = None, devices: Sequence[int] = (0,), **kwargs):
        self.devices = resolve_devices(devices)
        if num_envs < len(self.devices):
            self.devices = self.devices[:num_envs]
        kwargs.pop("device", None)
        super().__init__(num_envs, config=config, device=self.devices[0], **kwargs)

    def _build_steppers(self, stepper_cls, variant):
        self.shards = []                       # (lo, hi, stepper)
        F = self.proto.forcing.forcing.numpy()
        for g, dev in enumerate(self.devices):
            lo, hi = shard_envs(self.num_envs, g, len(self.devices))
            st = stepper_cls(hi - lo, self.N, self.L, self.dt, device=dev, mode=self.step_mode, variant=variant)
            st.set_forcing(F)
            self.shards.append((lo, hi, st))
        self.stepper = None                    # there is no single stepper
        self._pool = ThreadPoolExecutor(max_workers=len(self.shards), thread_name_prefix="ks-shard")
        self._in_flight = False

    # -- helpers -----------------------------------------------------------------------------
    def _set_mode(self, mode):
        for _, _, st in self.shards:
            if st.mode != mode:
                st.set_mode(mode)

    def _collect(self, jobs):
        """jobs: [(stepper, (obs, ssq, status) slices)] already begun; wait for all of them on the shard threads."""
        small = jobs and jobs[0][1][0] is not None and jobs[0][1][0].nbytes <= self.THREADS_FROM_BYTES
        if len(jobs) == 1 or small:
            # small blocks: waiting for the shards one after the other costs less than handing them to threads (1024 x 64
            # on four handles: 1.9 ms per step through the pool, 0.3 ms in sequence); the devices run concurrently either
            # way -- every launch was enqueued before the first wait
            err = None
            for st, out in jobs:
                try:
                    st.step_end(out)
                except Exception as exc:       # noqa: BLE001
                    err = err or exc
            if err is not None:
                raise err
            return
        futs = [self._pool.submit(st.step_end, out) for st, out in jobs]
        err = None
        for f in futs:                          # every shard is collected even if one failed: no handle stays pending
            try:
                f.result()
            except Exception as exc:           # noqa: BLE001
                err = err or exc
        if err is not None:
            raise err

    def _fresh_rows(self, ids):
        ids = np.asarray(ids, dtype=np.int64)
        n = len(ids)
        u0 = self._rng.uniform_rows(ids, -0.4, 0.4, self.N)
        obs = np.empty((n, self.N), dtype=np.float32)
        ssq = np.empty(n, dtype=np.float64)
        status = np.empty(n, dtype=np.int32)
        self._set_mode(self.reset_mode)
        jobs, parts = [], []
        for lo, hi, st in self.shards:
            sel = np.nonzero((ids >= lo) & (ids < hi))[0]
            if len(sel) == 0:
                continue
            local = (ids[sel] - lo).astype(np.int32)
            st.set_state_rows(local, u0[sel])
            st.step_begin(None, local, self.burn_in_substeps, want_obs=True)     # every shard's burn-in is enqueued ...
            part = (np.empty((len(sel), self.N), np.float32), np.empty(len(sel)), np.empty(len(sel), np.int32))
            jobs.append((st, part))
            parts.append((sel, part))
        self._collect(jobs)                                                         # ... before any is waited for
        for sel, (o, q, s_) in parts:
            obs[sel], ssq[sel], status[sel] = o, q, s_
        self._set_mode(self.step_mode)
        self._raise_on(status)
        self.timestep[ids] = 0
        return obs

    # -- gym.vector API ----------------------------------------------------------------------
    def step_async(self, actions):
        assert not self._in_flight, "step_async() twice without step_wait()"
        a = np.ascontiguousarray(np.asarray(actions, dtype=np.float32).reshape(self.num_envs, -1))
        for lo, hi, st in self.shards:
            st.step_begin(a[lo:hi], None, self.cfg_steps, want_obs=True)
        self._in_flight = True

    def step_wait(self, **kwargs):
        assert self._in_flight, "step_wait() without step_async()"
        obs = np.empty((self.num_envs, self.N), dtype=np.float32)
        ssq = np.empty(self.num_envs, dtype=np.float64)
        status = np.empty(self.num_envs, dtype=np.int32)
        self._in_flight = False
        self._collect([(st, (obs[lo:hi], ssq[lo:hi], status[lo:hi])) for lo, hi, st in self.shards])
        return self._finish_step(obs, ssq, status)

    def step_torch(self, actions):
        raise NotImplementedError("device-resident stepping is per GPU: build one KSBatchedVecEnv per device for that")

    def close_extras(self, **kwargs):
        for _, _, st in self.shards:
            st.close()
        self._pool.shutdown(wait=True)
