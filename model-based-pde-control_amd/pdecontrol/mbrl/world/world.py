"""Surrogate-backed vector environment for imagined rollouts.

API / behaviour mirror of the reference's ``pdecontrol/mbrl/world/world.py`` (``BaseWorldVecEnv``
:15-65, ``WorldVecEnv`` :68-204): ``reset`` samples warm-up windows from the replay and rolls the
surrogate (ensemble) over them, ``step_async`` advances every imagined trajectory by one surrogate
step under ``no_grad`` and evaluates the reward, ``step_wait`` truncates ALL trajectories together at
the rollout horizon / env time limit and restarts them.

Device handling (the reference is CPU-only): tensors follow the surrogate's parameters -- the warm-up
batch and the actions are moved to that device, observations / rewards come back as numpy at the gym
boundary.  ``batched_reward_func`` (optional) evaluates the reward for the whole batch at once; the
reference's per-sample Python loop (``world.py:170``) remains the default so arbitrary reward callables
keep working.

Device-resident mode (surrogate on a GPU, fused kernels on, a ``batched_reward_func`` given -- the way
``pdegym.kuramoto`` envs provide one): the imagined state, every ensemble member's hidden state and the
replay the warm-up windows come from all live in HBM.  One ``step`` is: ONE pinned H2D copy (actions + the
host-drawn elite choice), ONE hipGraph replay (every member's fused one-step rollout + the per-row elite
gather, state and hidden states updated in place), the reward on the device batch, ONE D2H copy
(observations + rewards packed).  ``reset`` gathers its warm-up windows from the device store with the
reference's own index stream (same ``RandomSampler`` draws, same padding rule) instead of per-item host
collation.  Values are those of the host path (same kernels as the fused rollout; tests/test_world_env.py).
"""
from typing import Any, Callable, Sequence

import os

import numpy as np
import torch
from torch.utils.data import Dataset, RandomSampler

from pdegym._gym import gym
from pdecontrol.mbrl.types import ModelRollout, Sample
from pdecontrol.surrogates.common.dataset import DeviceSubSeqStore, PDEDataLoader

try:  # real gym
    from gym.vector.utils.spaces import batch_space  # type: ignore
except ImportError:
    from pdegym._compat.gym_shim import batch_space


def _surrogate_device(surrogate):
    modules = getattr(surrogate, "modules", None)
    if isinstance(modules, (list, tuple)) and modules:     # PDEEnsemble keeps a plain list of training modules
        return next(modules[0].surrogate.parameters()).device
    return next(surrogate.parameters()).device


class BaseWorldVecEnv(gym.vector.VectorEnv):
    def __init__(self, surrogate, observation_space, action_space, max_episode_steps: int, stransf,
                 reward_func: Callable, num_envs: int, horizon: int, tstep: float):
        self.surrogate, self.max_episode_steps, self.stransf = surrogate, max_episode_steps, stransf
        self.reward_func, self.num_envs, self.horizon, self.tstep = reward_func, num_envs, horizon, tstep
        self.observation_space = gym.spaces.Box(observation_space.low, observation_space.high)
        self.action_space = gym.spaces.Box(action_space.low, action_space.high)
        self.is_vector_env = True

    def setup(self, starting: Dataset):
        """Endless stream of warm-up batches, sampled with replacement from ``starting``."""
        sampler = RandomSampler(starting, replacement=True, num_samples=int(1e10))
        self.loader = iter(PDEDataLoader(starting, batch_size=self.num_envs, shuffle=False, sampler=sampler,
                                         drop_last=True, collate_fn=PDEDataLoader.padding_collate))


def _members(surrogate):
    """[(rollout-capable surrogate), ...], ensemble or None."""
    modules = getattr(surrogate, "modules", None)
    if isinstance(modules, (list, tuple)) and modules:
        return [m.surrogate for m in modules], surrogate
    return [surrogate], None


class _DeviceWorldState:
    """HBM-resident state of the imagined trajectories + the captured ensemble step (see module docstring)."""

    def __init__(self, world, outputs, hiddens):
        from pdecontrol.surrogates.graph_step import capture_graph
        from pdecontrol.surrogates.hipops import pooled_streams
        self.world = world
        self.members, self.ensemble = _members(world.surrogate)
        self.signature = self._signature()
        dev = outputs.device
        self.device = dev
        b = outputs.shape[0]
        self.b, self.n = b, outputs.shape[-1]
        self.state = outputs.detach().clone()                                  # [B, 1, 1, N]
        self.hidden = [tuple(h.detach().clone() for h in hid) for hid in hiddens]
        act_shape = tuple(world.single_action_space.shape)
        self.act_elems = int(np.prod(act_shape))
        # one pinned staging buffer each way: [actions | chosen member per row] down, [obs | reward] up
        self.h2d_host = torch.empty(b * (self.act_elems + 1), dtype=torch.float32).pin_memory()
        self.h2d_dev = torch.empty_like(self.h2d_host, device=dev)
        self.act = self.h2d_dev[:b * self.act_elems].view((b, 1) + act_shape)      # [B, 1, C, A]
        self.chosen_f = self.h2d_dev[b * self.act_elems:]
        self.d2h_dev = torch.empty((b, self.n + 1), dtype=torch.float32, device=dev)
        self.d2h_host = torch.empty((b, self.n + 1), dtype=torch.float32).pin_memory()
        self.rows = torch.arange(b, device=dev)
        self.graph = torch.cuda.CUDAGraph()
        (stream,) = pooled_streams(dev, 1, "capture")
        keep = (self.state.clone(), [tuple(h.clone() for h in hid) for hid in self.hidden])
        self.h2d_dev.zero_()
        stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(stream), torch.no_grad():
            for _ in range(2):                                                 # warm-up (kernels, allocator)
                self._advance()
        capture_graph(self.graph, self._advance_nograd, stream)
        torch.cuda.current_stream(dev).wait_stream(stream)
        self.state.copy_(keep[0])
        for hid, saved in zip(self.hidden, keep[1]):
            for h, sv in zip(hid, saved):
                h.copy_(sv)

    def _signature(self):
        """What the captured launches carry by value: the members' parameter addresses and delta-scaling constants."""
        from pdecontrol.surrogates import hipops
        return [(next(m.parameters()).data_ptr(), hipops.scaling_signature(m)) for m in self.members]

    def valid(self):
        from pdecontrol.surrogates import hipops
        now = self._signature()
        return len(now) == len(self.signature) and all(a[0] == b[0] and hipops.same_signature(a[1], b[1])
                                                       for a, b in zip(self.signature, now))

    def _advance_nograd(self):
        with torch.no_grad():
            self._advance()

    def _advance(self):
        """Every member one step from the shared state, per-row pick of the chosen member, in-place update."""
        tstep = self.world.tstep
        outs = []
        for sur, hid in zip(self.members, self.hidden):
            r = sur.rollout(states=self.state, actions=self.act, times=0.0, targets=tstep, hidden=hid)
            outs.append(r.outputs)
            for h, new in zip(hid, r.hidden):
                h.copy_(new)
        if len(outs) == 1:
            new_state = outs[0]
        else:
            new_state = torch.stack(outs, dim=0)[self.chosen_f.to(torch.long), self.rows]
        self.state.copy_(new_state)

    def load(self, outputs, hiddens):
        self.state.copy_(outputs)
        for hid, new in zip(self.hidden, hiddens):
            for h, t in zip(hid, new):
                h.copy_(t)

    # -- the reset's warm-up rollout as a second captured graph -------------------------------------------------
    def capture_reset(self, states, actions):
        """Capture [every member's warm-up rollout over the given window | per-row pick | in-place state update] for
        windows of this shape (reference: world.py:113-131 runs it eagerly at every reset, i.e. every ``horizon`` steps).
        The trajectories' current state and hidden state are preserved."""
        from pdecontrol.surrogates.graph_step import capture_graph
        from pdecontrol.surrogates.hipops import pooled_streams
        dev, tstep = self.device, self.world.tstep
        self.rs_states, self.rs_actions = states.detach().clone(), actions.detach().clone()
        times, targets = tstep * torch.arange(actions.size(1)), tstep * actions.size(1)

        def run():
            with torch.no_grad():
                outs = []
                for sur, hid in zip(self.members, self.hidden):
                    r = sur.rollout(states=self.rs_states, actions=self.rs_actions, hidden=None, times=times, targets=targets)
                    outs.append(r.outputs)
                    for h, new in zip(hid, r.hidden):
                        h.copy_(new)
                new_state = outs[0] if len(outs) == 1 else torch.stack(outs, dim=0)[self.chosen_f.to(torch.long), self.rows]
                self.state.copy_(new_state)

        keep = (self.state.clone(), [tuple(h.clone() for h in hid) for hid in self.hidden], self.h2d_dev.clone())
        (stream,) = pooled_streams(dev, 1, "capture")
        stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(stream):
            for _ in range(2):
                run()
        self.reset_graph = torch.cuda.CUDAGraph()
        capture_graph(self.reset_graph, run, stream)
        torch.cuda.current_stream(dev).wait_stream(stream)
        self.state.copy_(keep[0])
        for hid, saved in zip(self.hidden, keep[1]):
            for h, sv in zip(hid, saved):
                h.copy_(sv)
        self.h2d_dev.copy_(keep[2])

    def can_replay_reset(self, states, actions):
        return (getattr(self, "reset_graph", None) is not None and states.shape == self.rs_states.shape
                and actions.shape == self.rs_actions.shape and states.dtype == self.rs_states.dtype)

    def replay_reset(self, states, actions):
        """The captured warm-up rollout on a new window batch; the ensemble's per-row member draw is the one
        PDEEnsemble.rollout makes (surrogate.py:46), made here so the global NumPy stream advances identically."""
        b = self.b
        self.rs_states.copy_(states, non_blocking=True)
        self.rs_actions.copy_(actions, non_blocking=True)
        if self.ensemble is not None:
            chosen = np.random.choice(self.ensemble.elite_idx, size=b)
            self.h2d_host[b * self.act_elems:].copy_(torch.from_numpy(np.asarray(chosen, dtype=np.float32)))
            self.chosen_f.copy_(self.h2d_host[b * self.act_elems:], non_blocking=True)
        self.reset_graph.replay()

    def host_obs(self):
        return self.state.detach().squeeze(1).cpu().numpy()

    def step(self, host_actions):
        w, b = self.world, self.b
        self.h2d_host[:b * self.act_elems].copy_(torch.from_numpy(host_actions.reshape(-1)))
        if self.ensemble is not None:   # the same draw PDEEnsemble.rollout makes (surrogate.py:46), once per step
            chosen = np.random.choice(self.ensemble.elite_idx, size=b)
            self.h2d_host[b * self.act_elems:].copy_(torch.from_numpy(np.asarray(chosen, dtype=np.float32)))
        self.h2d_dev.copy_(self.h2d_host, non_blocking=True)
        self.graph.replay()
        with torch.no_grad():
            obs = self.state.squeeze(1)                                        # [B, 1, N] (C = 1)
            orescaled = w.stransf.otransf(obs)
            arescaled = w.stransf.atransf(self.act.squeeze(1))
            rewards = torch.as_tensor(w.batched_reward_func(orescaled, arescaled), device=self.device).reshape(b)
            self.d2h_dev[:, :self.n].copy_(obs.reshape(b, self.n))
            self.d2h_dev[:, self.n].copy_(rewards)
        self.d2h_host.copy_(self.d2h_dev, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        host = self.d2h_host.numpy()
        return host[:, :self.n].reshape(b, 1, self.n).copy(), host[:, self.n].copy()


class _DeviceStartingStates:
    """``StartingStateDataset`` windows gathered in HBM: the replay is packed once (DeviceSubSeqStore), every warm-up
    batch is one gather per field from the (episode, first step, length) triples of the reference's index rule,
    left-padded by repeating the first slice like ``PDEDataLoader.padding_collate``."""

    def __init__(self, starting, device, batch_size):
        self.ds = starting
        first = starting.datasets[0]
        self.length = first.length
        self.store = DeviceSubSeqStore(first.fields, device)
        self.steps_host = self.store.tensors[6].cpu().numpy()
        self.stransf = first.stransf
        self.device = device
        sampler = RandomSampler(starting, replacement=True, num_samples=int(1e10))
        # the DataLoader the reference builds draws its base seed from the global torch generator when its iterator is
        # created; same draw here so that the index stream that follows is the one the host loader would produce
        torch.empty((), dtype=torch.int64).random_()
        self.batches = iter(torch.utils.data.BatchSampler(sampler, batch_size, drop_last=True))
        self.cum = starting.cumulative_sizes

    def next_batch(self):
        idx = np.asarray(next(self.batches), dtype=np.int64)
        L = self.length
        cum = np.asarray(self.cum, dtype=np.int64)
        which = np.searchsorted(cum, idx, side="right")              # ConcatDataset's bisect_right, for the whole batch
        local = idx - np.where(which > 0, cum[np.maximum(which - 1, 0)], 0)
        rows = np.empty((len(idx), L), dtype=np.int64)
        for d in np.unique(which):                                     # at most length + 1 window lengths
            sel = which == d
            sub = self.ds.datasets[d]
            keys, starts = sub.locate_many(local[sel])
            first = np.fromiter((self.store.starts[k] for k in keys), dtype=np.int64, count=len(keys)) + starts
            pos = np.maximum(np.arange(L) - (L - sub.length), 0)      # left padding repeats the window's first step
            rows[sel] = first[:, None] + pos[None, :]
        flat = torch.from_numpy(rows.reshape(-1)).to(self.device)
        shape = (len(idx), L)
        out = [t.index_select(0, flat).reshape(shape + tuple(t.shape[1:])) for t in self.store.tensors]
        out[6] = out[6].to(torch.int32)
        sample = Sample(*out)
        if self.stransf is not None:
            sample = self.stransf(sample)
        return sample, self.steps_host[rows[:, -1]]


class WorldVecEnv(BaseWorldVecEnv):
    def __init__(self, surrogate, observation_space, action_space, max_episode_steps: int, stransf,
                 reward_func: Callable, num_envs: int, horizon: int, tstep: float, batched_reward_func=None,
                 device_resident=None):
        super().__init__(surrogate, observation_space, action_space, max_episode_steps, stransf, reward_func, num_envs,
                         horizon, tstep)
        self.batched_reward_func = batched_reward_func
        self.device_resident = device_resident     # None: decide at the first reset (GPU + fused kernels + batched reward)
        self.capture_reset = os.environ.get("PDECONTROL_WORLD_CAPTURE_RESET", "1") != "0"   # device path: reset = a graph replay
        self._dev = None
        self._dev_starting = None
        # spaces as seen through the replay->world transforms (stransf is their inverse)
        unbatch = lambda fn, x: np.squeeze(fn(x[np.newaxis, ...]), axis=0)
        low = unbatch(self.stransf.atransf.Inverse, action_space.low)
        high = unbatch(self.stransf.atransf.Inverse, action_space.high)
        self.single_action_space = gym.spaces.Box(low, high, shape=low.shape)
        self.action_space = batch_space(self.single_action_space, self.num_envs)
        obs = unbatch(self.stransf.otransf.Inverse, observation_space.sample())
        self.single_observation_space = gym.spaces.Box(-np.inf, np.inf, shape=obs.shape)
        self.observation_space = batch_space(self.single_observation_space, self.num_envs)
        self.tmp = None

    def _use_device_path(self):
        if self.device_resident is None:
            from pdecontrol.surrogates import ops
            dev = _surrogate_device(self.surrogate)
            from pdecontrol.surrogates import hipops
            members = _members(self.surrogate)[0]
            self.device_resident = bool(dev.type == "cuda" and ops.fused_enabled() and self.batched_reward_func is not None
                                        and all(hipops.fused_supported(m) for m in members))
        return self.device_resident

    def setup(self, starting: Dataset):
        self._starting = starting
        self._dev_starting = None
        if self._use_device_path() and hasattr(starting, "datasets"):
            self._dev_starting = _DeviceStartingStates(starting, _surrogate_device(self.surrogate), self.num_envs)
            return
        super().setup(starting)

    def _host_obs(self):
        return self.output.outputs.detach().squeeze(1).cpu().numpy()

    def reset(self, **kwargs):
        self.surrogate.eval()
        with torch.no_grad():
            dev = _surrogate_device(self.surrogate)
            if self._dev_starting is not None:
                sample, last_steps = self._dev_starting.next_batch()
                states, actions = sample.obs, sample.actions
                self.timesteps = np.asarray(last_steps).copy()
            else:
                states, actions, _, _, _, _, steps = next(self.loader)
                self.timesteps = steps[:, -1].numpy()  # env step counter after the warm-up window
            times = self.tstep * torch.arange(actions.size(1))
            targets = self.tstep * actions.size(1)
            states, actions = states.to(dev), actions.to(dev)
            if self._dev is not None and not self._dev.valid():
                self._dev = None     # retrained / re-scaled members (mbrl.py:597-602): the captured graphs are stale
            if self._dev is not None and self._dev.can_replay_reset(states, actions):
                self._dev.replay_reset(states, actions)     # the warm-up rollout of every member: one graph replay
                self.output = None
            else:
                self.output: ModelRollout = self.surrogate.rollout(states=states, actions=actions, hidden=None, times=times,
                                                                    targets=targets)
                if self._use_device_path():
                    hiddens = self.output.hidden if _members(self.surrogate)[1] is not None else [self.output.hidden]
                    if self._dev is None:
                        self._dev = _DeviceWorldState(self, self.output.outputs, hiddens)
                    else:
                        self._dev.load(self.output.outputs, hiddens)
                    if self.capture_reset:
                        self._dev.capture_reset(states, actions)
        self.simulated = 0
        self.tmp = None
        self.surrogate.train()
        obs = self._dev.host_obs() if self._dev is not None else self._host_obs()
        self._host_cache = obs
        if kwargs.get("return_info", False):
            return obs, {"step": self.timesteps.copy()}
        return obs

    def step_async(self, actions: Sequence[Any]) -> None:
        self.surrogate.eval()
        self.simulated += 1
        self.timesteps += 1
        host_actions = np.array(actions, dtype=np.float32)
        if self._dev is not None:
            obs, rewards = self._dev.step(host_actions)
            self._host_cache = obs
            self.tmp = rewards
            self.surrogate.train()
            return
        with torch.no_grad():
            dev = self.output.outputs.device
            # the surrogate expects [B, T, C, A]; the env interface passes [B, C, A]
            act = torch.from_numpy(host_actions).to(dev).unsqueeze(1)
            self.output = self.surrogate.rollout(states=self.output.outputs, actions=act, hidden=self.output.hidden,
                                                 times=0.0, targets=self.tstep)
            obs = self._host_obs()
            self._host_cache = obs
            orescaled = self.stransf.otransf(obs)          # back to the env's observation scale
            arescaled = self.stransf.atransf(host_actions)
            if self.batched_reward_func is not None:
                rewards = np.asarray(self.batched_reward_func(orescaled, arescaled), dtype=np.float32)
            else:
                rewards = np.asarray([self.reward_func(o, a) for o, a in zip(orescaled, arescaled)], dtype=np.float32)
        self.tmp = rewards
        self.surrogate.train()

    def step_wait(self, **kwargs: Any):
        obs = self._host_cache
        rewards = self.tmp
        env_limit = np.broadcast_to(self.timesteps >= self.max_episode_steps, (self.num_envs,))
        rll_limit = np.broadcast_to(self.simulated >= self.horizon, (self.num_envs,))
        # all trajectories are cut together, once every one of them has hit a limit
        truncated = np.broadcast_to(np.all(env_limit | rll_limit), (self.num_envs,))
        terminated = np.zeros(self.num_envs, dtype=np.bool_)
        infos = {"step": self.timesteps.copy()}
        if np.any(truncated):
            infos["_final_observation"] = truncated.copy()
            infos["final_observation"] = obs[truncated]
            obs = self.reset()
        return obs, rewards, terminated, truncated, infos
