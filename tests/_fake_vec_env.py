"""Deterministic scripted vector env used to drive BOTH the reference's vector wrappers (in
oracle/gen_golden.py) and this repo's mirror (tests/test_vec_wrappers.py) through identical inputs."""
import numpy as np


def make_fake_vec_env(gym, num_envs=3, width=8, max_steps=3):
    class FakeVecEnv(gym.vector.VectorEnv):
        def __init__(self):
            obs_space = gym.spaces.Box(-np.inf, np.inf, shape=(1, width), dtype=np.float32)
            act_space = gym.spaces.Box(-1.0, 1.0, shape=(1, 4), dtype=np.float32)
            super().__init__(num_envs, obs_space, act_space)
            self.t = np.zeros(num_envs, dtype=np.int64)
            self.total = 0
            self._actions = None

        def _obs(self):
            i = np.arange(1, width + 1, dtype=np.float64)
            rows = [np.sin(0.1 * (self.t[e] + 1 + 0.01 * self.total) * i + e) * (1.0 + 0.1 * e) for e in range(num_envs)]
            return np.stack(rows).astype(np.float32).reshape(num_envs, 1, width)

        def reset_wait(self, seed=None, return_info=False, options=None, **kwargs):
            self.t[:] = 0
            obs = self._obs()
            if return_info:
                return obs, {"step": self.t.copy()}
            return obs

        def reset(self, **kwargs):
            return self.reset_wait(**kwargs)

        def step_async(self, actions):
            self._actions = np.asarray(actions, dtype=np.float32)

        def step_wait(self, **kwargs):
            self.t += 1
            self.total += 1
            obs = self._obs()
            rewards = -np.sum(self._actions.reshape(num_envs, -1).astype(np.float64) ** 2, axis=1)
            truncated = self.t >= max_steps
            infos = {"step": self.t.copy()}
            if truncated.any():
                finals = np.full(num_envs, None, dtype=object)
                for e in np.nonzero(truncated)[0]:
                    finals[e] = obs[e].copy()
                infos["final_observation"] = finals
                infos["_final_observation"] = truncated.copy()
                self.t[truncated] = 0
                obs = np.where(truncated[:, None, None], self._obs(), obs)
            return obs, rewards, np.zeros(num_envs, dtype=bool), truncated, infos

    return FakeVecEnv()


def scripted_actions(num_envs, n_steps):
    rs = np.random.RandomState(123)
    return rs.uniform(-1, 1, size=(n_steps, num_envs, 1, 4)).astype(np.float32)
