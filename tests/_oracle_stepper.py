"""Test double with the kspde.KSStepper interface, backed by the CPU oracle.

Lives under tests/ on purpose: it lets the `-m "not gpu"` suite exercise the HOST logic of the
gym env / vector env (timestep bookkeeping, autoreset, seeding, reward scaling, dtype contracts)
without a GPU.  The product never imports it -- the shipped envs default to the HIP stepper and
raise without it.
"""
import numpy as np

from oracle import ks_oracle as ko


class OracleStepper:
    def __init__(self, num_envs, N=64, L=22.0, dt=1e-3, device=0, mode="fast", variant="auto"):
        self.num_envs, self.N, self.L, self.dt = num_envs, N, L, dt
        self.dx = L / N
        self.mode = mode
        self.u = np.zeros((num_envs, N))
        self.F = None
        self.n_act = 0
        self.closed = False

    def close(self):
        self.closed = True

    def set_mode(self, mode):
        self.mode = mode

    def set_variant(self, variant):
        pass

    def set_forcing(self, F):
        self.F = np.ascontiguousarray(F, dtype=np.float32)
        self.n_act = self.F.shape[0]

    def set_state(self, u):
        self.u = np.array(u, dtype=np.float64).reshape(self.num_envs, self.N)

    def get_state(self):
        return self.u.copy()

    def set_state_rows(self, ids, u):
        self.u[np.asarray(ids)] = u

    def _run(self, ids, phi, n, want_obs):
        if phi is None:
            phi = np.zeros((len(ids), self.N), dtype=np.float32)
        with np.errstate(all="ignore"):
            u, _, ssq, st = ko.step(self.u[ids], phi, self.dx, self.dt, n)
        self.u[ids] = u
        return (u.astype(np.float32) if want_obs else None), ssq, st

    def step(self, phi=None, n_substeps=250, want_obs=True):
        return self._run(np.arange(self.num_envs), phi, n_substeps, want_obs)

    def step_actions(self, actions, n_substeps=250, want_obs=True):
        phi = ko.phi_from_actions(np.asarray(actions, np.float32).reshape(self.num_envs, -1), self.F)
        return self._run(np.arange(self.num_envs), phi, n_substeps, want_obs)

    def step_rows(self, ids, n_substeps, want_obs=True):
        return self._run(np.asarray(ids), None, n_substeps, want_obs)

    def rhs(self, u, phi):
        return ko.rhs(u, phi, self.dx)

    def layout(self):
        return {"variant": "oracle"}
