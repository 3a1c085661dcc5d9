"""CPU restatement of the reference's ONLY Burgers artefact.  TEST INFRASTRUCTURE ONLY.

The reference ships no Burgers environment (``pdegym/__init__.py:2`` imports a ``pdegym.burgers`` that does not exist)
and no FNO; what it does define is the discretisation, in ``BurgersPhyPDELoss``
(pdecontrol/surrogates/phyloss/phyloss.py:36-86):

    residual(u) = nu * laplace(u) - u * grad(u)                         (:62-81)
      grad    : cross-correlation with [-1/2, 0, 1/2] / dx,             circular   (:39,46-52)
      laplace : cross-correlation with [-1/12, 4/3, -5/2, 4/3, -1/12] / dx^2, circular   (:40,54-60)
    phyevolve(u) = u + dt * residual(u + dt/2 * residual(u))            (:83-86; "improved Euler" = explicit midpoint)

Parity pin: ``residual`` / ``phyevolve`` below are checked against tensors produced by the reference's own class
(oracle/gen_golden.py::burgers_fixtures -> tests/golden/burgers_golden.npz) in tests/test_burgers.py.  Everything
the Burgers ENVIRONMENT adds around this step (forcing, reward, episode logic) has no reference and is
"parity unpinned"; it follows the Kuramoto-Sivashinsky env's conventions.
"""
import numpy as np

GRAD = (-0.5, 0.0, 0.5)                                  # u_{i-1}, u_i, u_{i+1}
LAPLACE = (-1 / 12, 4 / 3, -5 / 2, 4 / 3, -1 / 12)       # u_{i-2} .. u_{i+2}


def grad(u, dx, dtype=np.float32):
    """2nd-order central first derivative of periodic rows [..., N] (taps in ascending index order)."""
    u = np.asarray(u, dtype=dtype)
    return sum(dtype(c) * np.roll(u, -(k - 1), axis=-1) for k, c in enumerate(GRAD) if c != 0.0) / dtype(dx)


def laplace(u, dx, dtype=np.float32):
    """4th-order central second derivative of periodic rows [..., N]."""
    u = np.asarray(u, dtype=dtype)
    return sum(dtype(c) * np.roll(u, -(k - 2), axis=-1) for k, c in enumerate(LAPLACE)) / dtype(dx ** 2)


def residual(u, dx, nu, phi=None, dtype=np.float32):
    """nu * u_xx - u * u_x (+ phi) for rows of a periodic field [..., N]; computed in ``dtype`` like the reference's
    fp32 torch convolutions."""
    u = np.asarray(u, dtype=dtype)
    r = dtype(nu) * laplace(u, dx, dtype) - u * grad(u, dx, dtype)
    return r if phi is None else r + np.asarray(phi, dtype=dtype)


def evolve(u, dx, dt, nu, phi=None, dtype=np.float32):
    """One explicit-midpoint step (the reference's ``phyevolve``)."""
    u = np.asarray(u, dtype=dtype)
    utilde = u + dtype(0.5) * dtype(dt) * residual(u, dx, nu, phi, dtype)
    return u + dtype(dt) * residual(utilde, dx, nu, phi, dtype)


def step(u, phi, dx, dt, nu, n_substeps, dtype=np.float32):
    """n_substeps midpoint steps with a constant forcing field; returns (u_new, sum over sub-steps of sum_i u_i^2 taken
    BEFORE each update, like the KS env's left-Riemann reward, in fp64)."""
    u = np.array(u, dtype=dtype, copy=True)
    ssq = np.zeros(u.shape[:-1], dtype=np.float64)
    for _ in range(int(n_substeps)):
        ssq += np.sum(u.astype(np.float64) ** 2, axis=-1)
        u = evolve(u, dx, dt, nu, phi, dtype)
    return u, ssq
