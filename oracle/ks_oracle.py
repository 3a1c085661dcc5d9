"""ctypes loader for the C oracle + a NumPy restatement.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg may import this
module; the product (model-based-pde-control_amd/) never does.

Parity pin: checked against tests/golden/ks_golden.npz (generated from the reference by
oracle/gen_golden.py) in tests/test_oracle_ks.py.

Reference: pdegym/kuramoto/kuramoto.py:78-98 (step), :118-129 (rhs);
pdegym/common/transforms.py:250-265 (GaussianForcing).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libks_oracle.so")
_lib = None

_dp = ctypes.POINTER(ctypes.c_double)
_fp = ctypes.POINTER(ctypes.c_float)
_ip = ctypes.POINTER(ctypes.c_int)

XI = (0.0, 0.25, 0.5, 0.75)  # kuramoto.py:18


def build(force=False):
    """Compile oracle/ks_oracle.c with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "ks_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        L.ks_oracle_rhs.argtypes = [_dp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_double, _dp, _dp, _dp, _dp]
        L.ks_oracle_step.argtypes = [_dp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                     ctypes.c_long, _dp, _dp, _ip, ctypes.c_int]
        L.ks_oracle_forcing.argtypes = [ctypes.c_double, ctypes.c_int, ctypes.c_double, _dp, ctypes.c_int, _fp]
        L.ks_oracle_phi.argtypes = [_fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp]
        for f in (L.ks_oracle_rhs, L.ks_oracle_step, L.ks_oracle_forcing, L.ks_oracle_phi):
            f.restype = ctypes.c_int
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp)


def _f(a):
    return a.ctypes.data_as(_fp)


def rhs(u, phi, dx):
    """u [E,N] f64, phi [E,N] f32 -> (rhs, ux, uxx, uxxxx) each [E,N] f64."""
    u = np.ascontiguousarray(np.atleast_2d(u), dtype=np.float64)
    phi = np.ascontiguousarray(np.atleast_2d(phi), dtype=np.float32)
    E, N = u.shape
    outs = [np.empty_like(u) for _ in range(4)]
    rc = lib().ks_oracle_rhs(_d(u), _f(phi), E, N, float(dx), *map(_d, outs))
    assert rc == 0
    return tuple(outs)


def step(u, phi, dx, dt, n_substeps, nthreads=1):
    """Advance u [E,N] by n_substeps RK4 sub-steps (copy).  Returns
    (u_new, reward_sum[E], ssq_sum[E], status[E]); reward_sum is the reference's
    running ``reward`` before the final ``/ cfg_steps``."""
    u = np.array(np.atleast_2d(u), dtype=np.float64, order="C", copy=True)
    phi = np.ascontiguousarray(np.atleast_2d(phi), dtype=np.float32)
    E, N = u.shape
    assert phi.shape == (E, N)
    rew = np.zeros(E)
    ssq = np.zeros(E)
    st = np.zeros(E, dtype=np.int32)
    rc = lib().ks_oracle_step(_d(u), _f(phi), E, N, float(dx), float(dt), int(n_substeps), _d(rew), _d(ssq),
                              st.ctypes.data_as(_ip), int(nthreads))
    assert rc == 0
    return u, rew, ssq, st


def forcing_matrix(L, N, sigma=0.4, Xi=XI):
    xi = np.asarray(Xi, dtype=np.float64)
    F = np.empty((len(xi), N), dtype=np.float32)
    lib().ks_oracle_forcing(float(L), int(N), float(sigma), _d(xi), len(xi), _f(F))
    return F


def phi_from_actions(actions, F):
    a = np.ascontiguousarray(np.asarray(actions, dtype=np.float32).reshape(-1, F.shape[0]))
    F = np.ascontiguousarray(F, dtype=np.float32)
    out = np.empty((a.shape[0], F.shape[1]), dtype=np.float32)
    lib().ks_oracle_phi(_f(a), _f(F), a.shape[0], F.shape[0], F.shape[1], _f(out))
    return out


# --------------------------------------------------------------------------- #
# NumPy restatement (np.roll form) -- a second, independent statement of rhs/RK4 used to
# cross-check the C oracle, and the per-env "reference-structured" CPU baseline.
# --------------------------------------------------------------------------- #
_FWD = (-25 / 12, 4.0, -3.0, 4 / 3, -1 / 4)      # q_i, q_{i+1} .. q_{i+4}
_D2 = (-49 / 18, 3 / 2, -3 / 20, 1 / 90)         # u_i, u_{i±1}, u_{i±2}, u_{i±3}
_D4 = (91 / 8, -122 / 15, 169 / 60, -2 / 5, 7 / 240)


def rhs_numpy(u, phi, dx):
    u = np.asarray(u, dtype=np.float64)
    q = u * u
    sh = lambda a, k: np.roll(a, -k, axis=-1)  # a_{i+k}
    fwd = sum(c * sh(q, k) for k, c in enumerate(_FWD)) / dx
    bwd = sum(-c * sh(q, -k) for k, c in enumerate(_FWD)) / dx
    ux = np.where(u < 0, fwd, bwd)
    uxx = (_D2[0] * u + sum(_D2[k] * (sh(u, k) + sh(u, -k)) for k in (1, 2, 3))) / dx ** 2
    uxxxx = (_D4[0] * u + sum(_D4[k] * (sh(u, k) + sh(u, -k)) for k in (1, 2, 3, 4))) / dx ** 4
    return -uxxxx - uxx - 0.5 * ux + np.asarray(phi, dtype=np.float64), ux, uxx, uxxxx


def step_numpy(u, phi, dx, dt, n_substeps):
    u = np.array(u, dtype=np.float64, copy=True)
    N = u.shape[-1]
    reward = np.zeros(u.shape[:-1])
    for _ in range(int(n_substeps)):
        reward += -(1.0 / N) * np.sum(u * u, axis=-1)
        k1 = rhs_numpy(u, phi, dx)[0]
        k2 = rhs_numpy(u + dt * k1 / 2.0, phi, dx)[0]
        k3 = rhs_numpy(u + dt * k2 / 2.0, phi, dx)[0]
        k4 = rhs_numpy(u + dt * k3, phi, dx)[0]
        u = u + dt * (k1 + 2.0 * k2 + 2.0 * k3 + k4) / 6.0
    return u, reward


# --------------------------------------------------------------------------- #
# "Reference-structured" per-env Python baseline (baseline A): one env
# object per process, a Python loop over sub-steps, 16 scipy.ndimage.convolve1d(mode="wrap")
# calls and one numpy->torch->numpy reward round trip per sub-step -- the cost structure of
# pdegym/kuramoto/kuramoto.py:78-129, restated (not copied) for timing only.
# --------------------------------------------------------------------------- #
def step_scipy_structured(u, phi, dx, dt, n_substeps):
    import torch
    from scipy.ndimage import convolve1d
    tab_fwd = [-1 / 4, 4 / 3, -3, 4, -25 / 12, 0, 0, 0, 0]
    tab_bwd = [0, 0, 0, 0, 25 / 12, -4, 3, -4 / 3, 1 / 4]
    tab_d2 = [1 / 90, -3 / 20, 3 / 2, -49 / 18, 3 / 2, -3 / 20, 1 / 90]
    tab_d4 = [7 / 240, -2 / 5, 169 / 60, -122 / 15, 91 / 8, -122 / 15, 169 / 60, -2 / 5, 7 / 240]
    N = u.shape[-1]

    def f(v):
        q = v ** 2
        fw = convolve1d(q, weights=tab_fwd, mode="wrap") / dx
        bw = convolve1d(q, weights=tab_bwd, mode="wrap") / dx
        ux = (v < 0) * fw + (v >= 0) * bw
        uxx = convolve1d(v, weights=tab_d2, mode="wrap") / dx ** 2
        uxxxx = convolve1d(v, weights=tab_d4, mode="wrap") / dx ** 4
        return -uxxxx - uxx - 0.5 * ux + phi

    u = np.array(u, dtype=np.float64, copy=True)
    reward = 0.0
    for _ in range(int(n_substeps)):
        reward += ((-1.0) * (1 / N) * torch.norm(torch.from_numpy(u)) ** 2).numpy()
        k1 = f(u)
        k2 = f(u + dt * k1 / 2.0)
        k3 = f(u + dt * k2 / 2.0)
        k4 = f(u + dt * k3)
        u = u + dt * (k1 + 2.0 * k2 + 2.0 * k3 + k4) / 6.0
    return u, float(reward)


def _structured_worker(argv):
    """``python oracle/ks_oracle.py --structured-worker N L seconds seed``: one process of bench.py's baseline (A) --
    one env, the reference's call structure, stepped for ~``seconds``.  Prints "ready" once imports and first calls
    are done, starts when a line arrives on stdin (the parent releases all workers together)."""
    import json
    import sys
    import time
    N, L, seconds, seed = int(argv[0]), float(argv[1]), float(argv[2]), int(argv[3])
    rs = np.random.RandomState(seed)
    u = rs.uniform(-0.4, 0.4, N)
    phi = (0.1 * rs.uniform(-1, 1, N)).astype(np.float32).astype(np.float64)
    u, _ = step_scipy_structured(u, phi, L / N, 1e-3, 20)      # imports + first-call costs out of the timed loop
    print("ready", flush=True)
    sys.stdin.readline()
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        u, _ = step_scipy_structured(u, phi, L / N, 1e-3, 250)
        done += 250
    dt = time.perf_counter() - t0
    assert np.isfinite(u).all()
    print(json.dumps({"substeps": done, "seconds": dt}))


if __name__ == "__main__":
    import sys
    if len(sys.argv) >= 6 and sys.argv[1] == "--structured-worker":
        _structured_worker(sys.argv[2:])
