"""kspde -- thin ctypes binding of libkspde.so (C ABI: include/kspde.h).

This is the only way Python reaches the HIP stepper.  There is no fallback: if the shared
library has not been built (``python __graft_entry__.py`` / ``make -C csrc``) or no HIP device is
visible, construction fails loudly.  ``device=-1`` (or ``"cpu"``) is an explicit request for the library's CPU twin
(csrc/ks_cpu.cpp, same entry points): BASELINE configs[0] on a GPU-less host.

``KSPDE_LIB`` overrides the library path (the sanitizer build: ``make -C csrc asan``).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "..", "lib", "libkspde.so")

MODE = {"fast": 0, "exact": 1}
VARIANT = {"auto": 0, "row16_dpp": 1, "row16_bperm": 2, "wave64_dpp": 3, "wave64_bperm": 4,
           "half32_bperm": 5, "lds": 6, "wave64_hybrid": 7, "wave64_hybrid1": 8}
VARIANT_NAME = {v: k for k, v in VARIANT.items()}

# every symbol include/kspde.h declares: (name, restype, argtypes)
_c = ctypes
_H = _c.c_void_p
_dp, _fp, _ip = _c.POINTER(_c.c_double), _c.POINTER(_c.c_float), _c.POINTER(_c.c_int)
SYMBOLS = (
    ("ks_create", _c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.c_double, _c.c_double, _c.POINTER(_H)]),
    ("ks_destroy", _c.c_int, [_H]),
    ("ks_set_stream", _c.c_int, [_H, _c.c_void_p]),
    ("ks_set_mode", _c.c_int, [_H, _c.c_int]),
    ("ks_set_variant", _c.c_int, [_H, _c.c_int]),
    ("ks_set_block_size", _c.c_int, [_H, _c.c_int]),
    ("ks_get_layout", _c.c_int, [_H, _ip, _ip, _ip, _ip, _ip]),
    ("ks_set_forcing", _c.c_int, [_H, _c.c_void_p, _c.c_int]),
    ("ks_set_state", _c.c_int, [_H, _c.c_void_p]),
    ("ks_get_state", _c.c_int, [_H, _c.c_void_p]),
    ("ks_set_state_rows", _c.c_int, [_H, _c.c_void_p, _c.c_int, _c.c_void_p]),
    ("ks_state_device_ptr", _c.c_int, [_H, _c.POINTER(_c.c_void_p)]),
    ("ks_step", _c.c_int, [_H, _c.c_void_p, _c.c_long, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    ("ks_step_actions", _c.c_int, [_H, _c.c_void_p, _c.c_long, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    ("ks_step_rows", _c.c_int, [_H, _c.c_void_p, _c.c_int, _c.c_long, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    ("ks_step_begin", _c.c_int, [_H, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_long, _c.c_int]),
    ("ks_step_end", _c.c_int, [_H, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    ("ks_step_device", _c.c_int, [_H, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_long, _c.c_void_p,
                                  _c.c_void_p, _c.c_void_p]),
    ("ks_sync", _c.c_int, [_H]),
    ("ks_rhs", _c.c_int, [_H, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                          _c.c_void_p]),
    ("ks_selftest", _c.c_int, [_H, _c.POINTER(_c.c_uint)]),
    ("ks_last_error", _c.c_char_p, []),
    ("ks_version", _c.c_char_p, []),
)

_lib = None


class KSError(RuntimeError):
    pass


def load():
    """dlopen libkspde.so and type every exported symbol.  Raises if it is missing."""
    global _lib
    if _lib is None:
        # torch ships its own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7).  Import it
        # FIRST so that libkspde's NEEDED libamdhip64.so.7 binds to that already-loaded copy: two HIP
        # runtimes in one process cannot both own the GPU ("No HIP GPUs are available").
        import torch  # noqa: F401
        path = os.path.abspath(os.environ.get("KSPDE_LIB") or LIB_PATH)
        if not os.path.exists(path):
            raise KSError(
                f"{path} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; "
                f"g.build()' or make -C model-based-pde-control_amd/csrc). There is no CPU fallback.")
        lib = ctypes.CDLL(path)
        for name, res, args in SYMBOLS:
            fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _check(rc):
    if rc != 0:
        msg = load().ks_last_error().decode(errors="replace")
        raise KSError(f"libkspde error {rc}: {msg}")


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class KSStepper:
    """A batch of ``num_envs`` independent KS states resident in HBM on one GPU."""

    def __init__(self, num_envs, N=64, L=22.0, dt=1e-3, device=0, mode="fast", variant="auto"):
        self._lib = load()
        self._h = _H()
        if isinstance(device, str):
            device = -1 if device == "cpu" else int(device)
        self.num_envs, self.N, self.L, self.dt, self.device = int(num_envs), int(N), float(L), float(dt), int(device)
        self.dx = self.L / self.N
        _check(self._lib.ks_create(self.device, self.num_envs, self.N, self.L, self.dt, ctypes.byref(self._h)))
        self.set_mode(mode)
        self.set_variant(variant)
        self.n_act = 0

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.ks_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- configuration ----------------------------------------------------------------------
    def set_mode(self, mode):
        _check(self._lib.ks_set_mode(self._h, MODE[mode] if isinstance(mode, str) else int(mode)))
        self.mode = mode

    def set_variant(self, variant):
        _check(self._lib.ks_set_variant(self._h, VARIANT[variant] if isinstance(variant, str) else int(variant)))

    def set_block_size(self, threads):
        _check(self._lib.ks_set_block_size(self._h, int(threads)))

    def set_stream(self, stream_handle):
        _check(self._lib.ks_set_stream(self._h, ctypes.c_void_p(int(stream_handle))))

    def layout(self):
        v = [ctypes.c_int() for _ in range(5)]
        _check(self._lib.ks_get_layout(self._h, *[ctypes.byref(x) for x in v]))
        return {"variant": VARIANT_NAME[v[0].value], "lanes_per_env": v[1].value, "points_per_lane": v[2].value,
                "block": v[3].value, "grid": v[4].value}

    def set_forcing(self, F):
        F = np.ascontiguousarray(F, dtype=np.float32)
        assert F.ndim == 2 and F.shape[1] == self.N, F.shape
        _check(self._lib.ks_set_forcing(self._h, _ptr(F), F.shape[0]))
        self.n_act = F.shape[0]

    # -- state ------------------------------------------------------------------------------
    def set_state(self, u):
        u = np.ascontiguousarray(u, dtype=np.float64)
        assert u.shape == (self.num_envs, self.N), u.shape
        _check(self._lib.ks_set_state(self._h, _ptr(u)))

    def get_state(self):
        u = np.empty((self.num_envs, self.N), dtype=np.float64)
        _check(self._lib.ks_get_state(self._h, _ptr(u)))
        return u

    def set_state_rows(self, env_ids, u):
        ids = np.ascontiguousarray(env_ids, dtype=np.int32)
        u = np.ascontiguousarray(u, dtype=np.float64)
        assert u.shape == (len(ids), self.N)
        _check(self._lib.ks_set_state_rows(self._h, _ptr(ids), len(ids), _ptr(u)))

    def state_device_ptr(self):
        p = ctypes.c_void_p()
        _check(self._lib.ks_state_device_ptr(self._h, ctypes.byref(p)))
        return p.value

    # -- hot path ---------------------------------------------------------------------------
    def _outs(self, n, want_obs):
        obs = np.empty((n, self.N), dtype=np.float32) if want_obs else None
        return obs, np.empty(n, dtype=np.float64), np.empty(n, dtype=np.int32)

    def step(self, phi=None, n_substeps=250, want_obs=True):
        """phi: fp32 [num_envs, N] or None (= 0).  Returns (obs_f32 | None, ssq_sum, status)."""
        if phi is not None:
            phi = np.ascontiguousarray(phi, dtype=np.float32)
            assert phi.shape == (self.num_envs, self.N), phi.shape
        obs, ssq, st = self._outs(self.num_envs, want_obs)
        _check(self._lib.ks_step(self._h, _ptr(phi), int(n_substeps), _ptr(obs), _ptr(ssq), _ptr(st)))
        return obs, ssq, st

    def step_actions(self, actions, n_substeps=250, want_obs=True):
        a = np.ascontiguousarray(np.asarray(actions, dtype=np.float32).reshape(self.num_envs, -1))
        assert a.shape[1] == self.n_act, (a.shape, self.n_act)
        obs, ssq, st = self._outs(self.num_envs, want_obs)
        _check(self._lib.ks_step_actions(self._h, _ptr(a), int(n_substeps), _ptr(obs), _ptr(ssq), _ptr(st)))
        return obs, ssq, st

    def step_rows(self, env_ids, n_substeps, want_obs=True):
        ids = np.ascontiguousarray(env_ids, dtype=np.int32)
        obs, ssq, st = self._outs(len(ids), want_obs)
        _check(self._lib.ks_step_rows(self._h, _ptr(ids), len(ids), int(n_substeps), _ptr(obs), _ptr(ssq), _ptr(st)))
        return obs, ssq, st

    def step_begin(self, actions=None, env_ids=None, n_substeps=250, want_obs=True):
        """Enqueue a host-boundary step and return without waiting (``step_end`` collects it): begin on every handle of
        a multi-device env first, then end on each."""
        a = None
        if actions is not None:
            a = np.ascontiguousarray(np.asarray(actions, dtype=np.float32).reshape(self.num_envs, -1))
            assert a.shape[1] == self.n_act, (a.shape, self.n_act)
        ids = None if env_ids is None else np.ascontiguousarray(env_ids, dtype=np.int32)
        self._pending_rows = self.num_envs if ids is None else len(ids)
        self._pending_obs = bool(want_obs)
        _check(self._lib.ks_step_begin(self._h, _ptr(a), _ptr(ids), 0 if ids is None else len(ids), int(n_substeps),
                                       int(bool(want_obs))))

    def step_end(self, out=None):
        """Wait for the step ``step_begin`` enqueued.  ``out`` = (obs, ssq, status) arrays (or slices of larger
        C-contiguous ones) to fill; allocated when None."""
        n = self._pending_rows
        obs, ssq, st = out if out is not None else self._outs(n, self._pending_obs)
        for a_, shape in ((obs, (n, self.N)), (ssq, (n,)), (st, (n,))):
            assert a_ is None or (a_.shape == shape and a_.flags["C_CONTIGUOUS"]), (None if a_ is None else a_.shape, shape)
        _check(self._lib.ks_step_end(self._h, _ptr(obs), _ptr(ssq), _ptr(st)))
        return obs, ssq, st

    def step_device(self, d_phi=0, d_actions=0, d_env_ids=0, n_rows=0, n_substeps=250, d_obs=0, d_ssq=0,
                    d_status=0):
        """Asynchronous launch on raw device pointers (ints, 0 = NULL); see ks_step_device."""
        vp = lambda x: ctypes.c_void_p(int(x)) if x else None
        _check(self._lib.ks_step_device(self._h, vp(d_phi), vp(d_actions), vp(d_env_ids), int(n_rows),
                                        int(n_substeps), vp(d_obs), vp(d_ssq), vp(d_status)))

    def sync(self):
        _check(self._lib.ks_sync(self._h))

    # -- test hooks -------------------------------------------------------------------------
    def rhs(self, u, phi):
        u = np.ascontiguousarray(np.atleast_2d(u), dtype=np.float64)
        phi = np.ascontiguousarray(np.atleast_2d(phi), dtype=np.float32)
        assert u.shape == phi.shape and u.shape[1] == self.N
        outs = [np.empty_like(u) for _ in range(4)]
        _check(self._lib.ks_rhs(self._h, _ptr(u), _ptr(phi), u.shape[0], *[_ptr(o) for o in outs]))
        return tuple(outs)

    def selftest(self):
        m = ctypes.c_uint(0)
        rc = self._lib.ks_selftest(self._h, ctypes.byref(m))
        return rc, m.value
