"""HIP stepper (through the C ABI) against the CPU oracle and the golden vectors.

Tolerances (BASELINE.json north_star): state L_inf < 1e-9 per sub-step in fp64.
  * KS_MODE_EXACT keeps the reference's operation order -> asserted BIT-EXACT.
  * KS_MODE_FAST (merged stencil, FMA) -> asserted <= 1e-12 after 1 sub-step (contract: 1e-9),
    and <= 1e-9 after 250 sub-steps.
"""
import numpy as np
import pytest

from conftest import KS_CONFIGS

pytestmark = pytest.mark.gpu

FUSED = ["row16_dpp", "row16_bperm", "wave64_dpp", "wave64_bperm", "half32_bperm", "lds", "wave64_hybrid", "wave64_hybrid1"]


def _supported(variant, N):
    if variant.startswith("wave64_hybrid"):
        return N == 64
    P = {"row16": 16, "wave64": 64, "half32": 32}.get(variant.split("_")[0])
    if P is None:
        return 9 <= N <= 2048
    return N % P == 0 and (N // P) in (1, 2, 3, 4, 6, 8, 12, 16)


@pytest.fixture(scope="module")
def kspde():
    import kspde
    kspde.load()
    return kspde


def test_selftest_cross_lane_primitives(kspde):
    s = kspde.KSStepper(4, 64)
    rc, mask = s.selftest()
    assert rc == 0 and mask == 0, f"failing variant mask 0x{mask:x}"


@pytest.mark.parametrize("tag", list(KS_CONFIGS))
def test_rhs_hook_bit_exact(kspde, ks_golden, tag):
    L, N = KS_CONFIGS[tag]
    s = kspde.KSStepper(1, N, L)
    outs = s.rhs(ks_golden[f"{tag}_rhs_u"], ks_golden[f"{tag}_rhs_phi"])
    for name, got in zip(("rhs", "ux", "uxx", "uxxxx"), outs):
        np.testing.assert_array_equal(got, ks_golden[f"{tag}_{name}"], err_msg=name)


@pytest.mark.parametrize("variant", FUSED)
@pytest.mark.parametrize("tag", list(KS_CONFIGS))
def test_exact_mode_bit_exact_vs_golden(kspde, ks_golden, tag, variant):
    L, N = KS_CONFIGS[tag]
    if not _supported(variant, N):
        pytest.skip("layout not instantiated for this N")
    s = kspde.KSStepper(8, N, L, mode="exact", variant=variant)
    assert s.layout()["variant"] == variant
    for n in (1, 2, 10, 250):
        s.set_state(ks_golden[f"{tag}_traj_u0"])
        obs, ssq, st = s.step(ks_golden[f"{tag}_phi"], n)
        u = s.get_state()
        np.testing.assert_array_equal(u, ks_golden[f"{tag}_traj_u{n}"], err_msg=f"n={n}")
        np.testing.assert_array_equal(obs, u.astype(np.float32))
        np.testing.assert_allclose(-(ssq / N) / n, ks_golden[f"{tag}_traj_rew{n}"], rtol=1e-13)
        assert not st.any()


@pytest.mark.parametrize("variant", FUSED)
@pytest.mark.parametrize("tag", list(KS_CONFIGS))
def test_fast_mode_within_tolerance(kspde, ks_golden, tag, variant):
    L, N = KS_CONFIGS[tag]
    if not _supported(variant, N):
        pytest.skip("layout not instantiated for this N")
    s = kspde.KSStepper(8, N, L, mode="fast", variant=variant)
    for n, tol in ((1, 1e-12), (2, 1e-12), (10, 1e-11), (250, 1e-9)):
        s.set_state(ks_golden[f"{tag}_traj_u0"])
        obs, ssq, st = s.step(ks_golden[f"{tag}_phi"], n)
        u = s.get_state()
        err = np.abs(u - ks_golden[f"{tag}_traj_u{n}"]).max()
        assert err <= tol, (n, err)
        np.testing.assert_allclose(-(ssq / N) / n, ks_golden[f"{tag}_traj_rew{n}"], rtol=1e-10)


@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_actions_path_matches_phi_path(kspde, ks_golden, mode):
    # in-kernel phi = actions @ F (fp32 FMA chain) must equal the golden fp32 phi bit for bit
    for tag in ("n64", "n256"):
        L, N = KS_CONFIGS[tag]
        s = kspde.KSStepper(8, N, L, mode=mode)
        s.set_forcing(ks_golden[f"{tag}_F"])
        s.set_state(ks_golden[f"{tag}_traj_u0"])
        s.step_actions(ks_golden[f"{tag}_actions"], 10)
        ua = s.get_state()
        s.set_state(ks_golden[f"{tag}_traj_u0"])
        s.step(ks_golden[f"{tag}_phi"], 10)
        np.testing.assert_array_equal(ua, s.get_state())


def test_ragged_batch_and_subset_rows(kspde, ks_golden):
    # 37 envs (not a multiple of the envs-per-wave of any layout) against the oracle
    from oracle import ks_oracle as ko
    L, N = KS_CONFIGS["n64"]
    rs = np.random.RandomState(3)
    u0 = rs.uniform(-0.4, 0.4, (37, N))
    phi = rs.uniform(-0.5, 0.5, (37, N)).astype(np.float32)
    ref, _, ssq_ref, _ = ko.step(u0, phi, L / N, 1e-3, 25)
    for variant in FUSED:
        s = kspde.KSStepper(37, N, L, mode="exact", variant=variant)
        s.set_state(u0)
        _, ssq, st = s.step(phi, 25)
        np.testing.assert_array_equal(s.get_state(), ref, err_msg=variant)
        np.testing.assert_allclose(ssq, ssq_ref, rtol=1e-13)
    # subset stepping (masked burn-in): only the listed envs move, phi = 0
    s = kspde.KSStepper(37, N, L, mode="exact")
    s.set_state(u0)
    ids = np.array([5, 0, 36, 17], dtype=np.int32)
    obs, ssq, st = s.step_rows(ids, 40)
    ref_rows, _, ssq_rows, _ = ko.step(u0[ids], np.zeros((4, N), np.float32), L / N, 1e-3, 40)
    u = s.get_state()
    np.testing.assert_array_equal(u[ids], ref_rows)
    mask = np.ones(37, bool)
    mask[ids] = False
    np.testing.assert_array_equal(u[mask], u0[mask])
    np.testing.assert_array_equal(obs, ref_rows.astype(np.float32))
    np.testing.assert_allclose(ssq, ssq_rows, rtol=1e-13)
    # scatter rows
    s.set_state_rows(ids, u0[ids] * 2.0)
    np.testing.assert_array_equal(s.get_state()[ids], u0[ids] * 2.0)


def test_generic_lds_kernel_odd_sizes(kspde):
    from oracle import ks_oracle as ko
    for N in (9, 50, 100, 333):
        L = 0.34375 * N
        rs = np.random.RandomState(N)
        u0 = rs.uniform(-0.4, 0.4, (5, N))
        phi = rs.uniform(-0.5, 0.5, (5, N)).astype(np.float32)
        ref, _, ssq_ref, _ = ko.step(u0, phi, L / N, 1e-3, 20)
        s = kspde.KSStepper(5, N, L, mode="exact")
        assert s.layout()["variant"] == "lds"
        s.set_state(u0)
        _, ssq, _ = s.step(phi, 20)
        np.testing.assert_array_equal(s.get_state(), ref)
        np.testing.assert_allclose(ssq, ssq_ref, rtol=1e-13)
        s.set_mode("fast")
        s.set_state(u0)
        s.step(phi, 20)
        assert np.abs(s.get_state() - ref).max() < 1e-11


def test_seeded_reset_burn_in_on_gpu(kspde, ks_golden):
    # 200 000 sub-steps in one launch, exact mode: bit-identical to the reference's reset()
    for tag in ("n64", "n256"):
        L, N = KS_CONFIGS[tag]
        s = kspde.KSStepper(1, N, L, mode="exact")
        s.set_state(ks_golden[f"{tag}_reset_u0"][None])
        s.step(None, 200000)
        np.testing.assert_array_equal(s.get_state()[0], ks_golden[f"{tag}_reset_u"])
        # fast mode: its ~1e-16 rounding differences are amplified by the chaotic dynamics over
        # T = 200 (observed 3e-3 at N=64), so the end state is only statistically comparable:
        # same attractor (energy within a factor 2), finite.  Bit-level reset parity is what
        # exact mode is for; the env runs its burn-in in exact mode by default.
        s.set_mode("fast")
        s.set_state(ks_golden[f"{tag}_reset_u0"][None])
        _, _, st = s.step(None, 200000)
        uf = s.get_state()[0]
        ref = ks_golden[f"{tag}_reset_u"]
        print(f"fast-mode burn-in deviation {tag}: {np.abs(uf - ref).max():.3e}")
        assert not st.any() and np.isfinite(uf).all()
        assert 0.5 < np.mean(uf ** 2) / np.mean(ref ** 2) < 2.0


def test_overflow_status_flag(kspde):
    u0 = np.random.RandomState(0).uniform(-0.4, 0.4, (3, 256))
    for mode in ("exact", "fast"):
        s = kspde.KSStepper(3, 256, 22.0, mode=mode)
        s.set_state(u0)
        _, _, st = s.step(None, 250)
        assert st.all()


def test_full_size_properties_c2_c3(kspde):
    """BASELINE configs at full size: size-independent properties instead of an oracle run.
    (a) env independence / permutation equivariance, (b) translation equivariance of the periodic
    stencils (roll the IC and phi -> rolled result), (c) spot rows against the oracle."""
    from oracle import ks_oracle as ko
    for (E, N, L) in ((1024, 64, 22.0), (4096, 256, 88.0)):
        rs = np.random.RandomState(11)
        u0 = rs.uniform(-0.4, 0.4, (E, N))
        phi = rs.uniform(-0.3, 0.3, (E, N)).astype(np.float32)
        s = kspde.KSStepper(E, N, L, mode="exact")
        s.set_state(u0)
        s.step(phi, 50)
        u = s.get_state()
        perm = rs.permutation(E)
        s.set_state(u0[perm])
        s.step(phi[perm], 50)
        np.testing.assert_array_equal(s.get_state(), u[perm])
        shift = 7
        s.set_state(np.roll(u0, shift, axis=1))
        s.step(np.roll(phi, shift, axis=1), 50)
        np.testing.assert_array_equal(s.get_state(), np.roll(u, shift, axis=1))
        rows = [0, 1, E // 2, E - 1]
        ref, _, _, _ = ko.step(u0[rows], phi[rows], L / N, 1e-3, 50)
        np.testing.assert_array_equal(u[rows], ref)
        s.set_mode("fast")
        s.set_state(u0)
        s.step(phi, 50)
        assert np.abs(s.get_state() - u).max() < 1e-10
