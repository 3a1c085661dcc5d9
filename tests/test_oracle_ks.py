"""The CPU oracle (oracle/ks_oracle.c) against golden vectors captured from the reference
(oracle/gen_golden.py importing /root/reference/pdegym/kuramoto/kuramoto.py).  No GPU."""
import numpy as np
import pytest

from conftest import KS_CONFIGS
from oracle import ks_oracle as ko


@pytest.mark.parametrize("tag", list(KS_CONFIGS))
def test_rhs_bit_exact(ks_golden, tag):
    L, N = KS_CONFIGS[tag]
    outs = ko.rhs(ks_golden[f"{tag}_rhs_u"], ks_golden[f"{tag}_rhs_phi"], L / N)
    for name, got in zip(("rhs", "ux", "uxx", "uxxxx"), outs):
        np.testing.assert_array_equal(got, ks_golden[f"{tag}_{name}"], err_msg=name)


@pytest.mark.parametrize("tag", list(KS_CONFIGS))
def test_rhs_numpy_restatement(ks_golden, tag):
    # independent np.roll form, different summation order: rounding-level agreement only
    L, N = KS_CONFIGS[tag]
    outs = ko.rhs_numpy(ks_golden[f"{tag}_rhs_u"], ks_golden[f"{tag}_rhs_phi"], L / N)
    for name, got in zip(("rhs", "ux", "uxx", "uxxxx"), outs):
        ref = ks_golden[f"{tag}_{name}"]
        assert np.abs(got - ref).max() <= 1e-14 * max(1.0, np.abs(ref).max()) * 64


@pytest.mark.parametrize("tag", list(KS_CONFIGS))
@pytest.mark.parametrize("n", [1, 2, 10, 250])
def test_step_bit_exact(ks_golden, tag, n):
    L, N = KS_CONFIGS[tag]
    u, rew, ssq, st = ko.step(ks_golden[f"{tag}_traj_u0"], ks_golden[f"{tag}_phi"], L / N, 1e-3, n)
    np.testing.assert_array_equal(u, ks_golden[f"{tag}_traj_u{n}"])
    # reward: torch.norm's reduction order is not restated -> 1e-15 relative
    np.testing.assert_allclose(rew / n, ks_golden[f"{tag}_traj_rew{n}"], rtol=1e-14, atol=0)
    np.testing.assert_allclose(-(ssq / N) / n, ks_golden[f"{tag}_traj_rew{n}"], rtol=1e-14, atol=0)
    assert not st.any()


def test_reward_terms_and_two_steps(ks_golden):
    L, N = KS_CONFIGS["n64"]
    u = ks_golden["n64_traj_u0"][:1]
    phi = ks_golden["n64_phi"][:1]
    terms = []
    for _ in range(10):
        u, rew, _, _ = ko.step(u, phi, L / N, 1e-3, 1)
        terms.append(rew[0])
    np.testing.assert_allclose(terms, ks_golden["n64_rew_terms"], rtol=1e-14)
    u, _, _, _ = ko.step(ks_golden["n64_traj_u0"][:1], ks_golden["n64_phi"][:1], L / N, 1e-3, 250)
    u, rew, _, _ = ko.step(u, ks_golden["n64_phi"][1:2], L / N, 1e-3, 250)
    np.testing.assert_array_equal(u[0], ks_golden["n64_two_steps_u"])
    np.testing.assert_allclose(rew[0] / 250, ks_golden["n64_two_steps_rew"], rtol=1e-14)


def test_survey_known_answer(ks_golden):
    # seed-0 IC, action [0.3,-0.7,1,-1] -> u[:3] after 250 sub-steps
    F = ks_golden["n64_F"]
    phi = ko.phi_from_actions(np.array([[0.3, -0.7, 1.0, -1.0]], dtype=np.float32), F)
    u, rew, _, _ = ko.step(ks_golden["seed0_u0"][None], phi, 22.0 / 64, 1e-3, 250)
    np.testing.assert_array_equal(u[0], ks_golden["seed0_u250"])
    assert u[0, 0] == -0.02812950050187089 and u[0, 1] == 0.01232271446321705
    np.testing.assert_allclose(rew[0] / 250, -0.01171066866857959, rtol=1e-14)


@pytest.mark.parametrize("tag", list(KS_CONFIGS))
def test_forcing_matrix_and_phi(ks_golden, tag):
    L, N = KS_CONFIGS[tag]
    F = ko.forcing_matrix(L, N)
    ref = ks_golden[f"{tag}_F"]
    # libm expf vs torch's vectorised exp: allow 2 ulp (observed: 0)
    assert np.all(np.abs(F - ref) <= 2 * np.spacing(np.abs(ref)))
    phi = ko.phi_from_actions(ks_golden[f"{tag}_actions"], ref)
    np.testing.assert_array_equal(phi, ks_golden[f"{tag}_phi"])
    if tag == "n64":
        assert ref[0, 0] == np.float32(0.6307831406593323) and ref[1, 16] == ref[0, 0]


def test_seeded_reset_burn_in(ks_golden):
    # kuramoto.py:100-116: 800 x 250 sub-steps with phi = 0 from the seeded IC (N=64: ~1 s in C)
    u, _, _, st = ko.step(ks_golden["n64_reset_u0"][None], np.zeros((1, 64), np.float32), 22.0 / 64, 1e-3, 200000)
    np.testing.assert_array_equal(u[0], ks_golden["n64_reset_u"])
    assert ks_golden["n64_reset_step"] == 0 and not st.any()


def test_overflow_is_flagged(ks_golden):
    # L=22, N=256, dt=1e-3 is unstable: the reference raises FloatingPointError
    assert ks_golden["overflow_raises"] == 1
    u0 = np.random.RandomState(0).uniform(-0.4, 0.4, (1, 256))
    _, _, _, st = ko.step(u0, np.zeros((1, 256), np.float32), 22.0 / 256, 1e-3, 250)
    assert st[0] == 1


def test_multithreaded_equals_scalar(ks_golden):
    L, N = KS_CONFIGS["n64"]
    a = ko.step(ks_golden["n64_traj_u0"], ks_golden["n64_phi"], L / N, 1e-3, 50, nthreads=1)
    b = ko.step(ks_golden["n64_traj_u0"], ks_golden["n64_phi"], L / N, 1e-3, 50, nthreads=4)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
