"""The whitened forms the SVD-free kernels compute (csrc/sai2b_cert.hpp), restated in numpy and checked against the CPU
oracle's projector-form results — no GPU: this pins the DERIVATION, the GPU tests pin the kernels.

With M = L L^T every nullspace of the cascade is N_prec = L^-T Q L^T with Q an orthogonal projector. A regular level:
Y = Q L^-1 Jr^T = Z R, Lambda = (R^T R)^-1, Q <- Q - Z Z^T. A MotionForceTask inside a singularity-blending region with
one singular direction (SingularityHandler.cpp:76-160, 313-367; round 3: cert::singular_part):
    Jp^T = X, thin SVD Jp = U S V^T, Y' = L^-1 X U:  regular columns Y'_ns = Z_ns R_ns, singular column y_s
    Lambda_s = 1 / (y_s . y_s),   posture  y_p = Q' L^-1 v_s  with  Q' = Q - Z_ns Z_ns^T,   Lambda_joint_s = 1 / (y_p . y_p)
    N N_prec = L^-T (Q' - y_p y_p^T / |y_p|^2) L^T
    tau = tau_ns + alpha clamp(tau_s) + (1 - alpha) tau_joint
"""
import numpy as np
import pytest

import cases
import oracle_lib as ol
import sai2_primitives_perso_amd as pkg

N = 7


def _solve_lower(L, x):
    return np.linalg.solve(L, x)


def _whitened_singular_mft(o, cfg, model, b, q, dq, M, J, Fu, Ff, sv, alpha, c1, c2, decoupling, householder=False):
    """torques and N N_prec of a first-level position task (rows 0..2 of J) with ONE singular direction, as the kernel forms them"""
    L = np.linalg.cholesky(M)
    Jp = J[:3]  # first level: N_prec = I
    X = Jp.T
    U, S, Vt = np.linalg.svd(Jp, full_matrices=False)
    assert np.allclose(S, sv[:3], rtol=1e-9)
    if householder:
        # the kernel's route (cert::singular_streamed): no SVD — the smallest eigenpair of the Gram matrix Jp Jp^T by inverse
        # iteration, sigma_s = |Jp^T u_s|, and a Householder reflector H with H e_3 = +-u_s; the rows of H^T Jp are [an
        # orthonormal mix of the regular directions | +-sigma_s v_s^T]: a regular level does not care which orthonormal
        # basis of the regular subspace it is given
        G = Jp @ Jp.T
        lam = np.linalg.eigvalsh(G)
        us = np.array([1.0, 1.37, 1.74])
        A = G - (lam[0] * (1 - 1e-6) - 1e-14 * lam[-1]) * np.eye(3)
        for _ in range(4):
            us = np.linalg.solve(A, us)
            us /= np.linalg.norm(us)
        sg = -1.0 if us[2] > 0 else 1.0
        w = us * sg - np.array([0, 0, 1.0])
        H = np.eye(3) - 2 * np.outer(w, w) / (w @ w)
        U = H.copy()
        U[:, 2] *= sg  # last column = u_s
        assert abs(np.linalg.norm(Jp.T @ us) - sv[2]) < 1e-11 * sv[0]
        Vt = np.zeros((3, N))
        Vt[2] = Jp.T @ us / np.linalg.norm(Jp.T @ us)
    Xs = X @ U  # columns sigma_j v_j (SVD) / an orthonormal mix of the regular ones and sigma_s v_s (Householder)
    Yp = _solve_lower(L, Xs)
    fu, ff = U.T @ Fu[:3], U.T @ Ff[:3]
    ns, s = [0, 1], 2
    bie = decoupling == pkg.BOUNDED_INERTIA_ESTIMATES
    impedance = decoupling == pkg.IMPEDANCE
    if bie:
        Mb = M.copy()
        for i in range(N):
            Mb[i, i] = max(Mb[i, i], cfg.bie_threshold)
        LB = np.linalg.cholesky(Mb)
        YB = _solve_lower(LB, Xs)
    # Lambda_s_modified U_s^T Fu
    g = (YB[:, s] @ YB[:, s]) if bie else (Yp[:, s] @ Yp[:, s])
    zs = fu[s] / g
    effort = np.array(list(model.effort)[:N])
    tau_s = np.clip(Xs[:, s] * (zs + ff[s]), -effort, effort)
    # regular block
    Z, R = np.linalg.qr(Yp[:, ns])
    tau = Xs[:, ns] @ ff[ns]
    if impedance:
        tau = tau + Xs[:, ns] @ fu[ns]
    elif bie:
        tau = tau + Xs[:, ns] @ np.linalg.solve(YB[:, ns].T @ YB[:, ns], fu[ns])
    else:
        tau = tau + L @ (Z @ np.linalg.solve(R.T, fu[ns]))
    Q1 = np.eye(N) - Z @ Z.T
    # posture
    v = Vt[s].copy()
    u = U[:, s].copy()
    if v[np.argmax(np.abs(v))] < 0:
        v, u = -v, -u
    yp = Q1 @ _solve_lower(L, v)
    if not impedance:
        if c1 > c2 or cfg.enforce_type_1_strategy:
            ut = -cfg.kp_type_1 * (q - q) - cfg.kv_type_1 * dq  # entering conditions: q_prior = q on the first singular tick
            hl, hd = v @ ut, 0.0
        else:
            Fs = Fu + Ff
            nrm = np.linalg.norm(Fs)
            fTd = (Fs[:3] / nrm if nrm > 0 else Fs[:3]) @ u
            qu, ql = np.array(list(model.q_upper)[:N]), np.array(list(model.q_lower)[:N])
            dirs = np.ones(N)
            for i in range(N):
                if v[i] != 0:
                    if abs(q[i] - qu[i]) < cfg.type_2_angle_threshold:
                        dirs[i] = -1
                    elif abs(q[i] - ql[i]) < cfg.type_2_angle_threshold:
                        dirs[i] = 1
            um = dirs * abs(fTd) * cfg.type_2_torque_ratio * effort
            hl, hd = v @ (-cfg.kv_type_2 * dq), v @ um
        if bie:
            yb = _solve_lower(LB, L @ yp)
            lam = 1.0 / (yb @ yb)
        else:
            lam = 1.0 / (yp @ yp)
        tau_j = L @ (yp * (lam * hl + hd))
        tau = tau + alpha * tau_s + (1 - alpha) * tau_j
    Q2 = Q1 - np.outer(yp, yp) / (yp @ yp)
    Linv = np.linalg.inv(L)
    return tau, Linv.T @ Q2 @ L.T, Q2, L


@pytest.mark.parametrize("householder", [False, True])
@pytest.mark.parametrize("decoupling", [0, 1, 2])
def test_singular_branch_in_whitened_coordinates_equals_the_projector_form(decoupling, householder):
    B = 768
    inp = pkg.workloads.make_inputs(4, B=B, seed=4600 + decoupling)
    go = ol.task_configs(inp["tasks"])
    for c in go:
        cases.apply_opts(c, {"decoupling": decoupling})
    model = ol.panda_model()
    o = ol.Oracle(model, go, B, threads=8)
    ol.load_inputs(o, inp)
    o.tick()
    sv, alpha, ro = o.get_mft_singularity(0)
    _, c1, c2 = o.get_mft_sh_state(0)
    Fu, Ff = o.get_mft_task_forces(0)
    Mall, Jall, _, _ = o.get_model(0)
    tau_o = o.get_task_torques(0)
    N0 = o.get_task_nullspace(0)
    N1 = o.get_task_nullspace(1)
    q, dq = inp["q"], inp["dq"]
    sing = np.where(ro == 2)[0]
    assert len(sing) > 20
    if decoupling != pkg.IMPEDANCE:  # (impedance returns before the joint strategy, SingularityHandler.cpp:310-312)
        assert (c1[sing] > 0).any() and (c2[sing] > 0).any()  # both joint strategies occur
    worst_t = worst_n = worst_n1 = 0.0
    for b in sing:
        M = Mall[:, b].reshape(N, N)
        J = Jall[:, b].reshape(6, N)
        tau, Ntot, Q2, L = _whitened_singular_mft(o, go[0], model, b, q[:, b], dq[:, b], M, J, Fu[:, b], Ff[:, b], sv[:, b], alpha[b], c1[b],
                                               c2[b], decoupling, householder)
        worst_t = max(worst_t, np.abs(tau - tau_o[:, b]).max() / max(1.0, np.abs(tau_o[:, b]).max()))
        worst_n = max(worst_n, np.abs(Ntot - N0[:, b].reshape(N, N)).max())
        # the cascade goes on in whitened form: the partial JointTask (joints 0 and 6) behind it, Y = Q2 L^-1 S^T
        S = np.zeros((2, N))
        S[0, 0] = S[1, 6] = 1
        Y = Q2 @ np.linalg.solve(L, S.T)
        Z, _ = np.linalg.qr(Y)
        Linv = np.linalg.inv(L)
        Njt = Linv.T @ (Q2 - Z @ Z.T) @ L.T  # N N_prec of the JointTask (getTaskAndPreviousNullspace)
        worst_n1 = max(worst_n1, np.abs(Njt - N1[:, b].reshape(N, N)).max())
    assert worst_t < 1e-9, worst_t
    assert worst_n < 1e-9 and worst_n1 < 1e-9, (worst_n, worst_n1)


def test_regular_level_in_whitened_coordinates_equals_the_projector_form():
    """a certified level: Lambda = (R^T R)^-1 from the Gram-Schmidt of Y, torques L Z R^-T a, N = L^-T (I - Z Z^T) L^T"""
    B = 64
    inp = pkg.workloads.make_inputs(3, B=B, seed=4700)
    go = ol.task_configs(inp["tasks"])
    for c in go:
        cases.apply_opts(c, {"decoupling": 0})
    o = ol.Oracle(ol.panda_model(), go, B, threads=4)
    ol.load_inputs(o, inp)
    o.tick()
    Mall, Jall, _, _ = o.get_model(0)
    Fu, Ff = o.get_mft_task_forces(0)
    tau_o, N0 = o.get_task_torques(0), o.get_task_nullspace(0)
    Lam, _ = o.get_mft_lambda(0)
    for b in range(B):
        M, J = Mall[:, b].reshape(N, N), Jall[:, b].reshape(6, N)
        L = np.linalg.cholesky(M)
        Y = np.linalg.solve(L, J.T)
        Z, R = np.linalg.qr(Y)
        assert np.abs(np.linalg.inv(R.T @ R) - Lam[:, b].reshape(6, 6)).max() < 1e-8 * np.abs(Lam[:, b]).max()
        tau = L @ (Z @ np.linalg.solve(R.T, Fu[:, b])) + J.T @ Ff[:, b]
        assert np.abs(tau - tau_o[:, b]).max() < 1e-9 * max(1.0, np.abs(tau_o[:, b]).max())
        Linv = np.linalg.inv(L)
        assert np.abs(Linv.T @ (np.eye(N) - Z @ Z.T) @ L.T - N0[:, b].reshape(N, N)).max() < 1e-9
