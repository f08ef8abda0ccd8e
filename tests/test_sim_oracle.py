"""CPU tests of the simulation harness' dynamics: the oracle's forward dynamics
against the numpy restatement, against the Lagrangian form by finite differences of the mass matrix
(independent of any Newton-Euler recursion), energy conservation, and a closed control loop.
The reference's simulation is the external sai2-simulation (not in its tree): "parity unpinned",
these are textbook rigid-body definitions."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

import make_golden as mg  # noqa: E402
import oracle_lib as ol  # noqa: E402

import sai2_primitives_perso_amd as pkg  # noqa: E402

N = pkg.DOF


def _state(B, seed):
    inp = pkg.workloads.make_inputs(3, B=B, seed=seed)
    rng = np.random.default_rng(seed)
    return inp, inp["q"], rng.normal(0, 0.8, (N, B))


def test_bias_vector_matches_numpy_and_lagrange():
    B = 6
    inp, q, dq = _state(B, 3)
    o = ol.Oracle(ol.panda_model(), ol.task_configs(inp["tasks"]), B)
    o.set_state(q, dq)
    for grav in (False, True):
        b = o.get_bias(grav)
        for k in range(B):
            ref = mg.bias_vector(q[:, k], dq[:, k], grav)
            assert np.abs(b[:, k] - ref).max() < 1e-12 * max(1.0, np.abs(ref).max())
    # Lagrange: b_i = sum_j Mdot_ij dq_j - 1/2 d(dq^T M dq)/dq_i, derivatives by central differences
    h = 1e-6
    b = o.get_bias(False)
    for k in range(2):
        qk, vk = q[:, k], dq[:, k]
        dM = []
        for i in range(N):
            e = np.zeros(N)
            e[i] = h
            dM.append((mg.mass_matrix(qk + e) - mg.mass_matrix(qk - e)) / (2 * h))
        Mdot = sum(dM[i] * vk[i] for i in range(N))
        lag = Mdot @ vk - 0.5 * np.array([vk @ dM[i] @ vk for i in range(N)])
        assert np.abs(b[:, k] - lag).max() < 1e-6 * max(1.0, np.abs(lag).max())
    # at rest the bias with gravity is the gravity vector
    o.set_state(q, np.zeros((N, B)))
    assert np.abs(o.get_bias(True) - o.get_gravity()).max() < 1e-12


def test_sim_step_matches_numpy_and_conserves_energy():
    B = 3
    inp, q, dq = _state(B, 5)
    o = ol.Oracle(ol.panda_model(), ol.task_configs(inp["tasks"]), B)
    o.set_state(q, dq)
    tau = np.random.default_rng(0).normal(0, 5, (N, B))
    qn, vn = q.copy(), dq.copy()
    for _ in range(10):
        o.sim_step(tau, 0.001, substeps=2)
        for k in range(B):
            qn[:, k], vn[:, k] = mg.sim_step(qn[:, k], vn[:, k], tau[:, k], 0.001, 2)
    qo, vo = o.get_state()
    assert np.abs(qo - qn).max() < 1e-12 and np.abs(vo - vn).max() < 1e-11
    # free motion without gravity: kinetic energy is conserved up to the integrator's O(h) error
    o.set_state(q, dq)
    e0 = np.array([0.5 * dq[:, k] @ mg.mass_matrix(q[:, k]) @ dq[:, k] for k in range(B)])
    for _ in range(200):
        o.sim_step(None, 0.001, substeps=10)
    q1, v1 = o.get_state()
    e1 = np.array([0.5 * v1[:, k] @ mg.mass_matrix(q1[:, k]) @ v1[:, k] for k in range(B)])
    assert np.abs(e1 / e0 - 1).max() < 2e-3, e1 / e0


def closed_loop(ctrl, inp, ticks, otg):
    """the examples' loop (examples/05-...cpp:143-196 + sim thread :215-236): state -> models -> torques
    -> simulation, everything staying with the controller object"""
    ctrl.set_state(inp["q"], np.zeros_like(inp["q"]))
    ctrl.reinitialize()
    g = inp["mft0"]
    ctrl.set_mft_goals(0, g["pos"], g["rot"], None, None, None, None)
    ctrl.set_jt_goals(1, inp["jt1"]["q"], None, None)
    for _ in range(ticks):
        ctrl.update_task_models()
        tau = ctrl.compute_control_torques()
        ctrl.sim_step(tau, 0.001, 1)
    return ctrl.get_state()


def test_closed_loop_converges_to_the_goal_pose():
    B = 3
    inp = pkg.workloads.make_inputs(3, B=B, seed=11)
    for otg in (False, True):
        tasks = [ol.motion_force_task("m", internal_otg=otg), ol.joint_task("j", internal_otg=otg)]
        o = ol.Oracle(ol.panda_model(), tasks, B)
        q, dq = closed_loop(o, inp, 2500, otg)
        assert np.abs(dq).max() < 2e-2
        for k in range(B):
            R, p = mg.fk(q[:, k])
            x = p[6] + R[6] @ np.array(pkg.workloads.EE_FRAME_POS)
            assert np.abs(x - inp["mft0"]["pos"][:, k]).max() < 2e-3, (otg, k)
            assert np.abs(R[6] - inp["mft0"]["rot"][:, k].reshape(3, 3)).max() < 5e-3
