"""Scripted scenarios for the OTG wrappers, shared by the fixture generator (numpy wrappers on the
reference's ruckig core, make_otg_golden.py) and the tests (oracle / GPU).

A scenario is (kind, dim, dt, x0, limits, n_ticks, events); events maps a tick to a list of
("goal", ...), ("reinit", ...), ("limits", ...) actions applied before that tick's update. Every
tick does what the tasks do: setGoal(current goal), update(), read the next state
(JointTask.cpp:313-320, MotionForceTask.cpp:394-407).
"""
import numpy as np
from scipy.spatial.transform import Rotation

DT = 0.001
RECORD_STRIDE = 3


def _rot(rng, max_angle):
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    return Rotation.from_rotvec(axis * rng.uniform(0.2, max_angle)).as_matrix()


def scenarios():
    rng = np.random.default_rng(20241218)
    out = {}
    # ---- joint OTG, 7 dof: re-goal mid-trajectory, goal = current, reinit, goal with velocity
    # (finishes with velocity -> re-goal with zero velocity), limit change while moving
    x0 = rng.uniform(-1, 1, 7)
    g1 = x0 + rng.uniform(-0.4, 0.4, 7)
    g2 = x0 + rng.uniform(-0.3, 0.3, 7)
    g3 = g2 + np.array([0.05, 0, 0, -0.04, 0, 0, 0.02])
    ev = {
        0: [("goal", g1, np.zeros(7))],
        120: [("goal", g2, np.zeros(7))],
        520: [("reinit", g2 + 0.01)],
        560: [("goal", g3, np.array([0.1, 0, 0, -0.05, 0, 0, 0.0]))],
        900: [("goal", g1, np.zeros(7))],
        960: [("limits", np.full(7, 0.5), np.full(7, 3.0))],
    }
    out["joints7"] = ("joints", 7, DT, x0, (np.full(7, np.pi / 3), np.full(7, 2 * np.pi)), 1500, ev)
    # ---- joint OTG, 7 dof, phase-synchronisable (collinear) motion and per-joint limits
    x0 = rng.uniform(-1, 1, 7)
    d = rng.uniform(-1, 1, 7)
    ev = {0: [("goal", x0 + 0.5 * d, np.zeros(7))], 250: [("goal", x0 - 0.2 * d, np.zeros(7))]}
    out["joints7_collinear"] = ("joints", 7, DT, x0, (rng.uniform(0.5, 1.5, 7), rng.uniform(2, 8, 7)), 1600, ev)
    # ---- joint OTG, 2 dof (partial task): includes an invalid target velocity (> max) for a while
    x0 = rng.uniform(-1, 1, 2)
    ev = {
        0: [("goal", x0 + np.array([0.3, -0.2]), np.zeros(2))],
        200: [("goal", x0 + np.array([0.1, 0.2]), np.array([2.0, 0.0]))],  # |vf| > vmax: ErrorInvalidInput
        260: [("goal", x0 + np.array([0.1, 0.2]), np.zeros(2))],
    }
    out["joints2_invalid"] = ("joints", 2, DT, x0, (np.full(2, np.pi / 3), np.full(2, 2 * np.pi)), 700, ev)
    # ---- joint OTG, 1 dof
    ev = {0: [("goal", np.array([0.7]), np.zeros(1))], 300: [("goal", np.array([-0.2]), np.zeros(1))]}
    out["joints1"] = ("joints", 1, DT, np.array([0.1]), (np.array([1.0]), np.array([4.0])), 1800, ev)
    # ---- Cartesian OTG: pose goal, re-goal mid-way, sub-threshold change (ignored), goal with
    # linear and angular velocity, reinit
    p0 = np.array([0.4, 0.1, 0.5])
    R0 = _rot(rng, 1.0)
    pa, Ra = p0 + np.array([0.1, -0.05, 0.08]), R0 @ _rot(rng, 0.6)
    pb, Rb = p0 + np.array([-0.05, 0.1, 0.02]), R0 @ _rot(rng, 0.9)
    z3 = np.zeros(3)
    ev = {
        0: [("goal", pa, Ra, z3, z3)],
        150: [("goal", pb, Rb, z3, z3)],
        800: [("goal", pb * (1 + 2e-4), Rb, z3, z3)],  # inside isApprox(1e-3): ignored
        820: [("goal", pa, Ra, np.array([0.05, 0, 0]), np.array([0, 0.1, 0]))],
        1500: [("reinit", p0, R0)],
        1530: [("goal", pb, Ra, z3, z3)],
    }
    out["cartesian"] = ("cartesian", 6, DT, (p0, R0), (0.3, 2.0, np.pi / 3, 2 * np.pi), 2200, ev)
    return out


def run(scn, make_joints, make_cartesian):
    """make_joints(x0, dt) / make_cartesian(pos, rot, dt) -> object with set_limits, set_goal*,
    reinitialize, update, next (the numpy wrappers, or adapters over the oracle)."""
    kind, dim, dt, x0, limits, n_ticks, events = scn
    rec = []
    if kind == "joints":
        o = make_joints(x0, dt)
        o.set_limits(*limits)
        o.disable_jerk_limits()
        goal = (np.array(x0, float), np.zeros(dim))
        for k in range(n_ticks):
            for e in events.get(k, []):
                if e[0] == "goal":
                    goal = (e[1], e[2])
                elif e[0] == "reinit":
                    o.reinitialize(e[1])
                    goal = (np.array(e[1], float), np.zeros(dim))
                elif e[0] == "limits":
                    o.set_limits(e[1], e[2])
                    o.disable_jerk_limits()
            o.set_goal(*goal)
            o.update()
            if k % RECORD_STRIDE == 0 or k in events:
                p, v, a = o.next()
                rec.append(np.concatenate([[k, float(o.goal_reached), float(o.result)], p, v, a]))
    else:
        o = make_cartesian(x0[0], x0[1], dt)
        o.set_limits(*limits)
        goal = (x0[0], x0[1], np.zeros(3), np.zeros(3))
        for k in range(n_ticks):
            for e in events.get(k, []):
                if e[0] == "goal":
                    goal = e[1:]
                elif e[0] == "reinit":
                    o.reinitialize(e[1], e[2])
                    goal = (e[1], e[2], np.zeros(3), np.zeros(3))
            o.set_goal_position(goal[0], goal[2])
            o.set_goal_orientation(goal[1], goal[3])
            o.update()
            if k % RECORD_STRIDE == 0 or k in events:
                p, R, v, w, a, al = o.next()
                rec.append(np.concatenate([[k, float(o.goal_reached), float(o.result)], p, R.ravel(), v, w, a, al]))
    return np.array(rec)
