"""Synthetic workloads for the BASELINE.json configs.

Host-side numpy only: this generates *inputs* (q, dq, goals) for tests and bench.py. The small
vectorised FK/Jacobian here exists to place goals near the current pose and to reject
near-singular poses; it is not part of the product path.

Panda numbers: reference examples/15-haptic_control_impedance_type/panda_arm.urdf:17-183.
"""
import numpy as np

DOF = 7

PANDA_XYZ = np.array(
    [[0, 0, 0.333], [0, 0, 0], [0, -0.316, 0], [0.0825, 0, 0], [-0.0825, 0.384, 0], [0, 0, 0], [0.088, 0, 0]],
    dtype=np.float64,
)
PANDA_ROLL = np.array(
    [0, -1.57079632679, 1.57079632679, 1.57079632679, -1.57079632679, 1.57079632679, 1.57079632679]
)
PANDA_LOWER = np.array([-2.8973, -1.7628, -2.8973, -3.0718, -2.8973, -0.0175, -2.8973])
PANDA_UPPER = np.array([2.8973, 1.7628, 2.8973, -0.0698, 2.8973, 3.7525, 2.8973])

# control frame used by every example on the path: link "end-effector" (fixed to link7 at z = 0.15,
# panda_arm.urdf:179-183) + offset (0, 0, 0.07) (examples/05-using_robot_controller.cpp:111-113)
EE_LINK = 6
EE_FRAME_POS = np.array([0.0, 0.0, 0.15 + 0.07])

CONFIG_BATCH = {1: 1, 2: 4096, 3: 65536, 4: 65536, 5: 524288}


def _rx(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], dtype=np.float64)


def fk(q):
    """q [B,7] -> (R [B,7,3,3], p [B,7,3]) link frames in the world."""
    q = np.asarray(q, dtype=np.float64)
    B = q.shape[0]
    R = np.empty((B, DOF, 3, 3))
    p = np.empty((B, DOF, 3))
    Rp = np.broadcast_to(np.eye(3), (B, 3, 3))
    pp = np.zeros((B, 3))
    for i in range(DOF):
        p[:, i] = pp + Rp @ PANDA_XYZ[i]
        c, s = np.cos(q[:, i]), np.sin(q[:, i])
        Rz = np.zeros((B, 3, 3))
        Rz[:, 0, 0] = c
        Rz[:, 0, 1] = -s
        Rz[:, 1, 0] = s
        Rz[:, 1, 1] = c
        Rz[:, 2, 2] = 1
        R[:, i] = Rp @ _rx(PANDA_ROLL[i]) @ Rz
        Rp, pp = R[:, i], p[:, i]
    return R, p


def frame_jacobian(R, p, link=EE_LINK, pos=EE_FRAME_POS):
    """-> (J [B,6,7] linear rows first, x [B,3], Rf [B,3,3])"""
    x = p[:, link] + R[:, link] @ pos
    B = R.shape[0]
    J = np.zeros((B, 6, DOF))
    for i in range(link + 1):
        z = R[:, i, :, 2]
        J[:, 0:3, i] = np.cross(z, x - p[:, i])
        J[:, 3:6, i] = z
    return J, x, R[:, link]


def _expmap(w):
    """rotation matrices exp([w]x) for w [B,3]"""
    th = np.linalg.norm(w, axis=1)
    B = w.shape[0]
    K = np.zeros((B, 3, 3))
    K[:, 0, 1], K[:, 0, 2] = -w[:, 2], w[:, 1]
    K[:, 1, 0], K[:, 1, 2] = w[:, 2], -w[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -w[:, 1], w[:, 0]
    th2 = th * th
    small = th < 1e-8
    a = np.where(small, 1 - th2 / 6, np.sin(th) / np.where(small, 1, th))
    b = np.where(small, 0.5 - th2 / 24, (1 - np.cos(th)) / np.where(small, 1, th2))
    return np.eye(3) + a[:, None, None] * K + b[:, None, None] * (K @ K)


def sample_poses(rng, B, reject_ratio=None, singular_fraction=0.0):
    """q [B,7] uniform in the middle 80 % of each joint range."""
    mid, half = 0.5 * (PANDA_LOWER + PANDA_UPPER), 0.5 * (PANDA_UPPER - PANDA_LOWER)
    out = np.empty((0, DOF))
    while out.shape[0] < B:
        n = max(64, int((B - out.shape[0]) * 1.6))
        q = mid + 0.8 * half * rng.uniform(-1, 1, size=(n, DOF))
        if reject_ratio is not None:
            J, _, _ = frame_jacobian(*fk(q))
            s = np.linalg.svd(J, compute_uv=False)
            q = q[s[:, 5] / s[:, 0] >= reject_ratio]
        out = np.concatenate([out, q])
    q = out[:B].copy()
    if singular_fraction > 0:
        # elbow-extended (q4 near its upper limit) and wrist-aligned (q6 near 0) poses: exercise the
        # blending / singular branches of the SingularityHandler
        n = int(round(B * singular_fraction))
        idx = rng.choice(B, size=n, replace=False)
        half_n = n // 2
        q[idx[:half_n], 3] = rng.uniform(-0.12, -0.07, size=half_n)
        q[idx[half_n:], 5] = rng.uniform(0.0, 0.05, size=n - half_n)
    return q


def make_inputs(config, B=None, seed=None, rank=0):
    """Inputs for BASELINE config 1..5 as a dict of SoA arrays ([C][B], C-contiguous float64).

    keys: q, dq [7,B]; per task t: 'mft{t}' -> dict(pos, rot, v, w, a, alpha) or
    'jt{t}' -> dict(q, dq, ddq). `tasks` lists (kind, params) in hierarchy order.
    """
    if B is None:
        B = CONFIG_BATCH[config] if config != 5 else CONFIG_BATCH[3]
    base = 20241218 + config if seed is None else seed
    rng = np.random.default_rng([base, rank])
    if config == 1:
        tasks = [("jt", {"selection": None})]
    elif config == 2:
        tasks = [("mft", {"partial": None})]
    elif config in (3, 5):
        tasks = [("mft", {"partial": None}), ("jt", {"selection": None})]
    elif config == 4:
        sel = np.zeros((2, DOF))
        sel[0, 0] = 1
        sel[1, 6] = 1
        tasks = [
            ("mft", {"partial": (np.eye(3), np.zeros((0, 3)))}),
            ("jt", {"selection": sel}),
            ("jt", {"selection": None}),
        ]
    else:
        raise ValueError(f"unknown config {config}")
    if config in (2, 3, 5):
        q = sample_poses(rng, B, reject_ratio=0.1)
    elif config == 4:
        q = sample_poses(rng, B, singular_fraction=0.10)
    else:
        q = sample_poses(rng, B)
    dq = rng.normal(0, 0.3, size=(B, DOF))
    out = {"config": config, "B": B, "tasks": tasks, "q": np.ascontiguousarray(q.T), "dq": np.ascontiguousarray(dq.T)}
    R, p = fk(q)
    _, x, Rf = frame_jacobian(R, p)
    for t, (kind, prm) in enumerate(tasks):
        if kind == "mft":
            pos = x + rng.uniform(-0.05, 0.05, size=(B, 3))
            axis = rng.normal(size=(B, 3))
            axis /= np.linalg.norm(axis, axis=1, keepdims=True)
            theta = rng.uniform(0, 0.2, size=(B, 1))
            rot = Rf @ _expmap(axis * theta)
            out[f"mft{t}"] = {
                "pos": np.ascontiguousarray(pos.T),
                "rot": np.ascontiguousarray(rot.reshape(B, 9).T),
                "v": np.ascontiguousarray(rng.normal(0, 0.05, size=(B, 3)).T),
                "w": np.ascontiguousarray(rng.normal(0, 0.05, size=(B, 3)).T),
                "a": np.ascontiguousarray(rng.normal(0, 0.1, size=(B, 3)).T),
                "alpha": np.ascontiguousarray(rng.normal(0, 0.1, size=(B, 3)).T),
            }
        else:
            S = prm["selection"] if prm["selection"] is not None else np.eye(DOF)
            k0 = S.shape[0]
            qg = (S @ q.T).T + rng.normal(0, 0.1, size=(B, k0))
            out[f"jt{t}"] = {
                "q": np.ascontiguousarray(qg.T),
                "dq": np.zeros((k0, B)),
                "ddq": np.zeros((k0, B)),
            }
    return out


def load_inputs(ctrl, inp):
    """feed a make_inputs() dict to a controller object (state, then every task's goals)"""
    ctrl.set_state(inp["q"], inp["dq"])
    for t, (kind, _) in enumerate(inp["tasks"]):
        if kind == "mft":
            g = inp[f"mft{t}"]
            ctrl.set_mft_goals(t, g["pos"], g["rot"], g["v"], g["w"], g["a"], g["alpha"])
        else:
            g = inp[f"jt{t}"]
            ctrl.set_jt_goals(t, g["q"], g["dq"], g["ddq"])
