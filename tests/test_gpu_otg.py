"""GPU parity tests (-m gpu) of the tasks' internal OTG (csrc/sai2b_otg.hip, sai2b_otg_core.hpp),
through the C ABI, against the CPU oracle (oracle/otg_oracle.c, itself pinned bit-for-bit against the
reference's own ruckig core) and against the fixture the numpy restatement made on that core.

Tolerances: the generator is compiled without FMA contraction and follows the reference's operation
order, so its states agree with the oracle to rounding of sin/cos/atan2 (1e-12 absolute on states of
order 1); torques keep the path's 1e-10 relative."""
import numpy as np
import pytest

import oracle_lib as ol
import sai2_primitives_perso_amd as pkg
from test_otg_oracle import GOLDEN, drive_otg_fixture

pytestmark = pytest.mark.gpu
N = pkg.DOF
TOL = 1e-10


def _err(tau, ref):
    scale = np.maximum(np.abs(ref).max(axis=0), 1.0)
    return np.abs(tau - ref).max(axis=0) / scale


def _c3_pair(B, otg=(True, True)):
    to = [ol.motion_force_task("motion_force_task_0", internal_otg=otg[0]), ol.joint_task("joint_task_1", internal_otg=otg[1])]
    tg = [pkg.motion_force_task_config("motion_force_task_0", internal_otg=otg[0]),
          pkg.joint_task_config("joint_task_1", internal_otg=otg[1])]
    return ol.Oracle(ol.panda_model(), to, B, threads=8), pkg.Controller(pkg.panda_model(), tg, B)


def test_gpu_otg_controller_follows_fixture_and_oracle():
    """[MFT, JT], both OTGs on (the reference's default), 420 ticks, goals changed at ticks 140 and
    300 (the last with goal velocities), through the SVD-free kernels"""
    import os

    z = np.load(os.path.join(GOLDEN, "c3_otg_ticks.npz"))
    B = z["tau"].shape[2]
    inp = pkg.workloads.make_inputs(3, B=B)
    o, g = _c3_pair(B)
    ro, rg = drive_otg_fixture(o, inp), drive_otg_fixture(g, inp)
    for k in rg:
        scale = max(1.0, np.abs(z[k]).max())
        assert np.abs(rg[k] - ro[k]).max() / scale < (TOL if k == "tau" else 1e-12), k
        assert np.abs(rg[k] - z[k]).max() / scale < TOL, k


def _random_goal_run(o, g, inp, ticks, rng, jt_task, mft_task, check_every=1):
    """both controllers get the same per-robot goal changes at random ticks"""
    B = inp["B"]
    for c in (o, g):
        c.set_state(inp["q"], inp["dq"])
        c.reinitialize()
    k0 = o.tasks[jt_task].task_dof if jt_task is not None else 0
    if jt_task is not None:
        gq = o.get_jt_desired(jt_task)[0].copy()
        gdq = np.zeros((k0, B))
    if mft_task is not None:
        d = o.get_mft_desired(mft_task)
        gpos, grot, gv, gw = d[0].copy(), d[1].copy(), np.zeros((3, B)), np.zeros((3, B))
    worst = {"tau": 0.0, "state": 0.0}
    from scipy.spatial.transform import Rotation

    for tick in range(ticks):
        change = rng.random(B) < (1.0 if tick == 0 else 0.01)  # each robot re-goals about every 100 ticks
        idx = np.nonzero(change)[0]
        if len(idx):
            if jt_task is not None:
                gq[:, idx] += rng.normal(0, 0.2, (k0, len(idx)))
                gdq[:, idx] = np.where(rng.random((1, len(idx))) < 0.2, rng.normal(0, 0.05, (k0, len(idx))), 0.0)
            if mft_task is not None:
                gpos[:, idx] += rng.uniform(-0.06, 0.06, (3, len(idx)))
                for b in idx:
                    R = grot[:, b].reshape(3, 3) @ Rotation.from_rotvec(rng.normal(0, 0.2, 3)).as_matrix()
                    grot[:, b] = R.ravel()
                gv[:, idx] = np.where(rng.random((1, len(idx))) < 0.2, rng.normal(0, 0.02, (3, len(idx))), 0.0)
            for c in (o, g):
                if jt_task is not None:
                    c.set_jt_goals(jt_task, gq, gdq, None)
                if mft_task is not None:
                    c.set_mft_goals(mft_task, gpos, grot, gv, gw, None, None)
        to, tg = o.tick(), g.tick()
        if tick % check_every == 0:
            worst["tau"] = max(worst["tau"], _err(tg, to).max())
            if jt_task is not None:
                for a, b_ in zip(o.get_jt_desired(jt_task), g.get_jt_desired(jt_task)):
                    worst["state"] = max(worst["state"], np.abs(a - b_).max())
            if mft_task is not None:
                for a, b_ in zip(o.get_mft_desired(mft_task), g.get_mft_desired(mft_task)):
                    worst["state"] = max(worst["state"], np.abs(a - b_).max())
    for t in (jt_task, mft_task):
        if t is not None:
            for a, b_ in zip(o.get_otg_status(t), g.get_otg_status(t)):
                assert np.array_equal(a, b_), "goal-reached / result flags differ"
    return worst


def test_gpu_otg_random_regoals_c3():
    """256 robots re-goal independently (divergent lanes: some re-plan, some sample, some idle)"""
    B = 256
    inp = pkg.workloads.make_inputs(3, B=B, seed=5)
    o, g = _c3_pair(B)
    w = _random_goal_run(o, g, inp, 260, np.random.default_rng(1), jt_task=1, mft_task=0, check_every=4)
    assert w["state"] < 1e-12 and w["tau"] < TOL, w
    reached = g.get_otg_status(1)[0]
    assert 0 < reached.sum() < B or reached.sum() in (0, B)


def test_gpu_otg_generic_hierarchy_partial_joint_task():
    """[MFT(translation), JT(2 joints), JT(7)] with every OTG on: generic kernel, 2- and 7-DoF
    joint generators, ragged batch"""
    B = 100
    inp = pkg.workloads.make_inputs(4, B=B, seed=8)
    # keep the robots away from the singular poses of config 4 (not what this test is about)
    sel = inp["tasks"][1][1]["selection"]
    part = inp["tasks"][0][1]["partial"]
    to = [ol.motion_force_task("m", partial=part, internal_otg=True), ol.joint_task("j2", sel, internal_otg=True),
          ol.joint_task("j7", internal_otg=True)]
    tg = [pkg.motion_force_task_config("m", partial=part, internal_otg=True), pkg.joint_task_config("j2", sel, internal_otg=True),
          pkg.joint_task_config("j7", internal_otg=True)]
    o = ol.Oracle(ol.panda_model(), to, B, threads=8)
    g = pkg.Controller(pkg.panda_model(), tg, B)
    q = pkg.workloads.make_inputs(3, B=B, seed=8)
    inp2 = dict(inp)
    inp2["q"], inp2["dq"] = q["q"], q["dq"]
    rng = np.random.default_rng(3)
    for c in (o, g):
        c.set_state(inp2["q"], inp2["dq"])
        c.reinitialize()
    g2 = o.get_jt_desired(1)[0] + rng.normal(0, 0.2, (1, B))
    g7 = o.get_jt_desired(2)[0] + rng.normal(0, 0.2, (7, B))
    d = o.get_mft_desired(0)
    gp = d[0] + rng.uniform(-0.05, 0.05, (3, B))
    worst = 0.0
    for tick in range(150):
        if tick in (0, 60):
            g2 = g2 + rng.normal(0, 0.1, (2, B))
            for c in (o, g):
                c.set_jt_goals(1, g2, None, None)
                c.set_jt_goals(2, g7, None, None)
                c.set_mft_goals(0, gp, None, None, None, None, None)
        to_, tg_ = o.tick(), g.tick()
        worst = max(worst, _err(tg_, to_).max())
        for t in (1, 2):
            for a, b_ in zip(o.get_jt_desired(t), g.get_jt_desired(t)):
                assert np.abs(a - b_).max() < 1e-12, (tick, t)
    assert worst < 10 * TOL, worst  # generic hierarchy: conditioning as in test_gpu_parity's certified hierarchies


def test_gpu_otg_enable_disable_and_limit_change():
    """disableInternalOtg / enableInternalOtgAccelerationLimited through sai2b_update_task_config while
    running: off -> goal passthrough; on again -> re-initialised at the current state; new limits ->
    moving robots re-plan"""
    B = 64
    inp = pkg.workloads.make_inputs(3, B=B, seed=21)
    o, g = _c3_pair(B)
    rng = np.random.default_rng(9)
    for c in (o, g):
        c.set_state(inp["q"], inp["dq"])
        c.reinitialize()
        c.set_jt_goals(1, inp["jt1"]["q"], None, None)
        c.set_mft_goals(0, inp["mft0"]["pos"], inp["mft0"]["rot"], None, None, None, None)

    def step(n):
        w = 0.0
        for _ in range(n):
            w = max(w, _err(g.tick(), o.tick()).max())
            for a, b_ in zip(o.get_jt_desired(1) + o.get_mft_desired(0), g.get_jt_desired(1) + g.get_mft_desired(0)):
                assert np.abs(a - b_).max() < 1e-12
        return w

    assert step(40) < TOL
    moving = g.get_jt_desired(1)[1]
    assert np.abs(moving).max() > 1e-3
    # off: the desired state is the goal again
    for c, mk in ((o, ol), (g, pkg)):
        cfg = c.tasks[1]
        cfg.use_internal_otg = 0
        c.update_task_config(1, cfg)
    assert step(3) < TOL
    assert np.array_equal(g.get_jt_desired(1)[0], inp["jt1"]["q"])
    # on with tighter limits: generator restarts from the current joint positions
    for c in (o, g):
        cfg = c.tasks[1]
        cfg.use_internal_otg = 1
        for i in range(N):
            cfg.otg_max_velocity[i], cfg.otg_max_acceleration[i] = 0.4, 2.0
        c.update_task_config(1, cfg)
    assert step(30) < TOL
    assert np.abs(g.get_jt_desired(1)[1]).max() <= 0.4 + 1e-12
    # limits of the Cartesian generator change mid-motion
    for c in (o, g):
        cfg = c.tasks[0]
        cfg.otg_max_linear_velocity, cfg.otg_max_angular_velocity = 0.1, 0.5
        c.update_task_config(0, cfg)
    assert step(60) < TOL
    # jerk-limited generator: its limits are checked like OTG_joints::setMaxJerk does (tests/test_gpu_otg3.py runs it)
    cfg = g.tasks[1]
    cfg.internal_otg_jerk_limited = 1
    cfg.otg_max_jerk[2] = -1.0
    with pytest.raises(ValueError, match="max jerk"):
        g.update_task_config(1, cfg)
    cfg.internal_otg_jerk_limited, cfg.otg_max_jerk[2] = 0, 10 * np.pi


def test_gpu_otg_reaches_goal_and_idles():
    """after the trajectories end every generator reports isGoalReached and the desired state is the
    goal with zero velocity; an unchanged goal then costs no state traffic (same outputs)"""
    B = 128
    inp = pkg.workloads.make_inputs(3, B=B, seed=2)
    _, g = _c3_pair(B)
    g.set_state(inp["q"], inp["dq"])
    g.reinitialize()
    q0 = g.get_jt_desired(1)[0]
    goal = q0 + 0.02
    g.set_jt_goals(1, goal, None, None)
    for _ in range(400):
        g.tick()
    reached, result = g.get_otg_status(1)
    assert reached.all() and (result == 1).all()
    q, dq, ddq = g.get_jt_desired(1)
    assert np.abs(q - goal).max() < 1e-12 and np.abs(dq).max() < 1e-12
    a = g.tick()
    b = g.tick()
    assert np.array_equal(a, b)
    assert g.get_otg_status(0)[0].all()  # the Cartesian generator never had to move


def test_gpu_facade_otg_on_by_default():
    B = 64
    inp = pkg.workloads.make_inputs(3, B=B, seed=4)
    robot = pkg.BatchedRobotModel(B)
    robot.setQ(inp["q"])
    robot.setDq(inp["dq"])
    robot.updateModel()
    mft, jt = pkg.MotionForceTask(robot, task_name="ee"), pkg.JointTask(robot)
    assert mft.getInternalOtgEnabled() and jt.getInternalOtgEnabled()
    ctl = pkg.RobotController(robot, [mft, jt])
    jt.setGoalPosition(inp["jt1"]["q"])
    jt.enableInternalOtgAccelerationLimited(0.5, 3.0)
    for _ in range(20):
        ctl.tick()
    v = jt.getDesiredVelocity()
    assert 0 < np.abs(v).max() <= 0.5 + 1e-12
    assert np.abs(jt.getDesiredPosition() - inp["jt1"]["q"]).max() > 1e-3  # still on its way
    with pytest.raises(ValueError, match="max jerk"):
        jt.enableInternalOtgJerkLimited(1.0, 1.0, 0.0)
    with pytest.raises(ValueError, match="max velocity"):
        mft.enableInternalOtgAccelerationLimited(0.0, 1.0, 1.0, 1.0)
    jt.disableInternalOtg()
    ctl.tick()
    assert np.array_equal(jt.getDesiredPosition(), inp["jt1"]["q"])


def test_gpu_otg_idle_shortcut_still_sees_new_goals():
    """an idle generator is skipped from its flag alone while the host knows the goals are untouched;
    a setter call (host array or device tensor) must end that, and so must handing out the goals'
    device pointer (the caller may then write goals behind the library's back)"""
    import torch

    B = 128
    inp = pkg.workloads.make_inputs(3, B=B, seed=6)
    o, g = _c3_pair(B)
    for c in (o, g):
        c.set_state(inp["q"], inp["dq"])
        c.reinitialize()
    for _ in range(5):  # goals = current pose: everything idles after the first tick
        to, tg = o.tick(), g.tick()
    assert g.get_otg_status(1)[0].all() and _err(tg, to).max() < TOL
    goal = inp["q"] + 0.05
    for c in (o, g):
        c.set_jt_goals(1, goal, None, None)
    for _ in range(30):
        to, tg = o.tick(), g.tick()
    assert not g.get_otg_status(1)[0].any() and _err(tg, to).max() < TOL
    assert np.abs(o.get_jt_desired(1)[0] - g.get_jt_desired(1)[0]).max() < 1e-12
    for _ in range(600):  # finish and idle again
        to, tg = o.tick(), g.tick()
    assert g.get_otg_status(1)[0].all()
    # goal from a device tensor
    goal2 = goal - 0.03
    g.set_jt_goals(1, torch.as_tensor(goal2, device="cuda"), None, None)
    o.set_jt_goals(1, goal2, None, None)
    for _ in range(700):
        to, tg = o.tick(), g.tick()
    assert g.get_otg_status(1)[0].all() and _err(tg, to).max() < TOL
    # the device pointer of the goals is handed out: the caller overwrites the goals without any setter
    ptr = g.device_buffer(pkg._abi.BUF_GOALS, 1)
    goal3 = goal2 + 0.02
    class _Raw:  # zero-copy torch view of the library's goals buffer (q, dq, ddq goals: [21][B])
        __cuda_array_interface__ = {"data": (int(ptr), False), "shape": (21, B), "typestr": "<f8", "version": 2}

    view = torch.as_tensor(_Raw(), device="cuda")
    view[:7].copy_(torch.as_tensor(goal3, device="cuda"))
    torch.cuda.synchronize()
    o.set_jt_goals(1, goal3, None, None)
    for _ in range(25):
        to, tg = o.tick(), g.tick()
    assert not g.get_otg_status(1)[0].any() and _err(tg, to).max() < TOL
    assert np.abs(o.get_jt_desired(1)[0] - g.get_jt_desired(1)[0]).max() < 1e-12


def test_generator_kernels_are_not_launched_while_every_generator_is_idle(monkeypatch):
    """round 3: otg_kernel leaves the count of robots with a generator that is not idle; read back (every 8th tick,
    asynchronously) as zero for a tick launched with the goals as they still are, it lets the host drop the two generator
    launches from the tick — an idle generator's update neither reads nor writes anything — until a goal or a generator's
    configuration is touched. Same torques as with SAI2B_NO_OTG_IDLE_SKIP=1, launch for launch fewer."""
    B = 256
    inp = pkg.workloads.make_inputs(3, B=B, seed=8)
    o, g = _c3_pair(B)
    monkeypatch.setenv("SAI2B_NO_OTG_IDLE_SKIP", "1")
    _, h = _c3_pair(B)
    monkeypatch.delenv("SAI2B_NO_OTG_IDLE_SKIP")
    for c in (o, g, h):
        c.set_state(inp["q"], inp["dq"])
        c.reinitialize()

    def per_tick(c, n=4):
        l0 = c.counters()[0]
        for _ in range(n):
            c.tick()
        return (c.counters()[0] - l0) / n

    for _ in range(24):  # goals = current pose: idle after the first tick; the host has seen it by now
        to, tg, th = o.tick(), g.tick(), h.tick()
        g.synchronize()
        assert np.array_equal(tg, th) and _err(tg, to).max() < TOL
    assert per_tick(h) == 3 and per_tick(g) == 1  # counted launches: [the generators' two] + the tick's kernels as one
    for _ in range(4):
        o.tick()  # (the launch-count ticks above, for the oracle)
    goal = inp["q"] + 0.04
    for c in (o, g, h):
        c.set_jt_goals(1, goal, None, None)
    for _ in range(40):
        to, tg, th = o.tick(), g.tick(), h.tick()
        assert np.array_equal(tg, th) and _err(tg, to).max() < TOL
    assert not g.get_otg_status(1)[0].any()
    assert per_tick(g) == 3  # on their way: the generator kernels run


def test_gpu_otg_reinitialize_mid_motion_ragged_batch():
    """reinitializeTasks() while the generators are moving (OTG objects re-initialised at the current
    pose, goals reset), a batch that fills neither a wavefront nor the planner's 8-lane groups, and two
    controllers alive on the device at once"""
    B = 70
    inp = pkg.workloads.make_inputs(3, B=B, seed=13)
    o, g = _c3_pair(B)
    o2, g2 = _c3_pair(B)  # a second, independent pair: contexts must not share any state
    rng = np.random.default_rng(2)
    for c in (o, g, o2, g2):
        c.set_state(inp["q"], inp["dq"])
        c.reinitialize()
    goal = inp["q"] + rng.normal(0, 0.15, (N, B))
    for c in (o, g):
        c.set_jt_goals(1, goal, None, None)
        c.set_mft_goals(0, inp["mft0"]["pos"], inp["mft0"]["rot"], None, None, None, None)
    for c in (o2, g2):
        c.set_jt_goals(1, goal[::-1].copy(), None, None)
    for tick in range(120):
        if tick == 60:
            q2 = inp["q"] + 0.01
            for c in (o, g):
                c.set_state(q2, inp["dq"])
                c.reinitialize()
                c.set_jt_goals(1, goal, None, None)
        to, tg = o.tick(), g.tick()
        t2o, t2g = o2.tick(), g2.tick()
        assert _err(tg, to).max() < TOL and _err(t2g, t2o).max() < TOL, tick
        for a, b_ in zip(o.get_jt_desired(1) + o.get_mft_desired(0), g.get_jt_desired(1) + g.get_mft_desired(0)):
            assert np.abs(a - b_).max() < 1e-12, tick
    for a, b_ in zip(o.get_otg_status(1) + o2.get_otg_status(1), g.get_otg_status(1) + g2.get_otg_status(1)):
        assert np.array_equal(a, b_)


@pytest.mark.parametrize("introspection", [True, False])
def test_gpu_otg_of_task_with_empty_range_waits_like_the_reference(introspection):
    """(introspection off: the range pass ahead of the generators is the SVD-free cascade for the robots it can
    certify — sai2b_cert.hpp: range_tick — and the generic kernel's for the others, here the singular half.)
    [MFT(6), JT(1 joint), JT(7)]: on a regular robot the last task has no degree of freedom left, and the
    reference then returns before it touches the task's generator (JointTask.cpp:302-306), so the
    generator waits — time, state and pending goal changes — until the task has a range again (here: when
    the robot is moved into the 6-DOF task's singular region, whose handling strategy is off so that the
    lost direction is passed down, and back). The GPU decides that per robot and tick with a model-only
    pass ahead of the generator kernel (DevTask::otg_gated)."""
    B = 96
    inp = pkg.workloads.make_inputs(3, B=B, seed=12)
    sel = np.zeros((1, N))
    sel[0, 2] = 1
    to = [ol.motion_force_task("m", internal_otg=True), ol.joint_task("j1", sel, internal_otg=True),
          ol.joint_task("j7", internal_otg=True)]
    tg = [pkg.motion_force_task_config("m", internal_otg=True), pkg.joint_task_config("j1", sel, internal_otg=True),
          pkg.joint_task_config("j7", internal_otg=True)]
    to[0].enforce_handling_strategy = tg[0].enforce_handling_strategy = 0
    o = ol.Oracle(ol.panda_model(), to, B, threads=8)
    g = pkg.Controller(pkg.panda_model(), tg, B, introspection=introspection)
    q_reg = inp["q"].copy()
    q_sing = q_reg.copy()
    q_sing[3, : B // 2] = -0.0715  # elbow nearly extended: the 6-DOF task loses a direction for half the batch
    rng = np.random.default_rng(5)
    for c in (o, g):
        c.set_state(q_reg, 0.1 * inp["dq"])
        c.reinitialize()
    g7 = o.get_jt_desired(2)[0] + rng.normal(0, 0.3, (N, B))
    g2 = o.get_jt_desired(1)[0] + rng.normal(0, 0.2, (1, B))
    waiting = moving = 0
    ever_active = np.zeros(B, dtype=bool)
    for tick in range(90):
        if tick == 0:
            for c in (o, g):
                c.set_jt_goals(1, g2, None, None)
                c.set_jt_goals(2, g7, None, None)
        if tick == 10:  # goal change while the task is waiting on every robot
            g7 = g7 + rng.normal(0, 0.2, (N, B))
            for c in (o, g):
                c.set_jt_goals(2, g7, None, None)
        if tick in (20, 50):  # half the batch into the singular pose / back out of it
            for c in (o, g):
                c.set_state(q_sing if tick == 20 else q_reg, 0.1 * inp["dq"])
        if tick == 35:  # and a goal change while it is moving for half of the robots
            g7 = g7 + rng.normal(0, 0.2, (N, B))
            for c in (o, g):
                c.set_jt_goals(2, g7, None, None)
        tau_o, tau_g = o.tick(), g.tick()
        _, _, ro = o.get_mft_singularity(0)
        e = _err(tau_g, tau_o)
        assert e[ro == 6].max() < 10 * TOL, (tick, e[ro == 6].max())
        assert e.max() < 1e-6, (tick, e.max())
        for t in (1, 2):
            for a, b_ in zip(o.get_jt_desired(t), g.get_jt_desired(t)):
                assert np.abs(a - b_).max() < 1e-12, (tick, t)
            for a, b_ in zip(o.get_otg_status(t), g.get_otg_status(t)):
                assert np.array_equal(a, b_), (tick, t)
        active = np.abs(o.get_task_torques(2)).max(axis=0) > 0
        ever_active |= active
        if 20 <= tick < 50:
            moving += int(active[: B // 2].sum())
        else:
            waiting += int((~active).sum())
    # the scenario did exercise both phases (a robot or two have the last task active in the regular pose too:
    # there the one-joint task is the one with the empty range)
    assert waiting > 0.95 * 60 * B and moving > 0.95 * 30 * (B // 2), (waiting, moving)
    # the generator of a robot that waited all along never advanced: still at the state of reinitialize()
    des_q = g.get_jt_desired(2)[0]
    assert (~ever_active).sum() > B // 3 and np.abs(des_q[:, ~ever_active] - q_reg[:, ~ever_active]).max() < 1e-12


def test_gpu_certified_range_pass_agrees_with_the_generic_one(monkeypatch):
    """BASELINE config 4's hierarchy [MFT(3), JT(2), JT(7)] with every generator on, 10 % of the poses near-singular:
    the range decision ahead of the generators from the SVD-free cascade (+ the generic kernel for the robots it
    declines) against the generic kernel's for the whole batch (SAI2B_NO_CERT_PATH=1): the generators' states must
    be bit-equal tick after tick — they only see "active or not" —, the torques agree as two kernels do, and both
    follow the oracle."""
    B = 2048 + 19
    inp = pkg.workloads.make_inputs(4, B=B, seed=77)
    mk = lambda make: [c for c in make(inp["tasks"])]
    co, cg, cr = ol.task_configs(inp["tasks"]), pkg.task_configs(inp["tasks"]), pkg.task_configs(inp["tasks"])
    for cfgs in (co, cg, cr):
        for c in cfgs:
            c.use_internal_otg = 1
    o = ol.Oracle(ol.panda_model(), co, B, threads=8)
    g = pkg.Controller(pkg.panda_model(), cg, B)
    monkeypatch.setenv("SAI2B_NO_CERT_PATH", "1")
    ref = pkg.Controller(pkg.panda_model(), cr, B)
    monkeypatch.delenv("SAI2B_NO_CERT_PATH")
    rng = np.random.default_rng(3)
    for c in (o, g, ref):
        c.set_state(inp["q"], inp["dq"])
        c.reinitialize()
    goals = {1: o.get_jt_desired(1)[0] + rng.normal(0, 0.2, (2, B)), 2: o.get_jt_desired(2)[0] + rng.normal(0, 0.3, (N, B))}
    for tick in range(12):
        if tick in (0, 6):
            for t in (1, 2):
                goals[t] = goals[t] + rng.normal(0, 0.1, goals[t].shape)
                for c in (o, g, ref):
                    c.set_jt_goals(t, goals[t], None, None)
        tau_o, tau_g, tau_r = o.tick(), g.tick(), ref.tick()
        _, _, ro = o.get_mft_singularity(0)
        regular = ro == 3
        assert _err(tau_g, tau_o)[regular].max() < 10 * TOL and _err(tau_g, tau_o).max() < 1e-6, tick
        assert _err(tau_g, tau_r)[regular].max() < 10 * TOL
        for t in (1, 2):
            for a, b_, c_ in zip(o.get_jt_desired(t), g.get_jt_desired(t), ref.get_jt_desired(t)):
                assert np.array_equal(b_, c_), (tick, t)
                assert np.abs(a - b_).max() < 1e-12, (tick, t)
            for b_, c_ in zip(g.get_otg_status(t), ref.get_otg_status(t)):
                assert np.array_equal(b_, c_), (tick, t)
        for c in (o, g, ref):
            c.sim_step(tau_o, 0.001, 1, with_gravity=False)
        qo, vo = o.get_state()
        for c in (g, ref):
            c.set_state(qo, vo)
    assert g.fallback_count() < B // 4 and ref.fallback_count() == B  # the two really ran different kernels
