"""-m gpu: the JERK-LIMITED internal OTG (enableInternalOtgJerkLimited: reference JointTask.h:295-310, JointTask.cpp:
383-406, MotionForceTask.h:416, MotionForceTask.cpp:525-538) on the device (csrc/sai2b_otg3_core.hpp, sai2b_otg.hip:
sample_jerk / plan_lane3) against the oracle, whose jerk-limited planner IS the reference's own ruckig
(oracle/_ref/libruckig_ref.so, compiled from the reference's sources) under restated wrappers.

Tolerance: the host build of the device code is bit-equal to the reference's ruckig (tests/test_otg3_core.py); on the
device cbrt / acos / cos / sin / atan of the root finders come from another math library, so roots differ in the last
bits before ruckig's Newton steps pull them together: generator states <= 1e-9, torques <= 1e-8 relative (a planner
that picks another of several valid profiles would show as 1e-3)."""
import numpy as np
import pytest

import oracle_lib as ol
import sai2_primitives_perso_amd as pkg
from test_gpu_otg import _err, _random_goal_run

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not ol.lib().otg_jerk_planner_available(), reason="oracle/_ref/libruckig_ref.so not built")]


def _jerk(cfgs, which, jt=(np.pi / 3, 2 * np.pi, 10 * np.pi), mft=(0.3, 2.0, 10.0, np.pi / 3, 2 * np.pi, 10 * np.pi)):
    for t in which:
        c = cfgs[t]
        c.use_internal_otg, c.internal_otg_jerk_limited = 1, 1
        if c.type == pkg.JOINT_TASK:
            for i in range(c.task_dof):
                c.otg_max_velocity[i], c.otg_max_acceleration[i], c.otg_max_jerk[i] = jt
        else:
            (c.otg_max_linear_velocity, c.otg_max_linear_acceleration, c.otg_max_linear_jerk, c.otg_max_angular_velocity,
             c.otg_max_angular_acceleration, c.otg_max_angular_jerk) = mft
    return cfgs


def _pair(B, jerk_tasks=(0, 1), otg=(True, True)):
    to = [ol.motion_force_task("motion_force_task_0", internal_otg=otg[0]), ol.joint_task("joint_task_1", internal_otg=otg[1])]
    tg = [pkg.motion_force_task_config("motion_force_task_0", internal_otg=otg[0]), pkg.joint_task_config("joint_task_1", internal_otg=otg[1])]
    _jerk(to, jerk_tasks), _jerk(tg, jerk_tasks)
    return ol.Oracle(ol.panda_model(), to, B, threads=8), pkg.Controller(pkg.panda_model(), tg, B)


def test_gpu_jerk_limited_generators_random_regoals():
    """[MFT, JT], both generators jerk-limited, 256 robots re-goaling independently (idle / sampling / planning lanes
    side by side), 300 ticks"""
    B = 256
    inp = pkg.workloads.make_inputs(3, B=B, seed=15)
    o, g = _pair(B)
    w = _random_goal_run(o, g, inp, 300, np.random.default_rng(3), jt_task=1, mft_task=0, check_every=3)
    assert w["state"] < 1e-9 and w["tau"] < 1e-8, w


def test_gpu_jerk_limited_joint_generator_reaches_its_goal_smoothly():
    """one JointTask, jerk-limited: the acceleration the law receives is continuous (jumps bounded by max_jerk * dt,
    which the acceleration-limited generator violates by construction) and every robot arrives"""
    B = 128
    inp = pkg.workloads.make_inputs(3, B=B, seed=16)
    jt = (1.0, 4.0, 40.0)
    to, tg = _jerk([ol.joint_task("j", internal_otg=True)], [0], jt=jt), _jerk([pkg.joint_task_config("j", internal_otg=True)], [0], jt=jt)
    o, g = ol.Oracle(ol.panda_model(), to, B, threads=8), pkg.Controller(pkg.panda_model(), tg, B)
    for c in (o, g):
        c.set_state(inp["q"], np.zeros_like(inp["q"]))
        c.reinitialize()
    goal = inp["q"] + np.random.default_rng(2).normal(0, 0.25, inp["q"].shape)
    prev_acc, worst_jump = None, 0.0
    for tick in range(1500):
        for c in (o, g):
            c.set_jt_goals(0, goal, None, None)
        to_, tg_ = o.tick(), g.tick()
        assert _err(tg_, to_).max() < 1e-8, tick
        dq_o, dq_g = o.get_jt_desired(0), g.get_jt_desired(0)
        for a, b in zip(dq_o, dq_g):
            assert np.abs(a - b).max() < 1e-9, tick
        if prev_acc is not None:
            worst_jump = max(worst_jump, np.abs(dq_g[2] - prev_acc).max())
        prev_acc = dq_g[2].copy()
    assert worst_jump <= jt[2] * 1e-3 * (1 + 1e-9), worst_jump
    reached_o, reached_g = o.get_otg_status(0)[0], g.get_otg_status(0)[0]
    assert np.array_equal(reached_o, reached_g) and reached_g.all()
    assert np.abs(g.get_jt_desired(0)[0] - goal).max() < 1e-9


def test_gpu_switching_between_acceleration_and_jerk_limited_at_run_time():
    """enableInternalOtgJerkLimited / ...AccelerationLimited on a running generator re-initialise it at the task's current
    pose (JointTask.cpp:374,399; MotionForceTask.cpp:514,529); new jerk limits on a jerk-limited one only re-plan"""
    B = 96
    inp = pkg.workloads.make_inputs(3, B=B, seed=17)
    o, g = _pair(B, jerk_tasks=())
    for c in (o, g):
        c.set_state(inp["q"], np.zeros_like(inp["q"]))
        c.reinitialize()
    rng = np.random.default_rng(4)
    gq = inp["q"] + rng.normal(0, 0.2, inp["q"].shape)
    pos = o.get_mft_status(0)["pos"] + rng.uniform(-0.05, 0.05, (3, B))
    for tick in range(360):
        if tick in (60, 180, 300):
            gq = gq + rng.normal(0, 0.15, gq.shape)
            pos = pos + rng.uniform(-0.04, 0.04, (3, B))
        for c in (o, g):
            c.set_jt_goals(1, gq, None, None)
            c.set_mft_goals(0, pos, None, None, None, None, None)
        if tick in (40, 120, 200, 260):
            for c in (o, g):
                for t in (0, 1):
                    cfg = c.tasks[t]
                    if tick == 40:  # acceleration -> jerk-limited, while moving
                        _jerk([cfg], [0])
                    elif tick == 120:  # new jerk limits on a jerk-limited generator
                        _jerk([cfg], [0], jt=(np.pi / 2, 3 * np.pi, 25.0), mft=(0.4, 3.0, 20.0, np.pi / 2, 3 * np.pi, 25.0))
                    elif tick == 200:  # back to acceleration-limited
                        cfg.internal_otg_jerk_limited = 0
                    else:  # and to jerk-limited again with the first limits
                        _jerk([cfg], [0])
                    c.update_task_config(t, cfg)
        to_, tg_ = o.tick(), g.tick()
        assert _err(tg_, to_).max() < 1e-8, (tick, _err(tg_, to_).max())
        for a, b in zip(o.get_jt_desired(1), g.get_jt_desired(1)):
            assert np.abs(a - b).max() < 1e-9, tick
        for a, b in zip(o.get_mft_desired(0), g.get_mft_desired(0)):
            assert np.abs(a - b).max() < 1e-9, tick


def test_gpu_jerk_limited_partial_joint_task_in_a_three_level_hierarchy():
    """C4's hierarchy with the 2-DoF partial JointTask and the full one jerk-limited (2- and 7-DoF generators, gated)"""
    B = 192
    inp = pkg.workloads.make_inputs(4, B=B, seed=18)
    to, tg = ol.task_configs(inp["tasks"]), pkg.task_configs(inp["tasks"])
    for cfgs in (to, tg):
        cfgs[0].use_internal_otg = 1
        _jerk(cfgs, [1, 2])
    o, g = ol.Oracle(ol.panda_model(), to, B, threads=8), pkg.Controller(pkg.panda_model(), tg, B)
    for c in (o, g):
        c.set_state(inp["q"], inp["dq"])
        c.reinitialize()
        ol.load_inputs(c, inp)
    _, _, ns = o.get_mft_singularity(0) if False else (None, None, None)
    for tick in range(80):
        to_, tg_ = o.tick(), g.tick()
        e = _err(tg_, to_)
        assert np.median(e) < 1e-9 and e.max() < 1e-4, (tick, e.max())  # (robots inside a blending region: 1e-6 .. 1e-4)
        for t in (1, 2):
            for a, b in zip(o.get_jt_desired(t), g.get_jt_desired(t)):
                assert np.abs(a - b).max() < 1e-9, (tick, t)


def test_jerk_limits_are_validated():
    cfg = pkg.joint_task_config("j", internal_otg=True)
    cfg.internal_otg_jerk_limited = 1
    cfg.otg_max_jerk[3] = 0.0
    with pytest.raises(ValueError, match="max jerk"):
        pkg.Controller(pkg.panda_model(), [cfg], 64)
