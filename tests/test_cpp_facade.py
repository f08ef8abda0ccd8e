"""The C++ facade (include/Sai2PrimitivesBatched.h) compiled with g++ against the C ABI: argument
checks on CPU, one tick against the oracle on the GPU."""
import os
import subprocess

import numpy as np
import pytest

import sai2_primitives_perso_amd as pkg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sai2-primitives-perso_amd", "csrc")


@pytest.fixture(scope="module")
def facade_bin(tmp_path_factory):
    pkg._abi.load_library()  # make sure it exists
    out = str(tmp_path_factory.mktemp("cpp") / "facade_test")
    subprocess.run(
        ["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "facade_test.cpp"),
         "-o", out, "-L", CSRC, "-lsai2b", f"-Wl,-rpath,{CSRC}", "-Wl,-rpath,/opt/rocm/lib"],
        check=True,
    )
    return out


def test_cpp_facade_argument_checks(facade_bin):
    r = subprocess.run([facade_bin, "validate"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "validate: ok" in r.stdout


@pytest.mark.gpu
def test_cpp_facade_tick_matches_oracle(facade_bin, tmp_path):
    import oracle_lib as ol

    B = 256
    inp = pkg.workloads.make_inputs(3, B=B, seed=99)
    g = inp["mft0"]
    blob = np.concatenate([inp["q"].ravel(), inp["dq"].ravel(), g["pos"].ravel(), g["rot"].ravel(), g["v"].ravel(), g["w"].ravel(),
                           g["a"].ravel(), g["alpha"].ravel(), inp["jt1"]["q"].ravel()])
    path = tmp_path / "in.bin"
    blob.astype(np.float64).tofile(path)
    r = subprocess.run([facade_bin, "tick", str(B), str(path)], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    out = np.frombuffer(r.stdout, dtype=np.float64).reshape(3, 7, B)
    tau, q1, dq1 = out
    o = ol.Oracle(ol.panda_model(), ol.task_configs(inp["tasks"]), B)
    ol.load_inputs(o, inp)
    ref = o.tick()
    assert (np.abs(tau - ref).max(axis=0) / np.maximum(np.abs(ref).max(axis=0), 1)).max() < 1e-10
    # BatchedSimulation::integrate() consumed those torques on the device (2 sub-steps of 0.5 ms)
    o.sim_step(ref, 0.001, substeps=2)
    qo, vo = o.get_state()
    assert np.abs(q1 - qo).max() < 1e-12 and np.abs(dq1 - vo).max() < 1e-10


@pytest.mark.gpu
def test_cpp_sharded_controller_equals_one_context(facade_bin, tmp_path):
    """ShardedRobotController (one context + one host thread per shard, contiguous slices, no collective: SURVEY §8(e)):
    2 and 3 shards on device 0 — an uneven split of 257 robots — give bit for bit the torques of one context of the
    whole batch, and those of the oracle to 1e-10"""
    import oracle_lib as ol

    B = 257
    inp = pkg.workloads.make_inputs(3, B=B, seed=1234)
    g = inp["mft0"]
    blob = np.concatenate([inp["q"].ravel(), inp["dq"].ravel(), g["pos"].ravel(), g["rot"].ravel(), g["v"].ravel(), g["w"].ravel(),
                           g["a"].ravel(), g["alpha"].ravel(), inp["jt1"]["q"].ravel()])
    path = tmp_path / "in.bin"
    blob.astype(np.float64).tofile(path)
    out = {}
    for shards in (1, 2, 3):
        r = subprocess.run([facade_bin, "sharded", str(B), str(path), str(shards)], capture_output=True)
        assert r.returncode == 0, (shards, r.returncode, r.stderr.decode())
        out[shards] = np.frombuffer(r.stdout, dtype=np.float64).reshape(2, 7, B)
    assert np.array_equal(out[1], out[2]) and np.array_equal(out[1], out[3])
    o = ol.Oracle(ol.panda_model(), ol.task_configs(inp["tasks"]), B)
    ol.load_inputs(o, inp)
    for k in range(2):
        ref = o.tick()
        assert _err(out[2][k], ref) < 1e-10


def _err(a, ref):
    return (np.abs(a - ref).max(axis=0) / np.maximum(np.abs(ref).max(axis=0), 1)).max()


@pytest.mark.gpu
def test_cpp_example_04_task_and_redundancy(facade_bin, tmp_path):
    """tests/cpp/facade_test.cpp::example04 = examples/04-task_and_redundancy.cpp:101-206 call for call (TemplateTask
    virtuals, no RobotController); the same calls on the oracle, period by period in closed loop"""
    import oracle_lib as ol

    B, ticks = 64, 24
    inp = pkg.workloads.make_inputs(3, B=B, seed=404)
    path = tmp_path / "q.bin"
    inp["q"].astype(np.float64).tofile(path)
    r = subprocess.run([facade_bin, "example04", str(B), str(path), str(ticks)], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    out = np.frombuffer(r.stdout, dtype=np.float64).reshape(ticks + 1, 7, B)
    o = ol.Oracle(ol.panda_model(), [ol.motion_force_task("motion_force_task"), ol.joint_task("joint_task", internal_otg=True)], B, threads=8)
    o.set_state(inp["q"], np.zeros_like(inp["q"]))
    o.reinitialize()
    st = o.get_mft_status(0)
    x0, R0 = st["pos"], st["rot"].reshape(3, 3, B)
    wake = ticks // 3
    for cycle in range(ticks):
        t = 0.001 * cycle
        o.task_update_model(0, None)
        o.task_update_model(1, o.task_nullspaces(0)[2])
        w_ori, amp = 2 * np.pi * 0.2, np.pi / 8
        ang = amp * np.sin(w_ori * t)
        c, s = np.cos(ang), np.sin(ang)
        Rt = np.array([[c, 0, -s], [0, 1, 0], [s, 0, c]])
        Rg = np.einsum("ik,kjb->ijb", Rt, R0).reshape(9, B)
        wg, ag = np.zeros((3, B)), np.zeros((3, B))
        wg[1], ag[1] = amp * w_ori * np.cos(w_ori * t), amp * w_ori * w_ori * -np.sin(w_ori * t)
        r_, wc = 0.05, 2 * np.pi * 0.33
        dp = np.array([0.0, np.sin(wc * t), 1 - np.cos(wc * t)])[:, None]
        dv = np.array([0.0, np.cos(wc * t), np.sin(wc * t)])[:, None]
        da = np.array([0.0, -np.sin(wc * t), np.cos(wc * t)])[:, None]
        o.set_mft_goals(0, x0 + r_ * dp, Rg, np.broadcast_to(r_ * wc * dv, (3, B)), wg, np.broadcast_to(r_ * wc * wc * da, (3, B)), ag)
        t0, t1 = o.task_compute_torques(0), o.task_compute_torques(1)
        if cycle < wake:
            t1 = np.zeros_like(t1)
        if cycle == wake:
            o.task_reinitialize(1)
            g = inp["q"].copy()
            g[0] += 1.5
            o.set_jt_goals(1, g)
        tau = t0 + t1
        assert _err(out[cycle], tau) < 1e-9, (cycle, _err(out[cycle], tau))
        o.sim_step(tau, 0.001, 1)
    assert np.abs(out[ticks] - o.get_state()[0]).max() < 1e-10


@pytest.mark.gpu
def test_cpp_example_01_joint_control(facade_bin, tmp_path):
    """tests/cpp/facade_test.cpp::example01 = examples/01-joint_control.cpp:123-191 call for call (BASELINE config 1)"""
    import oracle_lib as ol

    B, ticks = 64, 60
    inp = pkg.workloads.make_inputs(3, B=B, seed=101)
    path = tmp_path / "q.bin"
    inp["q"].astype(np.float64).tofile(path)
    r = subprocess.run([facade_bin, "example01", str(B), str(path), str(ticks)], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    out = np.frombuffer(r.stdout, dtype=np.float64).reshape(ticks + 1, 7, B)
    cfg = ol.joint_task("joint_task", internal_otg=True)
    o = ol.Oracle(ol.panda_model(), [cfg], B, threads=8)
    o.set_state(inp["q"], np.zeros_like(inp["q"]))
    o.reinitialize()

    def gains(kp, kv):
        for i in range(7):
            cfg.kp[i], cfg.kv[i], cfg.ki[i] = kp, kv, 0.0
        o.update_task_config(0, cfg)

    gains(100, 20)
    goal = inp["q"].copy()
    cfg.use_internal_otg = 0
    o.update_task_config(0, cfg)
    eye = np.repeat(np.eye(7).reshape(49, 1), B, axis=1)
    for cycle in range(ticks):
        o.task_update_model(0, eye)
        if cycle % 30 == 5:
            goal[2] += 0.4
            goal[3] -= 0.6
        if cycle % 30 == 20:
            goal[2] -= 0.4
            goal[3] += 0.6
        o.set_jt_goals(0, goal)
        if cycle == 35:
            gains(100, 10)
        if cycle == 45:
            cfg.use_velocity_saturation = 1
            for i in range(7):
                cfg.saturation_velocity[i] = np.pi / 4
            o.update_task_config(0, cfg)
        if cycle == 55:
            gains(100, 20)
        tau = o.task_compute_torques(0)
        assert _err(out[cycle], tau) < 1e-9, (cycle, _err(out[cycle], tau))
        o.sim_step(tau, 0.001, 1)
    assert np.abs(out[ticks] - o.get_state()[0]).max() < 1e-10


@pytest.mark.gpu
def test_cpp_example_06_partial_joint_task_on_the_sliding_base(facade_bin, tmp_path):
    """tests/cpp/facade_test.cpp::example06 = examples/06-partial_joint_task.cpp:99-176 call for call: an 8-joint robot with
    a prismatic base read from a URDF, RobotController over [partial JointTask (internal OTG on), MotionForceTask]"""
    import oracle_lib as ol
    import robots

    B, ticks = 64, 60
    urdf = tmp_path / "sliding_base.urdf"
    urdf.write_text(robots.TEXT["sliding_base"]())
    m, links = pkg.model_from_urdf(str(urdf))
    n = m.dof
    rng = np.random.default_rng(6)
    lo, hi = np.array(list(m.q_lower)[:n]), np.array(list(m.q_upper)[:n])
    q0 = (0.5 * (lo + hi))[:, None] + 0.5 * (0.5 * (hi - lo))[:, None] * rng.uniform(-1, 1, (n, B))
    path = tmp_path / "q.bin"
    q0.astype(np.float64).tofile(path)
    r = subprocess.run([facade_bin, "example06", str(B), str(urdf), str(path), str(ticks)], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    out = np.frombuffer(r.stdout, dtype=np.float64).reshape(ticks + 1, n, B)
    sel = np.zeros((2, n))
    sel[0, 0] = sel[1, 7] = 1
    link, fpos, frot = pkg.resolve_link_frame(links, "end-effector", (0.0, 0.0, 0.07))
    cfg = [ol.joint_task("partial_joint_task", sel, internal_otg=True, robot_dof=n),
           ol.motion_force_task("motion_force_task", link, fpos, frot, robot_dof=n)]
    o = ol.Oracle(m, cfg, B, threads=8)
    o.set_state(q0, np.zeros_like(q0))
    o.reinitialize()
    goal = o.get_jt_goals(0)[0] if hasattr(o, "get_jt_goals") else sel @ q0
    x0 = o.get_mft_status(1)["pos"]
    for cycle in range(ticks):
        o.update_task_models()
        if cycle % 40 == 10:
            goal[0] -= 1.0
        elif cycle % 40 == 30:
            goal[0] += 1.0
        o.set_jt_goals(0, goal)
        t, w = 0.001 * cycle, 2.0 * np.pi * 0.3
        dp, dv, da = o.get_jt_desired(0)
        gp, gv, ga = np.empty((3, B)), np.empty((3, B)), np.empty((3, B))
        gp[0], gp[1], gp[2] = x0[0] + 0.1 * np.sin(w * t), dp[0], x0[2] + 0.1 * (1 - np.cos(w * t))
        gv[0], gv[1], gv[2] = 0.1 * w * np.cos(w * t), dv[0], 0.1 * w * np.sin(w * t)
        ga[0], ga[1], ga[2] = -0.1 * w * w * np.sin(w * t), da[0], 0.1 * w * w * np.cos(w * t)
        o.set_mft_goals(1, gp, None, gv, None, ga, None)
        tau = o.compute_control_torques(True)
        # the last joint belongs to the joint task above, so the 6-DOF motion task has lost a direction for every robot:
        # it runs inside its singularity handling, where two FP64 implementations agree to ~1e-6
        assert _err(out[cycle], tau) < 1e-6, (cycle, _err(out[cycle], tau))
        o.sim_step(tau, 0.001, 2)
    assert np.abs(out[ticks] - o.get_state()[0]).max() < 1e-7


@pytest.mark.gpu
def test_cpp_example_18_panda_singularity(facade_bin, tmp_path):
    """tests/cpp/facade_test.cpp::example18 = examples/18-panda_singularity.cpp:104-228 call for call: position goals
    2 m outside the workspace drive the arm into its elbow / wrist singularities and back, with velocity saturation;
    most robots are inside a singularity-blending region most of the time (the reference's default thresholds), the
    handler classifies type 1 / type 2 and blends. The program prints the state it read and the torques it computed
    every period; the oracle gets the same state and makes the same calls (its controller state — singularity
    history, generator — is its own throughout)."""
    import oracle_lib as ol

    B, ticks = 32, 2400
    rng = np.random.default_rng(18)
    q0 = np.array([0.0, 0.2, 0.0, -1.3, 0.0, 1.6, 0.4])[:, None] + rng.normal(0, 0.08, (7, B))
    path = tmp_path / "q.bin"
    np.ascontiguousarray(q0).tofile(path)
    r = subprocess.run([facade_bin, "example18", str(B), str(path), str(ticks)], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    out = np.frombuffer(r.stdout, dtype=np.float64).reshape(ticks, 3, 7, B)
    mcfg = ol.motion_force_task("motion_force_task")
    mcfg.use_velocity_saturation = 1
    jcfg = ol.joint_task("joint_task", internal_otg=True)
    for i in range(7):
        jcfg.kp[i], jcfg.kv[i], jcfg.ki[i] = 100, 20, 0
    o = ol.Oracle(ol.panda_model(), [mcfg, jcfg], B, threads=8)
    o.set_state(q0, np.zeros_like(q0))
    o.reinitialize()
    x0 = o.get_mft_status(0)["pos"].copy()
    o.set_jt_goals(1, q0)
    offs = [(2, 0, 0), (0, 0, 0), (0, 2, 0), (0, 0, 0), (0, -2, 0), (0, 0, 0), (0, 0, 2), (0, 0, 0)]
    wait, cnt, prev = ticks // 8, 0, -(ticks // 8)
    worst_regular, worst_singular, singular_periods, big = 0.0, 0.0, 0, 0
    for cycle in range(ticks):
        q, dq, tau_g = out[cycle]
        o.set_state(q, dq)
        o.task_update_model(0, None)
        o.task_update_model(1, o.task_nullspaces(0)[2])
        if cycle - prev >= wait:
            o.set_mft_goals(0, x0 + np.array(offs[cnt], dtype=float)[:, None], None, None, None, None, None)
            cnt, prev = min(cnt + 1, 7), cycle
        tau = o.task_compute_torques(0) + o.task_compute_torques(1)
        _, _, ro = o.get_mft_singularity(0)
        sing = ro < 6
        e = np.abs(tau_g - tau).max(axis=0) / np.maximum(np.abs(tau).max(axis=0), 1)
        if (~sing).any():
            worst_regular = max(worst_regular, e[~sing].max())
        if sing.any():
            worst_singular = max(worst_singular, e[sing].max())
            singular_periods += int(sing.sum())
            big += int((e[sing] > 1e-5).sum())
    print("example18:", worst_regular, worst_singular, singular_periods, big)
    assert singular_periods > ticks * B // 4  # the scenario does what it is for
    assert worst_regular < 1e-9, worst_regular
    # (measured: 57 204 robot-periods inside a blending region, worst 2.7e-13 there, none above 1e-5)
    assert worst_singular < 1e-6 and big == 0, (big, singular_periods, worst_singular)


@pytest.mark.gpu
def test_cpp_example_11_planar_robot_controller(facade_bin, tmp_path):
    """tests/cpp/facade_test.cpp::example11 = examples/11-planar_robot_controller.cpp:99-166 call for call: the planar 4R
    from its URDF, partial MotionForceTask (x, y, rotation about z) on a link given by NAME + JointTask in a
    RobotController, the reference's default internal OTG of both tasks left ON (the goal steps are followed along
    generated trajectories). The oracle gets the state the program read each period and makes the same calls."""
    import oracle_lib as ol
    import robots

    B, ticks = 64, 400
    urdf = tmp_path / "rrrrbot.urdf"
    urdf.write_text(robots.TEXT["planar_4r"]())
    m, links = pkg.model_from_urdf(str(urdf))
    n = m.dof
    rng = np.random.default_rng(11)
    q0 = np.array([0.4, -0.7, 0.9, -0.6])[:, None] + rng.normal(0, 0.15, (n, B))
    path = tmp_path / "q.bin"
    np.ascontiguousarray(q0).tofile(path)
    r = subprocess.run([facade_bin, "example11", str(B), str(urdf), str(path), str(ticks)], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    out = np.frombuffer(r.stdout, dtype=np.float64).reshape(ticks, 3, n, B)
    link, fpos, frot = pkg.resolve_link_frame(links, "link4", (0.5, 0.0, 0.0))
    partial = (np.array([[1.0, 0, 0], [0, 1.0, 0]]), np.array([[0, 0, 1.0]]))
    cfg = [ol.motion_force_task("partial_motion_force_task", link, fpos, frot, partial, internal_otg=True, robot_dof=n),
           ol.joint_task("joint_task", None, internal_otg=True, robot_dof=n)]
    o = ol.Oracle(m, cfg, B, threads=8)
    o.set_state(q0, np.zeros_like(q0))
    o.reinitialize()
    st = o.get_mft_status(0)
    x0, R0 = st["pos"].copy(), st["rot"].reshape(3, 3, B).copy()
    c, s = np.cos(-np.pi / 4), np.sin(-np.pi / 4)
    Rz = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
    gp, gR = x0, R0.reshape(9, B)
    worst, moved = 0.0, 0.0
    for cycle in range(ticks):
        q, dq, tau_g = out[cycle]
        o.set_state(q, dq)
        o.update_task_models()
        if cycle % ticks == 0:
            gp, gR = x0, R0.reshape(9, B)
        elif cycle % ticks == ticks // 2:
            gp = x0 - np.array([0.25, 0.25, 0.0])[:, None]
            gR = np.einsum("ik,kjb->ijb", Rz, R0).reshape(9, B)
        o.set_mft_goals(0, gp, np.ascontiguousarray(gR), None, None, None, None)
        tau = o.compute_control_torques(True)
        _, _, ro = o.get_mft_singularity(0)
        reg = ro == 3
        e = np.abs(tau_g - tau).max(axis=0) / np.maximum(np.abs(tau).max(axis=0), 1)
        worst = max(worst, e[reg].max(initial=0.0))
        assert e.max() < 1e-5, (cycle, e.max())
        moved = max(moved, np.abs(q - q0).max())
    assert worst < 1e-9, worst
    assert moved > 0.1  # the generators carried the arm towards the stepped goal


@pytest.mark.gpu
def test_cpp_example_02_joint_control_internal_otg(facade_bin, tmp_path):
    """tests/cpp/facade_test.cpp::example02 = examples/02-joint_control_internal_otg.cpp:118-179 call for call: a JointTask
    following goal steps along acceleration-limited trajectories, limits raised in mid-run (re-planning of moving
    generators), then jerk limits added (enableInternalOtgJerkLimited: ruckig's third-order interface) and two more goal
    steps under them. On the oracle's side the jerk-limited planner is the reference's own ruckig (oracle/_ref)."""
    import oracle_lib as ol

    if not ol.lib().otg_jerk_planner_available():
        pytest.skip("oracle/_ref/libruckig_ref.so not built")
    B, ticks = 64, 960
    inp = pkg.workloads.make_inputs(3, B=B, seed=202)
    path = tmp_path / "q.bin"
    inp["q"].astype(np.float64).tofile(path)
    r = subprocess.run([facade_bin, "example02", str(B), str(path), str(ticks)], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    out = np.frombuffer(r.stdout, dtype=np.float64).reshape(ticks, 3, 7, B)
    cfg = ol.joint_task("joint_task", internal_otg=True)
    for i in range(7):
        cfg.kp[i], cfg.kv[i], cfg.ki[i] = 100, 20, 0
        cfg.otg_max_velocity[i], cfg.otg_max_acceleration[i] = np.pi / 3, np.pi
    o = ol.Oracle(ol.panda_model(), [cfg], B, threads=8)
    o.set_state(inp["q"], np.zeros_like(inp["q"]))
    o.reinitialize()
    goal = inp["q"].copy()
    eye = np.repeat(np.eye(7).reshape(49, 1), B, axis=1)
    u = ticks // 12
    period = 4 * u
    moved = moved_jerk = 0.0
    for cycle in range(ticks):
        q, dq, tau_g = out[cycle]
        o.set_state(q, dq)
        o.task_update_model(0, eye)
        if cycle % period == u:
            goal[1] -= 0.2
            goal[2] += 0.4
            goal[3] -= 0.6
        if cycle % period == 3 * u:
            goal[1] += 0.2
            goal[2] -= 0.4
            goal[3] += 0.6
        o.set_jt_goals(0, goal)
        if cycle == 5 * u:
            for i in range(7):
                cfg.otg_max_velocity[i], cfg.otg_max_acceleration[i] = np.pi, 3 * np.pi
            o.update_task_config(0, cfg)
        if cycle == 10 * u:
            cfg.internal_otg_jerk_limited = 1
            for i in range(7):
                cfg.otg_max_velocity[i], cfg.otg_max_acceleration[i], cfg.otg_max_jerk[i] = np.pi, 3 * np.pi, 3 * np.pi
            o.update_task_config(0, cfg)
            q_at_switch = q.copy()
        tau = o.task_compute_torques(0)
        assert _err(tau_g, tau) < 1e-9, (cycle, _err(tau_g, tau))
        moved = max(moved, np.abs(q - inp["q"]).max())
        if cycle > 10 * u:
            moved_jerk = max(moved_jerk, np.abs(q - q_at_switch).max())
    assert moved > 0.2  # the generators carried the joints towards the stepped goals
    assert moved_jerk > 0.01  # ... also along the jerk-limited trajectories of the last phase


@pytest.mark.gpu
def test_cpp_example_03_cartesian_motion_control(facade_bin, tmp_path):
    """tests/cpp/facade_test.cpp::example03 = examples/03-cartesian_motion_control.cpp:109-183 call for call: a
    MotionForceTask following position + orientation goal steps along the Cartesian generator's trajectories
    (new goals while still moving), the generator switched off in mid-run and back on with jerk limits
    (enableInternalOtgJerkLimited) for the last goal steps. On the oracle's side the jerk-limited planner is the
    reference's own ruckig (oracle/_ref)."""
    import oracle_lib as ol

    if not ol.lib().otg_jerk_planner_available():
        pytest.skip("oracle/_ref/libruckig_ref.so not built")
    B, ticks = 64, 1200
    inp = pkg.workloads.make_inputs(3, B=B, seed=303)
    path = tmp_path / "q.bin"
    inp["q"].astype(np.float64).tofile(path)
    r = subprocess.run([facade_bin, "example03", str(B), str(path), str(ticks)], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    out = np.frombuffer(r.stdout, dtype=np.float64).reshape(ticks, 3, 7, B)
    cfg = ol.motion_force_task("motion_force_task", frame_pos=(0.07, 0.0, 0.15), internal_otg=True)
    for i in range(3):
        cfg.kp_pos[i], cfg.kv_pos[i], cfg.ki_pos[i] = 100.0, 20.0, 0.0
        cfg.kp_ori[i], cfg.kv_ori[i], cfg.ki_ori[i] = 100.0, 20.0, 0.0
    o = ol.Oracle(ol.panda_model(), [cfg], B, threads=8)
    o.set_state(inp["q"], np.zeros_like(inp["q"]))
    o.reinitialize()
    st = o.get_mft_status(0)
    gp, gR = st["pos"].copy(), st["rot"].reshape(3, 3, B).copy()
    th = np.pi / 4
    R = np.array([[np.cos(th), np.sin(th), 0], [-np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    u = ticks // 30
    period = 6 * u
    worst_regular, moved = 0.0, 0.0
    for cycle in range(ticks):
        q, dq, tau_g = out[cycle]
        o.set_state(q, dq)
        o.task_update_model(0, None)
        if cycle % period == 4 * u:
            gp[2] += 0.1
            gR = np.einsum("ik,kjb->ijb", R, gR)
        elif cycle % period == u:
            gp[2] -= 0.1
            gR = np.einsum("ki,kjb->ijb", R, gR)
        o.set_mft_goals(0, gp, np.ascontiguousarray(gR.reshape(9, B)), None, None, None, None)
        if cycle == 13 * u:
            cfg.use_internal_otg = 0
            o.update_task_config(0, cfg)
        if cycle == 25 * u:
            cfg.use_internal_otg, cfg.internal_otg_jerk_limited = 1, 1
            cfg.otg_max_linear_velocity, cfg.otg_max_linear_acceleration, cfg.otg_max_linear_jerk = 0.3, 1.0, 3.0
            cfg.otg_max_angular_velocity, cfg.otg_max_angular_acceleration, cfg.otg_max_angular_jerk = np.pi / 3, np.pi, 3 * np.pi
            o.update_task_config(0, cfg)
        tau = o.task_compute_torques(0)
        _, _, ro = o.get_mft_singularity(0)
        e = np.abs(tau_g - tau).max(axis=0) / np.maximum(np.abs(tau).max(axis=0), 1)
        assert e.max() < 1e-5, (cycle, e.max())
        worst_regular = max(worst_regular, e[ro == 6].max(initial=0.0))
        moved = max(moved, np.abs(q - inp["q"]).max())
    assert worst_regular < 1e-9, worst_regular
    assert moved > 0.1


@pytest.mark.gpu
def test_cpp_example_09_position_then_force_control(facade_bin, tmp_path):
    """tests/cpp/facade_test.cpp::example09 = examples/09-3d_position_force_controller.cpp:106-206 call for call (on the
    Panda, contact = a virtual spring, the switch to force control when every robot of the batch reports contact: a
    task's configuration is one per controller): updateSensedForceAndMoment every period, then
    parametrizeForceMotionSpaces / setGoalForce / setClosedLoopForceControl / enablePassivity in mid-run."""
    import oracle_lib as ol

    B, ticks = 32, 720
    inp = pkg.workloads.make_inputs(3, B=B, seed=909)
    path = tmp_path / "q.bin"
    inp["q"].astype(np.float64).tofile(path)
    r = subprocess.run([facade_bin, "example09", str(B), str(path), str(ticks)], capture_output=True)
    assert r.returncode == 0, (r.returncode, r.stderr.decode())
    out = np.frombuffer(r.stdout, dtype=np.float64).reshape(ticks, 25, B)
    partial = (np.eye(3), np.zeros((0, 3)))
    mcfg = ol.motion_force_task("partial_motion_force_task", frame_pos=(0.0, 0.0, 0.15), partial=partial, internal_otg=True)
    jcfg = ol.joint_task("joint_task", internal_otg=True)
    o = ol.Oracle(ol.panda_model(), [mcfg, jcfg], B, threads=8)
    o.set_state(inp["q"], np.zeros_like(inp["q"]))
    o.reinitialize()
    goal = o.get_mft_status(0)["pos"].copy()
    k2000, k1000, k3000 = ticks // 3, ticks // 6, ticks // 2
    force_control, switched_at = False, None
    worst_regular = 0.0
    for cycle in range(ticks):
        q, dq, sensed, flag, tau_g = out[cycle, :7], out[cycle, 7:14], out[cycle, 14:17], out[cycle, 17], out[cycle, 18:]
        o.set_state(q, dq)
        o.update_task_models()
        o.set_mft_sensed_wrench(0, np.ascontiguousarray(sensed), np.zeros((3, B)))
        if cycle % k2000 == 0:
            goal[0] -= 0.07
            goal[1] -= 0.07
        elif cycle % k2000 == k1000:
            goal[0] += 0.07
            goal[1] += 0.07
        if k2000 < cycle < k3000:
            goal[2] -= 0.00015
        o.set_mft_goals(0, goal, None, None, None, None, None)
        if not force_control and flag[0] == 1.0:
            force_control, switched_at = True, cycle
            mcfg.force_space_dimension = 1
            mcfg.force_axis[0], mcfg.force_axis[1], mcfg.force_axis[2] = 0.0, 0.0, 1.0
            o.update_task_config(0, mcfg)
            gf = np.zeros((3, B))
            gf[2] = -5.0
            o.set_mft_goal_wrench(0, gf, np.zeros((3, B)))
            mcfg.closed_loop_force = 1
            o.update_task_config(0, mcfg)
            mcfg.passivity_enabled = 1
            o.update_task_config(0, mcfg)
        tau = o.compute_control_torques(True)
        _, _, ro = o.get_mft_singularity(0)
        e = np.abs(tau_g - tau).max(axis=0) / np.maximum(np.abs(tau).max(axis=0), 1)
        assert e.max() < 1e-5, (cycle, e.max())
        worst_regular = max(worst_regular, e[ro == 3].max(initial=0.0))
    assert switched_at is not None and k2000 < switched_at < ticks - 50, switched_at  # contact, then a stretch of force control
    assert worst_regular < 1e-9, worst_regular


@pytest.mark.gpu
def test_cpp_example_19_six_r_wrist_singularity(facade_bin, tmp_path):
    """tests/cpp/facade_test.cpp::example19 = examples/19-puma_singularity.cpp:129-270 call for call (wrist-lock variant)
    on the 6R fixture robot (6-joint build of the library): a 6-DOF MotionForceTask started inside the wrist
    singularity with the internal OTG on, JointTask behind it (no range left), nullspaces chained by hand."""
    import oracle_lib as ol
    import robots

    B, ticks = 32, 900
    urdf = tmp_path / "six_r.urdf"
    urdf.write_text(robots.TEXT["six_r"]())
    m, links = pkg.model_from_urdf(str(urdf))
    n = m.dof
    rng = np.random.default_rng(19)
    q0 = np.array([0.3, -1.2, 1.9, 0.4, 0.0, 0.5])[:, None] + rng.normal(0, 0.05, (n, B))
    q0[4] = rng.normal(0, 0.01, B)  # wrist lock: the axes of joints 4 and 6 aligned
    dq0 = rng.normal(0, 0.05, (n, B))  # (at rest on its goal the task force is rounding noise, see below)
    path = tmp_path / "q.bin"
    np.concatenate([q0.ravel(), dq0.ravel()]).tofile(path)
    r = subprocess.run([facade_bin, "example19", str(B), str(urdf), str(path), str(ticks)], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    out = np.frombuffer(r.stdout, dtype=np.float64).reshape(ticks, 3, n, B)
    link, fpos, frot = pkg.resolve_link_frame(links, "link6", (0.0, 0.0, 0.0))
    mcfg = ol.motion_force_task("motion_force_task", link, fpos, frot, internal_otg=True, robot_dof=n)
    mcfg.kp_type_1, mcfg.kv_type_1, mcfg.kv_type_2 = 50.0, 20.0, 20.0
    jcfg = ol.joint_task("joint_task", None, internal_otg=True, robot_dof=n)
    for i in range(n):
        jcfg.kp[i], jcfg.kv[i], jcfg.ki[i] = 100, 20, 0
    o = ol.Oracle(m, [mcfg, jcfg], B, threads=8)
    o.set_state(q0, dq0)
    o.reinitialize()
    o.set_jt_goals(1, q0)
    five, start, cnt, state = ticks // 6, 0, 0, 0
    p0 = R0 = None

    def away():
        p = p0.copy()
        p[2] -= 0.2
        R = R0.reshape(3, 3, B)
        o.set_mft_goals(0, p, np.ascontiguousarray(np.stack([R[:, 1], -R[:, 0], R[:, 2]], axis=1).reshape(9, B)), None, None, None, None)

    worst_regular, singular_periods, skipped = 0.0, 0, 0
    for cycle in range(ticks):
        q, dq, tau_g = out[cycle]
        o.set_state(q, dq)
        if cycle - start > five and state == 0:
            st = o.get_mft_status(0)
            p0, R0 = st["pos"].copy(), st["rot"].copy()
            away()
            state, start = 1, cycle
        if cycle - start > five and state == 1:
            if cnt in (0, 2):
                o.set_mft_goals(0, p0, R0, None, None, None, None)
            else:
                away()
            start, cnt = cycle, (cnt + 1) % 4
        o.task_update_model(0, None)
        o.task_update_model(1, o.task_nullspaces(0)[2])
        tau = o.task_compute_torques(0) + o.task_compute_torques(1)
        _, _, ro = o.get_mft_singularity(0)
        sing = ro < 6
        e = np.abs(tau_g - tau).max(axis=0) / np.maximum(np.abs(tau).max(axis=0), 1)
        # The type-2 strategy scales a fixed torque by the DIRECTION of the task force (SingularityHandler.cpp:339-346,
        # F / |F|): for a robot resting on its goal inside the singularity that force is rounding noise (1e-16 of
        # position error) and its direction — hence an O(0.01 Nm) torque — is arbitrary on either side. Those
        # robot-periods are the reference's own ill-posed corner, counted and left out.
        Fu, Ff = o.get_mft_task_forces(0)
        noise = sing & (np.linalg.norm(Fu + Ff, axis=0) < 1e-6)
        skipped += int(noise.sum())
        assert e[~noise].max(initial=0.0) < 1e-5, (cycle, e[~noise].max())
        worst_regular = max(worst_regular, e[~sing].max(initial=0.0))
        singular_periods += int((sing & ~noise).sum())
    assert worst_regular < 1e-9, worst_regular
    assert singular_periods > ticks * B // 10 and skipped < ticks * B // 10, (singular_periods, skipped)


@pytest.mark.gpu
def test_cpp_example_07_surface_surface_contact(facade_bin, tmp_path):
    """tests/cpp/facade_test.cpp::example07 = examples/07-surface_surface_contact.cpp:124-228 call for call (on the Panda,
    virtual contact wrench, batch-wide switch): a MotionForceTask parametrised in its compliant frame with the passivity
    observer on and the force sensor at the link origin; after contact: force control along the frame's z, moment
    control about its x and y, closed loop both, new force / moment gains."""
    import oracle_lib as ol

    B, ticks = 32, 500
    inp = pkg.workloads.make_inputs(3, B=B, seed=707)
    path = tmp_path / "q.bin"
    inp["q"].astype(np.float64).tofile(path)
    r = subprocess.run([facade_bin, "example07", str(B), str(path), str(ticks)], capture_output=True)
    assert r.returncode == 0, (r.returncode, r.stderr.decode())
    out = np.frombuffer(r.stdout, dtype=np.float64).reshape(ticks, 28, B)
    cfg = ol.motion_force_task("surface_alignment_task", frame_pos=(0.0, 0.0, 0.22), internal_otg=False)
    cfg.parametrization_in_compliant_frame = 1
    cfg.passivity_enabled = 1
    cfg.sensor_pos[0], cfg.sensor_pos[1], cfg.sensor_pos[2] = 0.0, 0.0, -0.22  # compliant_frame^-1 * identity
    o = ol.Oracle(ol.panda_model(), [cfg], B, threads=8)
    o.set_state(inp["q"], np.zeros_like(inp["q"]))
    o.reinitialize()
    goal = o.get_mft_status(0)["pos"].copy()
    contact, switched_at, worst_regular = False, None, 0.0
    for cycle in range(ticks):
        q, dq, sf, sm, flag, tau_g = (out[cycle, :7], out[cycle, 7:14], out[cycle, 14:17], out[cycle, 17:20], out[cycle, 20],
                                      out[cycle, 21:])
        o.set_state(q, dq)
        o.set_mft_sensed_wrench(0, np.ascontiguousarray(sf), np.ascontiguousarray(sm))
        o.update_task_models()
        if not contact:
            goal[2] -= 0.00003
            o.set_mft_goals(0, goal, None, None, None, None, None)
            if flag[0] == 1.0:
                contact, switched_at = True, cycle
                cfg.force_space_dimension = 1
                cfg.force_axis[0], cfg.force_axis[1], cfg.force_axis[2] = 0.0, 0.0, 1.0
                o.update_task_config(0, cfg)
                cfg.moment_space_dimension = 2
                cfg.moment_axis[0], cfg.moment_axis[1], cfg.moment_axis[2] = 0.0, 0.0, 1.0
                o.update_task_config(0, cfg)
                cfg.closed_loop_force = 1
                o.update_task_config(0, cfg)
                cfg.closed_loop_moment = 1
                o.update_task_config(0, cfg)
                gf = np.zeros((3, B))
                gf[2] = 10.0
                o.set_mft_goal_wrench(0, gf, np.zeros((3, B)))
                for i in range(3):
                    cfg.kp_force[i], cfg.kv_force[i], cfg.ki_force[i] = 0.7, 5.0, 1.5
                o.update_task_config(0, cfg)
                for i in range(3):
                    cfg.kp_moment[i], cfg.kv_moment[i], cfg.ki_moment[i] = 0.7, 4.0, 1.5
                o.update_task_config(0, cfg)
        tau = o.compute_control_torques(True)
        _, _, ro = o.get_mft_singularity(0)
        e = np.abs(tau_g - tau).max(axis=0) / np.maximum(np.abs(tau).max(axis=0), 1)
        assert e.max() < 1e-5, (cycle, e.max())
        worst_regular = max(worst_regular, e[ro == 6].max(initial=0.0))
    assert switched_at is not None and switched_at < ticks - 100, switched_at
    assert worst_regular < 1e-9, worst_regular


@pytest.mark.gpu
def test_cpp_example_08_partial_motion_force_task(facade_bin, tmp_path):
    """tests/cpp/facade_test.cpp::example08 = examples/08-partial_motion_force_task.cpp:106-200 call for call: a partial
    MotionForceTask whose projection is not a leading block (y, z, rotation about x) + JointTask in a RobotController
    (the SVD-free kernel's PU^T J path in closed loop), goal steps inside and outside the controlled directions."""
    import oracle_lib as ol

    B, ticks = 64, 540
    inp = pkg.workloads.make_inputs(3, B=B, seed=808)
    path = tmp_path / "q.bin"
    inp["q"].astype(np.float64).tofile(path)
    r = subprocess.run([facade_bin, "example08", str(B), str(path), str(ticks)], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    out = np.frombuffer(r.stdout, dtype=np.float64).reshape(ticks, 3, 7, B)
    partial = (np.array([[0, 1.0, 0], [0, 0, 1.0]]), np.array([[1.0, 0, 0]]))
    mcfg = ol.motion_force_task("partial_motion_force_task", frame_pos=(0.07, 0.0, 0.15), partial=partial, internal_otg=False)
    mcfg.use_velocity_saturation = 0
    for i in range(3):
        mcfg.kp_pos[i], mcfg.kv_pos[i], mcfg.ki_pos[i] = 100.0, 20.0, 0.0
        mcfg.kp_ori[i], mcfg.kv_ori[i], mcfg.ki_ori[i] = 100.0, 20.0, 0.0
    jcfg = ol.joint_task("joint_task", internal_otg=True)
    for i in range(7):
        jcfg.kp[i], jcfg.kv[i], jcfg.ki[i] = 100.0, 20.0, 0.0
    o = ol.Oracle(ol.panda_model(), [mcfg, jcfg], B, threads=8)
    o.set_state(inp["q"], np.zeros_like(inp["q"]))
    o.reinitialize()
    st = o.get_mft_status(0)
    gp, R0 = st["pos"].copy(), st["rot"].reshape(3, 3, B).copy()
    gR = R0

    def rotated(axis, angle):
        c, s = np.cos(angle), np.sin(angle)
        A = np.eye(3)
        i, j = (axis + 1) % 3, (axis + 2) % 3
        A[i, i], A[i, j], A[j, i], A[j, j] = c, -s, s, c
        return np.einsum("ik,kjb->ijb", A, R0)

    u = ticks // 9
    worst_regular = 0.0
    for cycle in range(ticks):
        q, dq, tau_g = out[cycle]
        o.set_state(q, dq)
        o.update_task_models()
        if cycle == u:
            gp[0] += 0.1
        if cycle == 2 * u:
            gp[1] += 0.1
            gp[2] += 0.1
        if cycle == 3 * u:
            gR = rotated(2, np.pi / 6)
        if cycle == 4 * u:
            gR = rotated(0, np.pi / 6)
        if cycle == 8 * u:
            qd = q.copy()
            qd[0] += 0.5
            o.set_jt_goals(1, qd)
        o.set_mft_goals(0, gp, np.ascontiguousarray(gR.reshape(9, B)), None, None, None, None)
        tau = o.compute_control_torques(True)
        _, _, ro = o.get_mft_singularity(0)
        e = np.abs(tau_g - tau).max(axis=0) / np.maximum(np.abs(tau).max(axis=0), 1)
        assert e.max() < 1e-5, (cycle, e.max())
        worst_regular = max(worst_regular, e[ro == 3].max(initial=0.0))
    assert worst_regular < 1e-9, worst_regular


@pytest.mark.gpu
def test_cpp_example_05_using_robot_controller(facade_bin, tmp_path):
    """tests/cpp/facade_test.cpp::example05 = examples/05-using_robot_controller.cpp:103-196 call for call: the BASELINE
    hierarchy (MotionForceTask + nullspace JointTask with its default internal OTG) through a RobotController in closed
    loop; updateControllerTaskModels() + computeControlTorques() run the fused tick (the headline SVD-free kernel)."""
    import oracle_lib as ol

    B, ticks = 256, 300
    inp = pkg.workloads.make_inputs(3, B=B, seed=505)
    path = tmp_path / "q.bin"
    inp["q"].astype(np.float64).tofile(path)
    r = subprocess.run([facade_bin, "example05", str(B), str(path), str(ticks)], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    out = np.frombuffer(r.stdout, dtype=np.float64).reshape(ticks, 3, 7, B)
    o = ol.Oracle(ol.panda_model(), [ol.motion_force_task("motion_force_task"), ol.joint_task("joint_task", internal_otg=True)], B, threads=8)
    o.set_state(inp["q"], np.zeros_like(inp["q"]))
    o.reinitialize()
    st = o.get_mft_status(0)
    x0, R0 = st["pos"].copy(), st["rot"].reshape(3, 3, B).copy()
    worst_regular = 0.0
    for cycle in range(ticks):
        q, dq, tau_g = out[cycle]
        t = 0.001 * cycle
        o.set_state(q, dq)
        o.update_task_models()
        w_ori, amp = 2 * np.pi * 0.2, np.pi / 8
        ang = amp * np.sin(w_ori * t)
        c, s = np.cos(ang), np.sin(ang)
        Rt = np.array([[c, 0, -s], [0, 1, 0], [s, 0, c]])
        Rg = np.einsum("ik,kjb->ijb", Rt, R0).reshape(9, B)
        wg, ag = np.zeros((3, B)), np.zeros((3, B))
        wg[1], ag[1] = amp * w_ori * np.cos(w_ori * t), amp * w_ori * w_ori * -np.sin(w_ori * t)
        r_, wc = 0.05, 2 * np.pi * 0.33
        dp = np.array([0.0, np.sin(wc * t), 1 - np.cos(wc * t)])[:, None]
        dv = np.array([0.0, np.cos(wc * t), np.sin(wc * t)])[:, None]
        da = np.array([0.0, -np.sin(wc * t), np.cos(wc * t)])[:, None]
        o.set_mft_goals(0, x0 + r_ * dp, np.ascontiguousarray(Rg), np.broadcast_to(r_ * wc * dv, (3, B)), wg,
                        np.broadcast_to(r_ * wc * wc * da, (3, B)), ag)
        if cycle == ticks // 2:
            g = inp["q"].copy()
            g[0] += 1.5
            o.set_jt_goals(1, g)
        tau = o.compute_control_torques(True)
        _, _, ro = o.get_mft_singularity(0)
        e = np.abs(tau_g - tau).max(axis=0) / np.maximum(np.abs(tau).max(axis=0), 1)
        assert e.max() < 1e-5, (cycle, e.max())
        worst_regular = max(worst_regular, e[ro == 6].max(initial=0.0))
    assert worst_regular < 1e-9, worst_regular


@pytest.mark.gpu
def test_cpp_example_10_orientation_controller(facade_bin, tmp_path):
    """tests/cpp/facade_test.cpp::example10 = examples/10-3d_orientation_controller.cpp:101-166 call for call (on the
    Panda): an orientation-only MotionForceTask with its default internal OTG + JointTask in a RobotController, the
    goal attitude stepped by 60 and 45 degrees."""
    import oracle_lib as ol

    B, ticks = 64, 600
    inp = pkg.workloads.make_inputs(3, B=B, seed=1010)
    path = tmp_path / "q.bin"
    inp["q"].astype(np.float64).tofile(path)
    r = subprocess.run([facade_bin, "example10", str(B), str(path), str(ticks)], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    out = np.frombuffer(r.stdout, dtype=np.float64).reshape(ticks, 3, 7, B)
    partial = (np.zeros((0, 3)), np.eye(3))
    cfg = [ol.motion_force_task("partial_motion_force_task", frame_pos=(0.0, 0.0, 0.0), partial=partial, internal_otg=True),
           ol.joint_task("joint_task", internal_otg=True)]
    o = ol.Oracle(ol.panda_model(), cfg, B, threads=8)
    o.set_state(inp["q"], np.zeros_like(inp["q"]))
    o.reinitialize()
    R0 = o.get_mft_status(0)["rot"].reshape(3, 3, B).copy()
    cx, sx, cy, sy = np.cos(np.pi / 3), np.sin(np.pi / 3), np.cos(np.pi / 4), np.sin(np.pi / 4)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    gR = R0
    worst_regular = 0.0
    for cycle in range(ticks):
        q, dq, tau_g = out[cycle]
        o.set_state(q, dq)
        o.update_task_models()
        if cycle == 0:
            gR = R0
        elif cycle == ticks // 3:
            gR = np.einsum("ik,kjb->ijb", Rx, R0)
        elif cycle == 2 * ticks // 3:
            gR = np.einsum("ik,kjb->ijb", Ry @ Rx, R0)
        o.set_mft_goals(0, None, np.ascontiguousarray(gR.reshape(9, B)), None, None, None, None)
        tau = o.compute_control_torques(True)
        _, _, ro = o.get_mft_singularity(0)
        e = np.abs(tau_g - tau).max(axis=0) / np.maximum(np.abs(tau).max(axis=0), 1)
        assert e.max() < 1e-5, (cycle, e.max())
        worst_regular = max(worst_regular, e[ro == 3].max(initial=0.0))
    assert worst_regular < 1e-9, worst_regular


@pytest.fixture(scope="module")
def eigen_bin(tmp_path_factory):
    """tests/cpp/eigen_adapter_test.cpp against include/Sai2PrimitivesEigen.h; Eigen itself is not in this image, the
    program compiles against tests/cpp/mini_eigen (a test double of the few Eigen operations used)"""
    pkg._abi.load_library()
    out = str(tmp_path_factory.mktemp("cpp") / "eigen_adapter_test")
    subprocess.run(
        ["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include", "sai2_compat"), "-I", os.path.join(ROOT, "tests", "cpp", "mini_eigen"),
         os.path.join(ROOT, "tests", "cpp", "eigen_adapter_test.cpp"), "-o", out, "-L", CSRC, "-lsai2b", f"-Wl,-rpath,{CSRC}",
         "-Wl,-rpath,/opt/rocm/lib"],
        check=True,
    )
    return out


def test_eigen_adapter_compiles(eigen_bin):
    assert os.path.exists(eigen_bin)


@pytest.mark.gpu
@pytest.mark.parametrize("base", [False, True])
def test_eigen_adapter_runs_example_05_with_the_references_own_types(eigen_bin, tmp_path, base):
    """include/Sai2PrimitivesEigen.h: the reference's signatures (shared_ptr<Sai2Model::Sai2Model>, Affine3d, Vector3d,
    Matrix3d, VectorXd) for one robot. tests/cpp/eigen_adapter_test.cpp is example 05 as a reference user wrote it; its
    torques against the oracle, period by period. base: with robot->setTRobotBase(T) (05-...cpp:69) for a base that is
    away from the world's origin — the MotionForceTask's poses and goals are then world quantities."""
    import oracle_lib as ol
    from test_urdf import _urdf_from_model

    ticks = 300
    urdf = tmp_path / "panda_arm.urdf"
    urdf.write_text(_urdf_from_model(pkg.panda_model()))
    q0 = pkg.workloads.make_inputs(3, B=4, seed=55)["q"][:, :1].copy()
    path = tmp_path / "q.bin"
    np.ascontiguousarray(q0[:, 0]).tofile(path)
    r = subprocess.run([eigen_bin, str(urdf), str(path), str(ticks)] + (["base"] if base else []), capture_output=True)
    assert r.returncode == 0, (r.returncode, r.stderr.decode())
    out = np.frombuffer(r.stdout, dtype=np.float64).reshape(ticks, 3, 7, 1)
    m, links = pkg.model_from_urdf(str(urdf))
    if base:
        a = np.array([1.0, 2.0, 3.0]) / np.sqrt(14.0)
        K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        m = pkg.with_base_transform(m, (0.4, -0.2, 0.35), np.eye(3) + np.sin(0.6) * K + (1 - np.cos(0.6)) * K @ K)
    link, fpos, frot = pkg.resolve_link_frame(links, "end-effector", (0.0, 0.0, 0.07))
    o = ol.Oracle(m, [ol.motion_force_task("motion_force_task", link, fpos, frot), ol.joint_task("joint_task", internal_otg=True)], 1)
    o.set_state(q0, np.zeros_like(q0))
    o.reinitialize()
    st = o.get_mft_status(0)
    x0, R0 = st["pos"].copy(), st["rot"].reshape(3, 3, 1).copy()
    for cycle in range(ticks):
        q, dq, tau_g = out[cycle]
        t = 0.001 * cycle
        o.set_state(q, dq)
        o.update_task_models()
        w_ori, amp = 2 * np.pi * 0.2, np.pi / 8
        ang = amp * np.sin(w_ori * t)
        c, s = np.cos(ang), np.sin(ang)
        Rt = np.array([[c, 0, -s], [0, 1, 0], [s, 0, c]])
        Rg = np.einsum("ik,kjb->ijb", Rt, R0).reshape(9, 1)
        wg, ag = np.zeros((3, 1)), np.zeros((3, 1))
        wg[1], ag[1] = amp * w_ori * np.cos(w_ori * t), amp * w_ori * w_ori * -np.sin(w_ori * t)
        r_, wc = 0.05, 2 * np.pi * 0.33
        dp = np.array([0.0, np.sin(wc * t), 1 - np.cos(wc * t)])[:, None]
        dv = np.array([0.0, np.cos(wc * t), np.sin(wc * t)])[:, None]
        da = np.array([0.0, -np.sin(wc * t), np.cos(wc * t)])[:, None]
        o.set_mft_goals(0, x0 + r_ * dp, np.ascontiguousarray(Rg), r_ * wc * dv, wg, r_ * wc * wc * da, ag)
        if cycle == ticks // 2:
            g = q0.copy()
            g[0] += 1.5
            o.set_jt_goals(1, g)
        tau = o.compute_control_torques(True)
        assert _err(tau_g, tau) < 1e-9, (cycle, _err(tau_g, tau))


@pytest.mark.gpu
def test_eigen_adapter_manual_hierarchy(eigen_bin, tmp_path):
    """the TemplateTask-level calls with Eigen types (examples/04-task_and_redundancy.cpp:141-150,188-206): the nullspace
    travels from task to task as a MatrixXd — the adapter's row- / column-major conversions of a non-symmetric matrix"""
    import oracle_lib as ol
    from test_urdf import _urdf_from_model

    ticks = 40
    urdf = tmp_path / "panda_arm.urdf"
    urdf.write_text(_urdf_from_model(pkg.panda_model()))
    q0 = pkg.workloads.make_inputs(3, B=4, seed=56)["q"][:, :1].copy()
    path = tmp_path / "q.bin"
    np.ascontiguousarray(q0[:, 0]).tofile(path)
    r = subprocess.run([eigen_bin, str(urdf), str(path), str(ticks), "manual"], capture_output=True)
    assert r.returncode == 0, (r.returncode, r.stderr.decode())
    out = np.frombuffer(r.stdout, dtype=np.float64).reshape(ticks, 3, 7, 1)
    m, links = pkg.model_from_urdf(str(urdf))
    link, fpos, frot = pkg.resolve_link_frame(links, "end-effector", (0.0, 0.0, 0.07))
    o = ol.Oracle(m, [ol.motion_force_task("motion_force_task", link, fpos, frot), ol.joint_task("joint_task", internal_otg=True)], 1)
    o.set_state(q0, np.zeros_like(q0))
    o.reinitialize()
    x0 = o.get_mft_status(0)["pos"].copy()
    for cycle in range(ticks):
        q, dq, tau_g = out[cycle]
        o.set_state(q, dq)
        o.task_update_model(0, np.eye(7).reshape(49, 1))
        o.task_update_model(1, o.task_nullspaces(0)[2])
        o.set_mft_goals(0, x0 + np.array([[0.0], [0.05], [-0.03]]), None, None, None, None, None)
        tau = o.task_compute_torques(0) + o.task_compute_torques(1)
        assert _err(tau_g, tau) < 1e-9, (cycle, _err(tau_g, tau))
