"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the
same seeded inputs and against the committed golden fixtures.

Tolerance (BASELINE.json north_star): torques within 1e-10 RELATIVE of the oracle, measured per
robot as ||tau - tau_ref||_inf / max(||tau_ref||_inf, 1). Robots inside the singularity-blending
region involve the inverse of a nearly singular matrix (SingularityHandler.cpp:120) and are held to
branch agreement + 1e-6 instead (SURVEY.md §7 "Ill-conditioning")."""
import numpy as np
import pytest

import cases
import oracle_lib as ol
import sai2_primitives_perso_amd as pkg

pytestmark = pytest.mark.gpu
N = pkg.DOF
TOL = 1e-10


def _err(tau, ref):
    scale = np.maximum(np.abs(ref).max(axis=0), 1.0)
    return np.abs(tau - ref).max(axis=0) / scale


def _pair(inp, opts=None, introspection=True):
    go, gg = ol.task_configs(inp["tasks"]), pkg.task_configs(inp["tasks"])
    if opts:
        for c, o in zip(go, opts):
            cases.apply_opts(c, o)
        for c, o in zip(gg, opts):
            cases.apply_opts(c, o)
    o = ol.Oracle(ol.panda_model(), go, inp["B"], threads=8)
    g = pkg.Controller(pkg.panda_model(), gg, inp["B"], introspection=introspection)
    return o, g


@pytest.mark.parametrize("name", list(cases.case_table()))
def test_gpu_matches_oracle_and_golden(name):
    inp, opts, kw, z = cases.load_case(name)
    o, g = _pair(inp, opts)
    tau_o = cases.run_case_on(o, inp, kw, z)
    tau_g = cases.run_case_on(g, inp, kw, z)
    singular = np.zeros(inp["B"], dtype=bool)
    assert cases.rel_err(g.get_model(), o.get_model()) < 1e-12
    for t, (kind, _) in enumerate(inp["tasks"]):
        if kind == "mft":
            so, ao, ro = o.get_mft_singularity(t)
            sg, ag, rg = g.get_mft_singularity(t)
            assert np.array_equal(ro, rg), "branch disagreement"
            assert np.abs(so - sg).max() < 1e-12 and np.abs(ao - ag).max() < 1e-10
            singular |= ro < (o.tasks[t].pos_range + o.tasks[t].ori_range)
            for fo, fg in zip(o.get_mft_task_forces(t), g.get_mft_task_forces(t)):
                assert np.abs(fo - fg).max() < 1e-10 * max(1.0, np.abs(fo).max())
            Mo, Jo, xo, Ro = o.get_model(t)
            Mg, Jg, xg, Rg = g.get_model(t)
            assert np.abs(Jo - Jg).max() < 1e-13 and np.abs(xo - xg).max() < 1e-13 and np.abs(Ro - Rg).max() < 1e-13
    ok = ~singular
    for t in range(len(inp["tasks"])):
        e = _err(g.get_task_torques(t), o.get_task_torques(t))
        assert e[ok].max() < TOL, (t, e[ok].max())
        if singular.any():
            assert e[singular].max() < 1e-6
        assert np.abs(g.get_task_nullspace(t) - o.get_task_nullspace(t))[:, ok].max() < 1e-9
    e = _err(tau_g, tau_o)
    assert e[ok].max() < TOL, e[ok].max()
    if singular.any():
        assert e[singular].max() < 1e-6
    # and against the committed fixture (independent numpy restatement)
    e = _err(tau_g, z["out_tau"])
    assert e[ok].max() < TOL, e[ok].max()


@pytest.mark.parametrize("config,B", [(2, 4096), (3, 4096), (4, 2048)])
def test_gpu_matches_oracle_on_seeded_batches(config, B):
    inp = pkg.workloads.make_inputs(config, B=B, seed=1000 + config)
    o, g = _pair(inp, introspection=True)
    ol.load_inputs(o, inp)
    ol.load_inputs(g, inp)
    tau_o, tau_g = o.tick(), g.tick()
    rank0 = o.tasks[0].pos_range + o.tasks[0].ori_range
    _, _, ro = o.get_mft_singularity(0)
    _, _, rg = g.get_mft_singularity(0)
    assert np.array_equal(ro, rg)
    ok = ro == rank0
    e = _err(tau_g, tau_o)
    assert e[ok].max() < TOL, e[ok].max()
    if (~ok).any():
        assert e[~ok].max() < 1e-6, e[~ok].max()


@pytest.mark.parametrize("config,B", [(2, 4096), (3, 8192)])
def test_fast_path_matches_oracle(config, B):
    """no introspection -> the SVD-free kernel variant (sai2b_fast.hpp); every robot of these
    workloads is certified non-singular, so every wavefront takes it"""
    inp = pkg.workloads.make_inputs(config, B=B, seed=2000 + config)
    o, g = _pair(inp, introspection=False)
    ol.load_inputs(o, inp)
    ol.load_inputs(g, inp)
    e = _err(g.tick(), o.tick())
    assert e.max() < TOL, e.max()
    # and it must agree with the generic (Jacobi-SVD) kernel variant
    g2 = pkg.Controller(pkg.panda_model(), pkg.task_configs(inp["tasks"]), B, introspection=True)
    ol.load_inputs(g2, inp)
    assert _err(g.tick(), g2.tick()).max() < TOL


@pytest.mark.parametrize("name", ["c2_mft", "c3_mft_jt", "c3_full_decoupling", "c3_impedance", "c3_gravity_nocomp",
                                  "c3_velocity_saturation", "c3_integral_3ticks", "c3_force_open_loop",
                                  "c3_force_closed_loop"])
def test_fast_path_matches_golden(name):
    inp, opts, kw, z = cases.load_case(name)
    _, g = _pair(inp, opts, introspection=False)
    tau = cases.run_case_on(g, inp, kw, z)
    assert _err(tau, z["out_tau"]).max() < TOL


def test_fast_path_falls_back_per_robot():
    """a batch with a few singular robots: those (and only those) go through the work list to the
    generic kernel behind the SVD-free one; every robot must match the oracle"""
    B = 1024
    inp = pkg.workloads.make_inputs(3, B=B, seed=41)
    q = inp["q"].copy()
    q[3, 64:70] = -0.0715  # wavefront 1: elbow nearly extended
    q[5, 700:704] = 0.004  # wavefront 10: wrist nearly aligned
    q[5, 710] = 0.3  # wavefront 11: inside the certificate's grey zone or plain regular
    inp["q"] = q
    o, g = _pair(inp, introspection=False)
    ol.load_inputs(o, inp)
    ol.load_inputs(g, inp)
    for _ in range(3):
        tau_o, tau_g = o.tick(), g.tick()
        _, _, ro = o.get_mft_singularity(0)
        assert (ro < 6).sum() >= 2
        e = _err(tau_g, tau_o)
        assert e[ro == 6].max() < TOL
        assert e.max() < 1e-6
        # the work list holds every singular robot and at most the 11 touched ones (the certificate is
        # conservative: a regular robot near the bound may be declined too)
        assert (ro < 6).sum() <= g.fallback_count() <= 11
    # robots leave the singular region: history must be cleared by the generic path, then fast again
    o.set_state(pkg.workloads.make_inputs(3, B=B, seed=41)["q"], inp["dq"])
    g.set_state(pkg.workloads.make_inputs(3, B=B, seed=41)["q"], inp["dq"])
    n_left = int((ro < 6).sum())
    for k in range(2):
        assert _err(g.tick(), o.tick()).max() < TOL
        assert g.fallback_count() == (n_left if k == 0 else 0)  # one generic tick clears the history


def _custom_inputs(tasks, B, seed, singular_fraction=0.0):
    """workload for an arbitrary hierarchy: tasks = [("mft", {"partial": ...}) | ("jt", {"selection": ...})]"""
    rng = np.random.default_rng(seed)
    wl = pkg.workloads
    q = wl.sample_poses(rng, B, singular_fraction=singular_fraction)
    dq = rng.normal(0, 0.3, size=(B, N))
    inp = {"B": B, "tasks": tasks, "q": np.ascontiguousarray(q.T), "dq": np.ascontiguousarray(dq.T)}
    R, p = wl.fk(q)
    _, x, Rf = wl.frame_jacobian(R, p)
    for t, (kind, prm) in enumerate(tasks):
        if kind == "mft":
            ax = rng.normal(size=(B, 3))
            ax /= np.linalg.norm(ax, axis=1, keepdims=True)
            rot = Rf @ wl._expmap(ax * rng.uniform(0, 0.2, size=(B, 1)))
            inp[f"mft{t}"] = {"pos": np.ascontiguousarray((x + rng.uniform(-0.05, 0.05, size=(B, 3))).T),
                              "rot": np.ascontiguousarray(rot.reshape(B, 9).T),
                              "v": np.ascontiguousarray(rng.normal(0, 0.05, size=(B, 3)).T),
                              "w": np.ascontiguousarray(rng.normal(0, 0.05, size=(B, 3)).T),
                              "a": np.ascontiguousarray(rng.normal(0, 0.1, size=(B, 3)).T),
                              "alpha": np.ascontiguousarray(rng.normal(0, 0.1, size=(B, 3)).T)}
        else:
            S = np.eye(N) if prm["selection"] is None else prm["selection"]
            k0 = S.shape[0]
            inp[f"jt{t}"] = {"q": np.ascontiguousarray(((S @ q.T).T + rng.normal(0, 0.1, size=(B, k0))).T),
                             "dq": np.ascontiguousarray(rng.normal(0, 0.1, size=(k0, B))),
                             "ddq": np.ascontiguousarray(rng.normal(0, 0.2, size=(k0, B)))}
    return inp


def _sel(*joints):
    S = np.zeros((len(joints), N))
    for r, j in enumerate(joints):
        S[r, j] = 1
    return S


HIERARCHIES = {
    "c4": [("mft", {"partial": (np.eye(3), np.zeros((0, 3)))}), ("jt", {"selection": _sel(0, 6)}), ("jt", {"selection": None})],
    "partial_mft_mixed": [("mft", {"partial": (np.array([[1.0, 0, 0], [0, 1.0, 1.0]]), np.array([[0, 0, 1.0]]))}),
                          ("jt", {"selection": _sel(1, 3, 5)}), ("jt", {"selection": None})],
    "jt_first": [("jt", {"selection": _sel(0, 1)}), ("mft", {"partial": (np.eye(3), np.zeros((0, 3)))}),
                 ("jt", {"selection": None})],
    "mft_then_overconstrained_jt": [("mft", {"partial": None}), ("jt", {"selection": _sel(2, 4)}), ("jt", {"selection": None})],
    "four_levels": [("mft", {"partial": (np.array([[0, 0, 1.0]]), np.zeros((0, 3)))}),
                    ("mft", {"partial": (np.zeros((0, 3)), np.eye(3))}), ("jt", {"selection": _sel(0)}),
                    ("jt", {"selection": None})],
    "weighted_selection": [("jt", {"selection": np.array([[1.0, 0.5, 0, 0, 0, 0, 0], [0, 0, 0, 1.0, -1.0, 0, 0]])}),
                           ("jt", {"selection": None})],
}


@pytest.mark.parametrize("name", list(HIERARCHIES))
def test_certified_generic_path_matches_oracle(name):
    """no introspection -> the generic kernel skips the Jacobi SVDs for wavefronts whose robots all
    carry the non-singularity / full-row-rank certificates (sai2b_device.hpp: certify_gram, Chain);
    results must equal the oracle's (which always runs the SVDs) and the always-SVD introspection build"""
    B = 1024
    import zlib

    inp = _custom_inputs(HIERARCHIES[name], B, seed=zlib.crc32(name.encode()) % 1000, singular_fraction=0.05)
    o, g = _pair(inp, introspection=False)
    _, g_svd = _pair(inp, introspection=True)
    for c in (o, g, g_svd):
        ol.load_inputs(c, inp)
    for tick in range(2):
        tau_o, tau_g, tau_s = o.tick(), g.tick(), g_svd.tick()
        regular = np.ones(B, dtype=bool)
        for t, (kind, _) in enumerate(inp["tasks"]):
            if kind == "mft":
                _, _, ro = o.get_mft_singularity(t)
                regular &= ro == (o.tasks[t].pos_range + o.tasks[t].ori_range)
        assert regular.sum() > B // 4
        # These hierarchies take unfiltered random poses (no s5/s0 >= 0.1 rejection as in C2/C3/C5), so a
        # few robots have operational-space inertias with condition numbers of 1e5-1e6 (torques of 1e4 Nm)
        # where two different FP64 factorisations legitimately differ by ~1e-10 relative: 1e-9 here.
        for tau in (tau_g, tau_s):
            e = _err(tau, tau_o)
            assert e[regular].max() < 10 * TOL, (name, tick, e[regular].max())
            if (~regular).any():
                assert e[~regular].max() < 1e-6, (name, tick, e[~regular].max())


def _level_conditioning(o, tasks, B):
    """per robot: the worst condition number among the operational-space inertias of the hierarchy's levels — Lambda of a
    MotionForceTask (SingularityHandler.cpp:110-134), M_partial of a JointTask (JointTask.cpp:241-245), taken over their
    range (singular values below 1e-9 of the largest are the directions the level does not have)"""
    kappa = np.ones(B)
    for t, (kind, _) in enumerate(tasks):
        A = o.get_mft_lambda(t)[0] if kind == "mft" else o.get_jt_inertia(t)[0]
        k = int(round(np.sqrt(A.shape[0])))
        sv = np.linalg.svd(A.reshape(k, k, B).transpose(2, 0, 1), compute_uv=False)
        keep = sv > 1e-9 * sv[:, :1]
        smin = np.where(keep, sv, np.inf).min(axis=1)
        kappa = np.maximum(kappa, np.where(sv[:, 0] > 0, sv[:, 0] / smin, 1.0))
    return kappa


@pytest.mark.parametrize("name", ["full_mft_jt", "c4", "mft_then_overconstrained_jt"])
def test_error_against_the_oracle_grows_with_the_conditioning_it_is_blamed_on(name):
    """Outside BASELINE's filtered workloads (s5 / s0 >= 0.1) other tests assert 1e-9 .. 1e-8 instead of 1e-10 and blame
    ill-conditioned operational-space inertias. MEASURED here, 8 192 UNFILTERED random poses per hierarchy: per robot
    kappa = the worst condition number among the inertias the levels invert (_level_conditioning); two backward-stable
    FP64 factorisations differ by ~ eps * kappa. Seen on the MI355X: [full MFT, JT] kappa <= 8e2 (the non-singular
    branch bounds cond(J J^T) by (1 / 0.06)^2 = 278), C4's hierarchy kappa <= 1.7e4, [MFT, JT(2), JT] <= 7.4e2; relative
    torque error of every robot outside a blending region <= 2.7e-13, error / (eps * kappa) <= 20. Asserted: error <= 64 *
    eps * kappa and <= 1e-11 for every such robot. So on random poses of this arm 1e-10 holds with three orders to
    spare in every hierarchy tried; the 1e-9 / 1e-8 of the closed-loop and fuzz tests are headroom for states a
    controller drives itself into (robots parked at the edge of a blending region, random gains, light arms), not
    something these workloads need."""
    B = 8192
    tasks = [("mft", {"partial": None}), ("jt", {"selection": None})] if name == "full_mft_jt" else HIERARCHIES[name]
    inp = _custom_inputs(tasks, B, seed=4242)
    eps = np.finfo(float).eps
    o, g = _pair(inp, introspection=False)
    ol.load_inputs(o, inp)
    ol.load_inputs(g, inp)
    tau_o, tau_g = o.tick(), g.tick()
    regular = np.ones(B, dtype=bool)
    for t, (kind, _) in enumerate(tasks):
        if kind == "mft":
            _, _, ns = o.get_mft_singularity(t)
            regular &= ns == (o.tasks[t].pos_range + o.tasks[t].ori_range)
    kappa = _level_conditioning(o, tasks, B)
    e = _err(tau_g, tau_o)
    ratio = e[regular] / (eps * kappa[regular])
    print(f"{name}: regular {regular.sum()}, kappa median {np.median(kappa[regular]):.1e} 99.9 % {np.quantile(kappa[regular], 0.999):.1e} max "
          f"{kappa[regular].max():.1e}; err max {e[regular].max():.2e}; max err / (eps kappa) {ratio.max():.3f}")
    assert ratio.max() < 64, (name, float(ratio.max()))
    assert e[regular].max() < 1e-11
    assert kappa[regular].max() < 1e5  # what the non-singular branch admits on this arm (see the docstring)


def test_fused_tick_equals_split_api_and_is_repeatable():
    inp = pkg.workloads.make_inputs(3, B=1024, seed=77)
    g = pkg.Controller(pkg.panda_model(), pkg.task_configs(inp["tasks"]), inp["B"])
    ol.load_inputs(g, inp)
    a = g.tick()
    b = g.tick()
    assert np.array_equal(a, b), "stateless config must be bit-repeatable"
    # split API: the torque pass after update_task_models() runs the generic (Jacobi-SVD) kernel
    # variant, the fused tick above the SVD-free one: equal to rounding, not bit for bit
    g.update_task_models()
    c = g.compute_control_torques()
    assert _err(c, a).max() < 1e-12
    g.update_task_models()
    assert np.array_equal(g.compute_control_torques(), c)
    # device-resident output path
    import torch

    out = torch.empty((N, inp["B"]), dtype=torch.float64, device="cuda")
    g.tick(out=out)
    assert np.array_equal(out.cpu().numpy(), a)


def test_multi_tick_singular_history_matches_oracle():
    """robots parked inside the blending region for several ticks: history counters, type decision,
    entering posture and blended torques must follow the oracle tick by tick"""
    B = 256
    inp = pkg.workloads.make_inputs(3, B=B, seed=5)
    rng = np.random.default_rng(8)
    q = inp["q"].copy()
    q[3, : B // 2] = rng.uniform(-0.11, -0.0750, size=B // 2)  # elbow nearly extended
    q[5, B // 2 :] = rng.uniform(0.0, 0.04, size=B - B // 2)  # wrist nearly aligned
    inp["q"] = q
    o, g = _pair(inp, introspection=True)
    ol.load_inputs(o, inp)
    ol.load_inputs(g, inp)
    for tick in range(4):
        dq = inp["dq"] * (1.0 - 0.2 * tick)
        o.set_state(q, dq)
        g.set_state(q, dq)
        tau_o, tau_g = o.tick(), g.tick()
        _, ao, ro = o.get_mft_singularity(0)
        _, ag, rg = g.get_mft_singularity(0)
        assert np.array_equal(ro, rg)
        e = _err(tau_g, tau_o)
        assert (ro < 6).sum() > B // 4, "test should exercise the singular branches"
        assert e[ro == 6].max() < TOL if (ro == 6).any() else True
        assert e.max() < 1e-6, (tick, e.max())


@pytest.mark.parametrize("config,B", [(3, 65536), (2, 4096), (2, 65536), (4, 65536)])
def test_benchmarked_variant_matches_oracle_at_full_batch(config, B):
    """The kernels bench.py times (introspection off: tick_fast_kernel + the pass over its work list for
    C2 / C3, the generic kernel for C4), at BASELINE's batch sizes, EVERY robot against the OpenMP oracle:
    1e-10 for robots in the fully non-singular branch, 1e-6 inside a singularity-blending region, and the
    branch taken (non-singular rank of every MotionForceTask) equal for every robot."""
    inp = pkg.workloads.make_inputs(config, B=B)
    o, g = _pair(inp, introspection=False)
    ol.load_inputs(o, inp)
    ol.load_inputs(g, inp)
    tau_o, tau_g = o.tick(), g.tick()
    assert np.isfinite(tau_g).all()
    rank0 = o.tasks[0].pos_range + o.tasks[0].ori_range
    _, _, ro = o.get_mft_singularity(0)
    ok = ro == rank0
    if config in (2, 3):
        assert ok.all(), "the C2/C3 workloads reject near-singular poses (SURVEY 8(d))"
        assert g.fallback_count() == 0, "every robot of the benchmark workload takes the SVD-free kernel"
    else:
        assert 0.02 * B < (~ok).sum() < 0.5 * B, "C4 injects near-singular poses"
        # the branch decision is observable without introspection through the singularity state the
        # handler keeps: a robot with a singular range has a non-empty classification history
        assert np.array_equal(g.get_singularity_types_count(0) > 0, ~ok)
    e = _err(tau_g, tau_o)
    assert e[ok].max() < TOL, e[ok].max()
    if (~ok).any():
        assert e[~ok].max() < 1e-6, e[~ok].max()
    # second tick on the same state (integrators and singularity history advanced on both sides)
    e = _err(g.tick(), o.tick())
    assert e[ok].max() < TOL and e.max() < 1e-6


def test_size_independent_properties_at_full_batch():
    """65 536 robots (BASELINE config 3): properties that do not need the oracle"""
    B = 65536
    inp = pkg.workloads.make_inputs(3, B=B)
    g = pkg.Controller(pkg.panda_model(), pkg.task_configs(inp["tasks"]), B, introspection=True)
    ol.load_inputs(g, inp)
    tau = g.tick()
    assert np.isfinite(tau).all()
    _, J, _, _ = g.get_model(0)
    J = J.T.reshape(B, 6, N)
    Nm = g.get_task_nullspace(0).T.reshape(B, N, N)
    assert np.abs(J @ Nm).max() < 1e-9  # J N = 0
    assert np.abs(Nm @ Nm - Nm).max() < 1e-9  # N^2 = N
    # dynamic consistency of the nullspace torques: J M^-1 tau_jt = 0
    M = g.get_model().T.reshape(B, N, N)
    tau_jt = g.get_task_torques(1).T
    acc = np.einsum("bij,bj->bi", J, np.linalg.solve(M, tau_jt[..., None])[..., 0])
    assert np.abs(acc).max() < 1e-8
    # linearity in the goals: zero pose/velocity errors and zero feed-forward => zero torque
    g.set_state(inp["q"], np.zeros_like(inp["dq"]))
    g.reinitialize()
    assert np.abs(g.tick()).max() < 1e-9
    # a slice of the big batch equals the same robots run alone (batch independence), vs oracle too
    sl = slice(12345, 12345 + 64)
    sub = pkg.workloads.make_inputs(3, B=B)
    small = {"B": 64, "tasks": sub["tasks"], "q": np.ascontiguousarray(sub["q"][:, sl]), "dq": np.ascontiguousarray(sub["dq"][:, sl])}
    for k in ("mft0", "jt1"):
        small[k] = {n: np.ascontiguousarray(v[:, sl]) for n, v in sub[k].items()}
    o = ol.Oracle(ol.panda_model(), ol.task_configs(small["tasks"]), 64)
    ol.load_inputs(o, small)
    assert _err(tau[:, sl], o.tick()).max() < TOL


def test_facade_mirrors_reference_api():
    B = 512
    inp = pkg.workloads.make_inputs(3, B=B, seed=31)
    robot = pkg.BatchedRobotModel(B)
    robot.setQ(inp["q"])
    robot.setDq(inp["dq"])
    robot.updateModel()
    mft = pkg.MotionForceTask(robot, task_name="ee")
    mft.disableInternalOtg()
    jt = pkg.JointTask(robot)
    assert jt.getInternalOtgEnabled() and not mft.getInternalOtgEnabled()  # on by default (JointTask.h:38)
    jt.disableInternalOtg()
    ctl = pkg.RobotController(robot, [mft, jt])
    assert ctl.getTaskNames() == ["ee", "joint_task"]
    assert ctl.getMotionForceTaskByName("ee") is mft
    with pytest.raises(ValueError, match="not a JointTask"):
        ctl.getJointTaskByName("ee")
    with pytest.raises(ValueError, match="not found"):
        ctl.getJointTaskByName("nope")
    # goals default to the current pose (reInitializeTask): only damping torques
    g0 = inp["mft0"]
    mft.setGoalPosition(g0["pos"])
    mft.setGoalOrientation(g0["rot"])
    mft.setGoalLinearVelocity(g0["v"])
    mft.setGoalAngularVelocity(g0["w"])
    mft.setGoalLinearAcceleration(g0["a"])
    mft.setGoalAngularAcceleration(g0["alpha"])
    jt.setGoalPosition(inp["jt1"]["q"])
    ctl.updateControllerTaskModels()
    tau = ctl.computeControlTorques()
    o = ol.Oracle(ol.panda_model(), ol.task_configs(inp["tasks"]), B, threads=8)
    ol.load_inputs(o, inp)
    assert _err(tau, o.tick()).max() < TOL
    # run-time re-parametrisation reports whether anything changed (MotionForceTask.cpp:830-858) and, if so,
    # the goal of that half is the current pose
    assert mft.parametrizeForceMotionSpaces(1, (0, 0, 1)) is True
    assert mft.parametrizeForceMotionSpaces(1, (0, 0, 3.0)) is False
    assert mft.parametrizeForceMotionSpaces(1, (0, 1, 0)) is True
    assert mft.parametrizeMomentRotMotionSpaces(0) is False
    assert np.abs(mft.getGoalPosition() - mft.getCurrentPosition()).max() < 1e-12
    assert mft.parametrizeForceMotionSpaces(0) is True
    # goal / state getters of the reference (MotionForceTask.h:173-247,369-385,437-440,653-659; JointTask.h:120-151,207-216)
    assert np.array_equal(mft.getGoalLinearVelocity(), np.zeros((3, B)))  # zeroed with the re-parametrised half
    assert np.array_equal(mft.getGoalAngularVelocity(), g0["w"]) and np.array_equal(mft.getGoalAngularAcceleration(), g0["alpha"])
    gf = np.random.default_rng(2).normal(size=(3, B))
    mft.setGoalForce(gf)
    assert np.array_equal(mft.getGoalForce(), gf)  # world-frame parametrisation: as given
    mft2 = pkg.MotionForceTask(robot, task_name="ee_frame", is_force_motion_parametrization_in_compliant_frame=True)
    ctl2 = pkg.RobotController(robot, [mft2])
    mft2.setGoalForce(gf)
    R = mft2.getCurrentOrientation().T.reshape(B, 3, 3)
    assert np.abs(mft2.getGoalForce() - np.einsum("bij,jb->ib", R, gf)).max() < 1e-14  # compliant frame: turned into the world
    sf, sm = np.random.default_rng(3).normal(size=(2, 3, B))
    mft.updateSensedForceAndMoment(sf, sm)
    assert np.array_equal(mft.getSensedForceSensor(), sf) and np.array_equal(mft.getSensedMomentSensor(), sm)
    assert np.array_equal(mft.posSelectionProjector(), np.eye(3)) and mft.getLinearSaturationVelocity() > 0
    assert np.array_equal(jt.getCurrentVelocity(), inp["dq"]) and np.array_equal(jt.getJointSelectionMatrix(), np.eye(7))
    assert len(jt.getGains()) == 1
    otg = jt.getInternalOtg()  # JointTask.h:324: the read-only side of the generator object
    assert otg.isGoalReached().shape == (B,) and not otg.getJerkLimitEnabled()
    assert np.array_equal(otg.getNextPosition(), jt.getDesiredPosition())
    jt.setGains(np.arange(1.0, 8.0), np.ones(7))
    assert len(jt.getGains()) == 7 and jt.getGains()[6][0] == 7.0
    with pytest.raises(ValueError, match="inconsistent with number of task dofs"):
        jt.setGains(np.ones(3), np.ones(3))
    jt.setGainsUnsafe(-np.ones(7), np.ones(7), np.zeros(7))
    assert jt.getGains()[0][0] == -1.0
    del ctl2
    with pytest.raises(ValueError):
        jt.setGoalPosition(np.zeros((3, B)))
    with pytest.raises(ValueError, match="same robot model"):
        pkg.RobotController(pkg.BatchedRobotModel(B), [mft])


@pytest.mark.parametrize("B", [1, 63, 65, 100, 128])
def test_ragged_batch_sizes(B):
    """batches that are not a multiple of the wavefront size (tail lanes exit; the SVD-free kernel takes them
    too, a declined robot going to the generic kernel on its own) and the smallest batch"""
    inp = pkg.workloads.make_inputs(3, B=B, seed=500 + B)
    q = inp["q"].copy()
    q[3, B - 1] = -0.0715  # the very last robot singular: through the work list from a partial wavefront
    inp["q"] = q
    o, g = _pair(inp, introspection=False)
    ol.load_inputs(o, inp)
    ol.load_inputs(g, inp)
    for _ in range(2):
        e = _err(g.tick(), o.tick())
        assert e[: B - 1].max(initial=0.0) < TOL and e[B - 1] < 1e-6
        assert g.fallback_count() >= 1
    assert g.profile_tick(2)[1] > 0  # the SVD-free kernel ran, with the generic one behind it


def test_runtime_reconfiguration_and_error_paths():
    B = 256
    inp = pkg.workloads.make_inputs(3, B=B, seed=61)
    o, g = _pair(inp, introspection=False)
    ol.load_inputs(o, inp)
    ol.load_inputs(g, inp)
    assert _err(g.tick(), o.tick()).max() < TOL
    # setters of the reference (gains, decoupling type) after construction
    for ctrl in (o, g):
        c0, c1 = ctrl.tasks[0], ctrl.tasks[1]
        for i in range(3):
            c0.kp_pos[i], c0.kv_pos[i] = 250.0, 31.0
        c0.dynamic_decoupling_type = pkg.FULL_DYNAMIC_DECOUPLING
        c1.dynamic_decoupling_type = pkg.IMPEDANCE
        for i in range(N):
            c1.kp[i] = 80.0
        ctrl.update_task_config(0, c0)
        ctrl.update_task_config(1, c1)
    assert _err(g.tick(), o.tick()).max() < TOL
    # structural fields must not change; wrong task kinds and shapes are invalid arguments
    bad = pkg.joint_task_config("x", np.eye(N)[:2])
    with pytest.raises(ValueError, match="structural"):
        g.update_task_config(1, bad)
    with pytest.raises(ValueError, match="not a MotionForceTask"):
        g.set_mft_goals(1, pos=np.zeros((3, B)))
    with pytest.raises(ValueError, match="not a JointTask"):
        g.set_jt_goals(0, q=np.zeros((N, B)))
    with pytest.raises(ValueError, match="shape"):
        g.set_state(np.zeros((N, B + 1)), None)
    with pytest.raises(ValueError, match="introspection"):
        g.get_task_nullspace(0)


def test_device_resident_inputs_and_outputs():
    import torch

    B = 512
    inp = pkg.workloads.make_inputs(3, B=B, seed=71)
    o, g = _pair(inp, introspection=False)
    ol.load_inputs(o, inp)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    g.set_state(dev(inp["q"]), dev(inp["dq"]))
    m = inp["mft0"]
    g.set_mft_goals(0, dev(m["pos"]), dev(m["rot"]), dev(m["v"]), dev(m["w"]), dev(m["a"]), dev(m["alpha"]))
    j = inp["jt1"]
    g.set_jt_goals(1, dev(j["q"]), dev(j["dq"]), dev(j["ddq"]))
    out = torch.empty((N, B), dtype=torch.float64, device="cuda")
    g.tick(out=out)
    assert _err(out.cpu().numpy(), o.tick()).max() < TOL
    with pytest.raises(ValueError, match="mixing"):
        g.set_state(dev(inp["q"]), inp["dq"])


def test_long_run_history_wraps_and_integrators_follow_oracle():
    """260 ticks with moving robots, integral gains on, closed-loop... the 200-deep singularity history
    ring wraps, type decisions flip, integrators accumulate: torques must follow the oracle every tick"""
    B = 64
    inp = pkg.workloads.make_inputs(3, B=B, seed=17)
    rng = np.random.default_rng(3)
    q0 = inp["q"].copy()
    q0[3, :24] = rng.uniform(-0.10, -0.072, size=24)  # elbow nearly extended
    q0[5, 24:48] = rng.uniform(0.0, 0.04, size=24)  # wrist nearly aligned
    amp = rng.uniform(0.0, 0.03, size=(N, B))
    phase = rng.uniform(0, 2 * np.pi, size=(N, B))
    opts = [{"ki_pos": 4.0, "ki_ori": 2.0}, {"ki": 3.0}]
    o, g = _pair(inp, opts, introspection=False)
    ol.load_inputs(o, inp)
    ol.load_inputs(g, inp)
    worst, sing_ticks, flips = 0.0, 0, 0
    prev_c1 = None
    for t in range(260):
        w = 2 * np.pi * t / 97.0
        q = q0 + amp * np.sin(w + phase)
        q = np.clip(q, pkg.workloads.PANDA_LOWER[:, None] + 1e-3, pkg.workloads.PANDA_UPPER[:, None] - 1e-3)
        dq = amp * np.cos(w + phase) * (2 * np.pi / 97.0) / 1e-3 * 1e-3
        o.set_state(q, dq)
        g.set_state(q, dq)
        tau_o, tau_g = o.tick(), g.tick()
        _, _, ro = o.get_mft_singularity(0)
        _, c1, c2 = o.get_mft_sh_state(0)
        sing_ticks += int((ro < 6).sum())
        if prev_c1 is not None:
            flips += int(((c1 > c2) != prev_c1).sum())
        prev_c1 = c1 > c2
        e = _err(tau_g, tau_o)
        assert e[ro == 6].max() < 10 * TOL, (t, e[ro == 6].max())
        worst = max(worst, e.max())
        assert e.max() < 1e-5, (t, e.max())
    assert sing_ticks > 2000, "the run should spend many robot-ticks in the singular branches"
    assert (c1 + c2).max() == 200, "history ring must have wrapped (cap 200)"


def test_modified_robot_model_runtime_constants():
    """a 7-DOF arm that is NOT the built-in Panda (other link lengths, masses, a general inertia
    tensor, a tilted joint axis): the kernels read the model from the ctx parameter block instead of the
    compile-time Panda constants; SVD-free and generic kernels against the oracle"""
    B = 512
    inp = pkg.workloads.make_inputs(3, B=B, seed=88)
    mo, mg = ol.panda_model(), pkg.panda_model()
    for m in (mo, mg):
        m.joint_xyz[2][1] = -0.29
        m.joint_xyz[4][1] = 0.41
        m.joint_rpy[3][1] = 0.2
        m.link_mass[1] = 4.2
        m.link_com[5][2] = 0.03
        for k, v in enumerate((0.12, 0.08, 0.1, 0.01, -0.02, 0.015)):
            m.link_inertia[4][k] = v
    for introspection in (False, True):
        o = ol.Oracle(mo, ol.task_configs(inp["tasks"]), B, threads=8)
        g = pkg.Controller(mg, pkg.task_configs(inp["tasks"]), B, introspection=introspection)
        ol.load_inputs(o, inp)
        ol.load_inputs(g, inp)
        tau_o, tau_g = o.tick(), g.tick()
        _, _, ro = o.get_mft_singularity(0)
        e = _err(tau_g, tau_o)
        assert e[ro == 6].max() < 10 * TOL, e[ro == 6].max()
        assert e.max() < 1e-6


def test_reset_integrators_and_config_accessors():
    """resetIntegrators / resetIntegratorsLinear / resetIntegratorsAngular (MotionForceTask.cpp:988-1001)
    and the run-time setters that only touch batch-uniform parameters, against the oracle"""
    B = 64
    inp = pkg.workloads.make_inputs(3, B=B, seed=17)
    opts = [{"ki_pos": 5.0, "ki_ori": 3.0}, {"ki": 2.0}]
    o, g = _pair(inp, opts, introspection=False)
    for c in (o, g):
        ol.load_inputs(c, inp)
    for step in range(9):
        if step == 3:
            for c in (o, g):
                c.reset_integrators(0, 1)  # linear only
        if step == 5:
            for c in (o, g):
                c.reset_integrators(0, 2)
                c.reset_integrators(1, 0)
        if step == 7:  # gains of the singularity handler and force-loop limits change at run time
            for c in (o, g):
                cfg = c.tasks[0]
                cfg.kp_type_1, cfg.kv_type_1, cfg.kv_type_2 = 40.0, 12.0, 4.0
                cfg.kff_force, cfg.max_force_feedback = 0.9, 15.0
                c.update_task_config(0, cfg)
        assert _err(g.tick(), o.tick()).max() < TOL, step


def test_task_observers_between_ticks():
    """getCurrentPosition/Orientation, sensed wrench in the world frame, position/orientation errors and
    the goal...Reached norms (MotionForceTask.h:121-165, MotionForceTask.cpp:540-579), goal getters —
    with a force-space parametrisation so that sigma_position is not the identity"""
    B = 96
    inp = pkg.workloads.make_inputs(3, B=B, seed=23)
    opts = [{"force_space_dimension": 1, "moment_space_dimension": 2, "force_axis": (0, 0, 1), "moment_axis": (1, 0, 0),
             "in_compliant_frame": True}, {}]
    o, g = _pair(inp, opts, introspection=False)
    rng = np.random.default_rng(1)
    sf, sm = rng.normal(0, 5, (3, B)), rng.normal(0, 1, (3, B))
    for c in (o, g):
        ol.load_inputs(c, inp)
        c.set_mft_sensed_wrench(0, sf, sm)
        c.tick()
    so, sg = o.get_mft_status(0), g.get_mft_status(0)
    for k in so:
        assert np.abs(so[k] - sg[k]).max() < 1e-12 * max(1.0, np.abs(so[k]).max()), k
    assert sg["pos_error_norm"].max() > 1e-3  # the goals are away from the current pose
    goals = g.get_mft_goals(0)
    assert np.array_equal(goals[0], inp["mft0"]["pos"]) and np.array_equal(goals[1], inp["mft0"]["rot"])
    assert np.array_equal(g.get_jt_goals(1)[0], inp["jt1"]["q"])
    # facade spelling
    robot = pkg.BatchedRobotModel(B)
    robot.setQ(inp["q"])
    robot.setDq(inp["dq"])
    mft, jt = pkg.MotionForceTask(robot, task_name="ee"), pkg.JointTask(robot)
    ctl = pkg.RobotController(robot, [mft, jt])
    assert mft.goalPositionReached(1e-9).all() and mft.goalOrientationReached(1e-9).all()  # goals = current pose
    mft.setGoalPosition(inp["mft0"]["pos"])
    assert not mft.goalPositionReached(1e-4).any()
    assert np.abs(mft.getCurrentPosition() - sg["pos"]).max() < 1e-12
    assert np.array_equal(jt.getCurrentPosition(), inp["q"]) and np.array_equal(mft.getGoalPosition(), inp["mft0"]["pos"])
    del ctl


def test_new_entry_points_reject_bad_arguments():
    """argument checks of the observers / simulation / integrator entry points (no state is touched)"""
    B = 64
    inp = pkg.workloads.make_inputs(3, B=B, seed=3)
    g = pkg.Controller(pkg.panda_model(), pkg.task_configs(inp["tasks"]), B)
    ol.load_inputs(g, inp)
    tau0 = g.tick()
    with pytest.raises(ValueError, match="dt must be > 0"):
        g.sim_step(None, dt=0.0)
    with pytest.raises(ValueError, match="substeps"):
        g.sim_step(None, dt=0.001, substeps=0)
    with pytest.raises(ValueError, match="not a JointTask"):
        g.get_jt_desired(0)
    with pytest.raises(ValueError, match="not a MotionForceTask"):
        g.get_mft_desired(1)
    with pytest.raises(ValueError, match="not a MotionForceTask"):
        g.get_mft_status(1)
    with pytest.raises(ValueError, match="not a JointTask"):
        g.get_jt_goals(0)
    with pytest.raises(ValueError, match="bad arguments"):
        g.reset_integrators(7)
    with pytest.raises(ValueError, match="bad arguments"):
        g.get_otg_status(-1)
    with pytest.raises(ValueError):
        g.sim_step(np.zeros((7, B + 1)))
    assert np.array_equal(g.tick(), tau0)  # nothing moved


def test_force_space_reparametrisation_side_effects():
    """parametrizeForceMotionSpaces / parametrizeMomentRotMotionSpaces / setClosedLoopForceControl at run
    time (MotionForceTask.cpp:830-890, 973-986) through sai2b_update_task_config: goal := current pose of
    the half that changed, its integrators reset, the other half untouched"""
    B = 128
    inp = pkg.workloads.make_inputs(3, B=B, seed=21)
    opts = [{"ki_pos": 4.0, "ki_ori": 3.0, "force_space_dimension": 1, "force_axis": (0, 0, 1), "closed_loop_force": True}, {}]
    o, g = _pair(inp, opts, introspection=False)
    rng = np.random.default_rng(4)
    sf, sm = rng.normal(0, 3, (3, B)), rng.normal(0, 0.5, (3, B))
    for c in (o, g):
        ol.load_inputs(c, inp)
        c.set_mft_goal_wrench(0, sf, sm)
        c.set_mft_sensed_wrench(0, 0.5 * sf, 0.5 * sm)
    for _ in range(3):
        assert _err(g.tick(), o.tick()).max() < TOL
    pos_goal, rot_goal = (x.copy() for x in g.get_mft_goals(0)[:2])
    st = g.get_mft_status(0)

    def change(**kw):
        for c in (o, g):
            cfg = type(c.tasks[0]).from_buffer_copy(c.tasks[0])
            cases.apply_opts(cfg, kw)
            c.update_task_config(0, cfg)

    # same dimension, same axis up to scale: nothing happens
    change(force_space_dimension=1, force_axis=(0, 0, 2.5))
    assert np.array_equal(g.get_mft_goals(0)[0], pos_goal)
    # new axis: linear half only
    change(force_space_dimension=1, force_axis=(1, 0, 0))
    goals = g.get_mft_goals(0)
    assert np.abs(goals[0] - st["pos"]).max() < 1e-12 and np.array_equal(goals[1], rot_goal)
    assert np.abs(goals[2]).max() == 0  # goal linear velocity
    tau_o, tau_g = o.tick(), g.tick()
    assert _err(tau_g, tau_o).max() < TOL
    # moment space: angular half
    st = g.get_mft_status(0)
    change(moment_space_dimension=2, moment_axis=(0, 1, 0), closed_loop_moment=True)
    goals = g.get_mft_goals(0)
    assert np.abs(goals[1] - st["rot"]).max() < 1e-12
    for _ in range(3):
        assert _err(g.tick(), o.tick()).max() < TOL
    # open loop again: force integrator reset, visible when the loop is closed once more
    change(closed_loop_force=False)
    change(closed_loop_force=True)
    for _ in range(2):
        assert _err(g.tick(), o.tick()).max() < TOL


def test_split_calls_run_the_fused_tick_and_flush_when_observed():
    """sai2b_update_task_models() is deferred and consumed by the torque call behind it (the reference's loop runs the
    same kernels as tick()); anything else in between — here the singularity counters, a state change — makes the
    model update happen first, so what a caller can observe is unchanged."""
    B = 2048
    inp = pkg.workloads.make_inputs(3, B=B, seed=77)
    q = inp["q"].copy()
    q[3, :5] = -0.0715  # a few robots inside the blending region: the once-per-update bookkeeping is visible
    inp["q"] = q
    o, g = _pair(inp, introspection=False)
    ol.load_inputs(o, inp)
    ol.load_inputs(g, inp)
    l0 = g.counters()[0]
    for c in (o, g):
        c.update_task_models()
    assert g.counters()[0] == l0  # nothing launched yet
    tau_o, tau_g = o.compute_control_torques(True), g.compute_control_torques(True)
    assert g.counters()[0] == l0 + 1 and 5 <= g.fallback_count() < 64  # one fused tick: SVD-free kernel + its work list
    _, _, ro = o.get_mft_singularity(0)
    e = _err(tau_g, tau_o)
    assert e[ro == 6].max() < TOL and e.max() < 1e-6
    # observed in between: the update runs when asked about its result, the torque pass does not commit it again
    for c in (o, g):
        c.update_task_models()
    n_g = g.get_singularity_types_count(0)
    assert g.counters()[0] == l0 + 2 and (n_g[:5] > 0).all() and (n_g[5:] == 0).all()
    tau_o, tau_g = o.compute_control_torques(True), g.compute_control_torques(True)
    e = _err(tau_g, tau_o)
    assert e[ro == 6].max() < TOL and e.max() < 1e-6
    # a new state between the two calls: the pending update belongs to the OLD state
    for c in (o, g):
        c.update_task_models()
        c.set_state(inp["q"], 0.5 * inp["dq"])
    tau_o, tau_g = o.compute_control_torques(True), g.compute_control_torques(True)
    e = _err(tau_g, tau_o)
    assert e[ro == 6].max() < 1e-9 and e.max() < 1e-5


def test_robot_base_away_from_the_world_origin():
    """Sai2Model::setTRobotBase (examples/05-using_robot_controller.cpp:69) = sai2b_model_set_base_transform: poses,
    goals and Jacobians of a MotionForceTask are WORLD quantities (MotionForceTask.cpp:100-103, 262). (1) a Panda on a
    shifted and tilted base against the oracle, SVD-free and generic kernels, gravity compensation on (g(q) sees the
    tilt); (2) a property no restatement shares: the law is covariant — the same robot at the world's origin, with
    the goals carried back into its base frame, must produce the same torques (gravity off)."""
    B = 1024 + 5
    inp = pkg.workloads.make_inputs(3, B=B, seed=91)
    a = np.array([2.0, -1.0, 0.5]) / np.sqrt(5.25)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    Rb, pb = np.eye(3) + np.sin(0.8) * K + (1 - np.cos(0.8)) * K @ K, np.array([0.4, -0.2, 0.35])
    moved = pkg.with_base_transform(pkg.panda_model(), pb, Rb)
    world = {k: (dict(v) if isinstance(v, dict) else v) for k, v in inp.items()}
    g0 = world["mft0"]
    g0["pos"] = pb[:, None] + Rb @ inp["mft0"]["pos"]
    g0["rot"] = np.ascontiguousarray(np.einsum("ik,kjb->ijb", Rb, inp["mft0"]["rot"].reshape(3, 3, B)).reshape(9, B))
    for k in ("v", "w", "a", "alpha"):
        g0[k] = Rb @ inp["mft0"][k]
    at_origin = pkg.Controller(pkg.panda_model(), pkg.task_configs(inp["tasks"]), B)
    ol.load_inputs(at_origin, inp)
    tau_origin = at_origin.tick()
    for introspection in (False, True):
        o = ol.Oracle(moved, ol.task_configs(inp["tasks"]), B, threads=8)
        g = pkg.Controller(moved, pkg.task_configs(inp["tasks"]), B, introspection=introspection)
        ol.load_inputs(o, world)
        ol.load_inputs(g, world)
        tau_o, tau_g = o.tick(), g.tick()
        _, _, ro = o.get_mft_singularity(0)
        regular = ro == 6
        assert _err(tau_g, tau_o)[regular].max() < 10 * TOL and _err(tau_g, tau_o).max() < 1e-6
        st = g.get_mft_status(0)
        st0 = at_origin.get_mft_status(0)
        assert np.abs(st["pos"] - (pb[:, None] + Rb @ st0["pos"])).max() < 1e-13
        e = _err(tau_g, tau_origin)
        assert e[regular].max() < 1e-9, e[regular].max()  # covariance
        assert e.max() < 1e-5
        for c in (o, g):
            c.enable_gravity_compensation(True)
        tau_o, tau_g = o.tick(), g.tick()
        assert _err(tau_g, tau_o)[regular].max() < 10 * TOL
        assert np.abs(tau_g - tau_origin).max() > 1.0  # the tilted base's gravity torques are in there
