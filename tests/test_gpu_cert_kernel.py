"""The SVD-free kernel for general hierarchies (sai2b_cert.hip: one lane per robot, whitened cascade) — what a
tick() runs first for every hierarchy outside [full MFT(, full JT)] — against the CPU oracle and the golden
fixtures: 1e-10 per robot it keeps; the robots it declines (singular, near-singular, rank-deficient levels) must
come out of the generic kernel behind it with the same guarantees as before, and it must KEEP the regular ones."""
import numpy as np
import pytest

import cases
import oracle_lib as ol
import robots
import sai2_primitives_perso_amd as pkg

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _err(tau, ref):
    return np.abs(tau - ref).max(axis=0) / np.maximum(np.abs(ref).max(axis=0), 1.0)


def _pair(inp, opts=None, model=None, omodel=None):
    go, gg = ol.task_configs(inp["tasks"]), pkg.task_configs(inp["tasks"])
    for cfgs in (go, gg):
        for c, o in zip(cfgs, opts or []):
            cases.apply_opts(c, o)
    return (ol.Oracle(omodel or ol.panda_model(), go, inp["B"], threads=8),
            pkg.Controller(model or pkg.panda_model(), gg, inp["B"]))


def _singular(o, inp):
    s = np.zeros(inp["B"], dtype=bool)
    for t, (kind, _) in enumerate(inp["tasks"]):
        if kind == "mft":
            _, _, ro = o.get_mft_singularity(t)
            s |= ro < (o.tasks[t].pos_range + o.tasks[t].ori_range)
    return s


@pytest.mark.parametrize("name", list(cases.case_table()))
def test_cert_kernel_on_golden_cases(name):
    inp, opts, kw, z = cases.load_case(name)
    o, g = _pair(inp, opts)
    tau_o = cases.run_case_on(o, inp, kw, z)
    tau_g = cases.run_case_on(g, inp, kw, z)
    sing = _singular(o, inp)
    e = _err(tau_g, tau_o)
    assert e[~sing].max() < TOL, e[~sing].max()
    if sing.any():
        assert e[sing].max() < 1e-6
    assert _err(tau_g, z["out_tau"])[~sing].max() < TOL


@pytest.mark.parametrize("decoupling", [0, 1, 2])
def test_cert_kernel_c4_hierarchy_keeps_regular_and_singular_robots(decoupling, monkeypatch):
    """[MFT(3), JT(2), JT(7)] (BASELINE config 4: one robot in ten near a singularity), every decoupling type, integral
    gains on so that the state the kernel holds back until a robot is known to finish in it is visible in the torques
    of the next ticks: per-robot parity over three ticks. Round 3: the robots inside a blending region stay in the
    kernel too (cert::singular_part, the SingularityHandler's singular branch in whitened coordinates) — to the same
    1e-10 as the regular ones — and keep the oracle's singularity bookkeeping (types, type-1 / type-2 counters);
    SAI2B_NO_INLANE_SINGULAR=1 sends them through the work list to the generic kernel as before."""
    B = 4096 + 37
    inp = pkg.workloads.make_inputs(4, B=B, seed=4100 + decoupling)
    opts = [{"decoupling": decoupling, "ki": 40.0, "ki_pos": 40.0, "ki_ori": 40.0} for _ in inp["tasks"]]
    o, g = _pair(inp, opts)
    monkeypatch.setenv("SAI2B_NO_INLANE_SINGULAR", "1")
    _, h = _pair(inp, opts)
    monkeypatch.delenv("SAI2B_NO_INLANE_SINGULAR")
    for c in (o, g, h):
        ol.load_inputs(c, inp)
    for tick in range(3):
        tau_o, tau_g, tau_h = o.tick(), g.tick(), h.tick()
        sing = _singular(o, inp)
        assert sing.sum() > B // 30
        e = _err(tau_g, tau_o)
        assert e.max() < TOL, (tick, e[~sing].max(), e[sing].max())
        assert g.fallback_count() <= 2, g.fallback_count()  # (a fully singular task would still go to the work list)
        e = _err(tau_h, tau_o)
        assert e[~sing].max() < TOL and e.max() < 1e-6
        # declined: the singular robots and a margin of near-singular ones (the certificate is sufficient, not necessary)
        assert sing.sum() <= h.fallback_count() <= sing.sum() + B // 20, (h.fallback_count(), sing.sum())
        _, c1o, c2o = o.get_mft_sh_state(0)
        for c in (g, h):
            n, c1, c2 = c.get_mft_singularity_state(0)
            assert np.array_equal(n > 0, sing) and np.array_equal(c1, c1o) and np.array_equal(c2, c2o)


@pytest.mark.parametrize("config", [4, 3])
def test_singular_branch_in_the_kernel_follows_robots_in_and_out_of_the_region(config, monkeypatch):
    """robots carried across the boundary of the blending region and back (the elbow joint swept through its extended pose
    over 40 ticks): entering conditions, the history ring (one bit per tick, 20 deep here), the switch between the type-1
    and type-2 joint strategies and the clearing on the way out all happen in cert::singular_tail / flush_singular —
    torques and bookkeeping must follow the oracle tick by tick. The C4 hierarchy (3-row task, tick_cert_kernel<3>) and
    the headline hierarchy through the 6-row kernel with the branch (SAI2B_FORCE_SING6=1)."""
    B = 256
    inp = pkg.workloads.make_inputs(config, B=B, seed=4400)
    go, gg = ol.task_configs(inp["tasks"]), pkg.task_configs(inp["tasks"])
    for cfgs in (go, gg):
        cfgs[0].sh_buffer_size = 20
    rank = 3 if config == 4 else 6
    o = ol.Oracle(ol.panda_model(), go, B, threads=8)
    if config == 3:
        monkeypatch.setenv("SAI2B_FORCE_SING6", "1")
    g = pkg.Controller(pkg.panda_model(), gg, B)
    monkeypatch.delenv("SAI2B_FORCE_SING6", raising=False)
    for c in (o, g):
        ol.load_inputs(c, inp)
    rng = np.random.default_rng(5)
    q0 = inp["q"].copy()
    amp = rng.uniform(0.1, 0.5, size=B)
    seen_in = seen_out = 0
    for tick in range(40):
        q = q0.copy()
        q[3] = -0.02 - amp * (1 + np.cos(2 * np.pi * tick / 40)) / 2 * 3.0  # elbow: far from extended -> nearly extended -> back
        dq = inp["dq"] * 0.2
        o.set_state(q, dq)
        g.set_state(q, dq)
        tau_o, tau_g = o.tick(), g.tick()
        _, _, ro = o.get_mft_singularity(0)
        seen_in += int((ro < rank).sum())
        seen_out += int((ro == rank).sum())
        one = ro >= rank - 1  # (two singular directions at once: the generic kernel's, 1e-6 as everywhere)
        e = _err(tau_g, tau_o)
        assert e[one].max() < 1e-9 and e.max() < 1e-6, (tick, e[one].max(), e.max())
        assert g.fallback_count() <= (~one).sum() + 2
        _, c1o, c2o = o.get_mft_sh_state(0)
        n, c1, c2 = g.get_mft_singularity_state(0)
        assert np.array_equal(c1, c1o) and np.array_equal(c2, c2o) and np.array_equal(n > 0, ro < rank), tick
    # (the 6-row task is inside a region for most of this sweep; the 3-row task for about half of it)
    assert seen_in > 20 * B // 10 and seen_out > (20 * B // 10 if config == 4 else B), (seen_in, seen_out)


@pytest.mark.parametrize("name", ["c4", "partial_mft_mixed", "jt_first"])
def test_singular_branch_for_projected_and_lower_level_tasks(name):
    """the in-lane singular branch where the MotionForceTask is not a first-level, axis-aligned position task: a partial
    task whose motion space is not spanned by coordinate axes (rows PU^T J: the singular column of U goes back to the six
    task coordinates through PU for the classification) and a MotionForceTask BEHIND a JointTask (Q is not the identity
    when the level begins). Every robot to 1e-9, next to none through the work list."""
    import zlib

    import test_gpu_parity as tp

    B = 2048
    inp = tp._custom_inputs(tp.HIERARCHIES[name], B, seed=zlib.crc32(name.encode()) % 1000, singular_fraction=0.05)
    o, g = _pair(inp)
    for c in (o, g):
        ol.load_inputs(c, inp)
    for tick in range(3):
        tau_o, tau_g = o.tick(), g.tick()
        sing = _singular(o, inp)
        assert sing.sum() >= 20
        assert _err(tau_g, tau_o).max() < 1e-9, (tick, _err(tau_g, tau_o).max())
        assert g.fallback_count() <= 4  # (two singular directions at once still go to the generic kernel)


def test_cert_kernel_backs_off_when_it_declines_most_of_the_batch():
    """example 06's hierarchy on the sliding-base Panda ([partial JT(2), MFT(6), JT(8)]): most random poses are inside
    a blending region, the SVD-free kernel would only add its own time in front of the generic one. The host sees
    the count of declined robots (async read-back every 8th tick) and runs the generic kernel alone for a while;
    results are the same either way. Also a one-robot batch through the same kernels."""
    import test_gpu_robots as tr

    m, kinds, o, g, q, dq = tr._setup("sliding_base", 512, False, False, seed=2)
    seen = []
    for tick in range(14):
        tau_o, tau_g = o.tick(), g.tick()
        sing = np.zeros(512, dtype=bool)
        _, _, ro = o.get_mft_singularity(1)
        sing |= ro < 6
        sing |= g.get_singularity_types_count(1) > 0
        e = _err(tau_g, tau_o)
        assert e[~sing].max() < 1e-9 and e.max() < 1e-5, (tick, e.max())
        seen.append(g.fallback_count())
    assert seen[0] > 512 * 0.4 and seen[0] < 512, seen  # the kernel ran and declined most robots ...
    assert seen[-1] == 512, seen  # ... and is being skipped by now
    m, kinds, o, g, q, dq = tr._setup("six_r", 1, False, False, seed=5)
    for tick in range(3):
        assert _err(g.tick(), o.tick()).max() < 1e-9


def test_six_row_kernel_with_the_singular_branch_takes_over_when_most_of_a_big_batch_is_singular(monkeypatch):
    """[MFT(6), JT(7)] on 49 152 UNFILTERED random poses: more than half are inside a blending region of the 6-row task (the
    BASELINE workloads reject such poses). The headline kernel declines them; once the host has seen the count (an async
    read-back every 8th tick) it runs tick_cert_kernel<6, S6> instead — the singular branch of a 4- to 6-row task in the
    lane (cert::singular_streamed: eigenvalues of the 6 x 6 Gram matrix, the smallest singular triplet by inverse
    iteration, a Householder reflector that puts the singular direction last) — and only robots with two singular
    directions still go through the work list. Same torques (1e-9 against the oracle) and singularity bookkeeping either
    way; SAI2B_NO_SING6=1 keeps the first route."""
    import test_gpu_parity as tp

    B = 49152  # (the host switches above 20 480 declined robots)
    inp = tp._custom_inputs([("mft", {"partial": None}), ("jt", {"selection": None})], B, seed=711, singular_fraction=0.1)
    o, g = _pair(inp)
    monkeypatch.setenv("SAI2B_NO_SING6", "1")
    _, h = _pair(inp)
    monkeypatch.delenv("SAI2B_NO_SING6")
    for c in (o, g, h):
        ol.load_inputs(c, inp)
    seen, seen_h = [], []
    for tick in range(12):
        tau_o, tau_g, tau_h = o.tick(), g.tick(), h.tick()
        g.synchronize()  # (so that the read-back of tick 0 and 8 has certainly arrived by the next tick)
        _, _, ro = o.get_mft_singularity(0)
        for tau in (tau_g, tau_h):
            e = _err(tau, tau_o)
            assert e[ro >= 5].max() < 1e-9 and e.max() < 1e-6, (tick, e[ro >= 5].max(), e.max())
        seen.append(g.fallback_count())
        seen_h.append(h.fallback_count())
        _, c1o, c2o = o.get_mft_sh_state(0)
        for c in (g, h):
            n, c1, c2 = c.get_mft_singularity_state(0)
            assert np.array_equal(n, 6 - ro) and np.array_equal(c1, c1o) and np.array_equal(c2, c2o), tick
    n_sing, n_two = int((ro < 6).sum()), int((ro < 5).sum())
    assert n_sing > B // 3
    assert seen[0] >= n_sing and seen_h[-1] >= n_sing  # the headline kernel declines every robot inside a region ...
    assert n_two <= seen[-1] <= n_two + B // 100, (seen, n_two)  # ... the 6-row kernel with the branch only those with two directions


def _rows(*idx):
    out = np.zeros((len(idx), 3))
    for r, i in enumerate(idx):
        out[r, i] = 1
    return out


@pytest.mark.parametrize("name", ["five_rows", "four_rows_rotated", "six_rows_behind_nothing", "five_rows_behind_a_joint_task"])
def test_singular_branch_of_four_to_six_row_tasks(name, monkeypatch):
    """cert::singular_streamed on every task size it is instantiated for (4, 5 and 6 rows), axis-aligned and rotated motion
    spaces, first and lower levels: SAI2B_FORCE_SING6=1 runs tick_cert_kernel<6, S6> from the first tick (a batch this small
    would never make the host choose it). Robots with at most one singular direction stay and match the oracle to 1e-9."""
    c, s_ = np.cos(0.4), np.sin(0.4)
    tasks = {
        "five_rows": [("mft", {"partial": (np.eye(3), _rows(0, 1))}), ("jt", {"selection": None})],
        "four_rows_rotated": [("mft", {"partial": (np.array([[c, s_, 0], [-s_, c, 0]]), np.array([[0, c, s_], [0, -s_, c]]))}),
                              ("jt", {"selection": None})],
        "six_rows_behind_nothing": [("mft", {"partial": None}), ("jt", {"selection": None})],
        "five_rows_behind_a_joint_task": [("jt", {"selection": np.eye(7)[6:7]}), ("mft", {"partial": (np.eye(3), _rows(0, 2))}),
                                          ("jt", {"selection": None})],
    }[name]
    import test_gpu_parity as tp

    B = 1024 + 5
    inp = tp._custom_inputs(tasks, B, seed=len(name), singular_fraction=0.1)
    monkeypatch.setenv("SAI2B_FORCE_SING6", "1")
    o, g = _pair(inp)
    monkeypatch.delenv("SAI2B_FORCE_SING6")
    for c_ in (o, g):
        ol.load_inputs(c_, inp)
    t_m = [k for k, (kind, _) in enumerate(tasks) if kind == "mft"][0]
    rank = o.tasks[t_m].pos_range + o.tasks[t_m].ori_range
    for tick in range(3):
        tau_o, tau_g = o.tick(), g.tick()
        _, _, ro = o.get_mft_singularity(t_m)
        one = ro >= rank - 1
        assert (ro == rank - 1).sum() > B // 50, (ro == rank - 1).sum()
        e = _err(tau_g, tau_o)
        assert e[one].max() < 1e-9 and e.max() < 1e-6, (tick, e[one].max(), e.max())
        assert (~one).sum() <= g.fallback_count() <= (~one).sum() + B // 50, (g.fallback_count(), (~one).sum())
        _, c1o, c2o = o.get_mft_sh_state(t_m)
        n, c1, c2 = g.get_mft_singularity_state(t_m)
        assert np.array_equal(n, rank - ro) and np.array_equal(c1, c1o) and np.array_equal(c2, c2o), tick


@pytest.mark.parametrize("config", [2, 3])
def test_six_row_instantiation_on_the_full_motion_force_task(config, monkeypatch):
    """SAI2B_PREFER_CERT=1 (read when a controller is created) sends [MFT(6)] and [MFT(6), JT(7)] — normally
    sai2b_fast.hpp's — through tick_cert_kernel<6>: a second, independently written SVD-free path for the headline
    workload. Same 1e-10 against the oracle, and the two GPU paths agree with each other."""
    B = 2048 + 11
    inp = pkg.workloads.make_inputs(config, B=B, seed=4200 + config)
    o, g_fast = _pair(inp)
    monkeypatch.setenv("SAI2B_PREFER_CERT", "1")
    g_cert = pkg.Controller(pkg.panda_model(), pkg.task_configs(inp["tasks"]), B)
    monkeypatch.delenv("SAI2B_PREFER_CERT")
    for c in (o, g_fast, g_cert):
        ol.load_inputs(c, inp)
    for tick in range(2):
        tau_o, tau_f, tau_c = o.tick(), g_fast.tick(), g_cert.tick()
        assert _err(tau_c, tau_o).max() < TOL
        assert _err(tau_c, tau_f).max() < TOL
        assert g_cert.fallback_count() == 0
