"""Randomised configurations, GPU (through the C ABI) against the oracle: for each seed a hierarchy, task
options (decoupling, gains, integral terms, velocity saturation, force / moment spaces, open or closed
loop, compliant-frame parametrisation, singularity-handling switches), gravity compensation, the internal
OTG and the kernel variant are drawn at random; controller and simulation then run in closed loop for a
few control periods on both sides (state, integrators, singularity history and generator state all
evolving) and torques and joint states are compared every period.

The option vocabulary is the one of the golden cases (tests/cases.py: apply_opts); what the golden cases
pin one option at a time, this pins in combination."""
import os
import zlib

import numpy as np
import pytest

import cases
import oracle_lib as ol
import sai2_primitives_perso_amd as pkg
from test_gpu_parity import HIERARCHIES, _custom_inputs

pytestmark = pytest.mark.gpu

N = pkg.DOF
FULL, BIE, IMPEDANCE = pkg.FULL_DYNAMIC_DECOUPLING, pkg.BOUNDED_INERTIA_ESTIMATES, pkg.IMPEDANCE
SHAPES = dict(HIERARCHIES)
SHAPES["c2"] = [("mft", {"partial": None})]
SHAPES["c3"] = [("mft", {"partial": None}), ("jt", {"selection": None})]
SHAPES["c1"] = [("jt", {"selection": None})]


def _draw_opts(rng, tasks):
    decoupling = [FULL, BIE, IMPEDANCE][rng.integers(3)]  # one type per controller keeps one shared M_BIE
    opts = []
    for kind, prm in tasks:
        o = {"decoupling": decoupling}
        if kind == "mft":
            o.update(kp_pos=float(rng.uniform(50, 400)), kv_pos=float(rng.uniform(10, 40)),
                     kp_ori=float(rng.uniform(50, 400)), kv_ori=float(rng.uniform(10, 40)))
            if rng.random() < 0.4:
                o.update(ki_pos=float(rng.uniform(1, 20)), ki_ori=float(rng.uniform(1, 20)))
            if rng.random() < 0.4:
                o["velocity_saturation"] = (float(rng.uniform(0.05, 0.4)), float(rng.uniform(0.3, 1.5)))
            if prm["partial"] is None and rng.random() < 0.5:  # force / moment spaces on full tasks
                o["force_space_dimension"] = int(rng.integers(0, 4))
                o["moment_space_dimension"] = int(rng.integers(0, 4))
                o["force_axis"] = tuple(rng.normal(size=3))
                o["moment_axis"] = tuple(rng.normal(size=3))
                o["closed_loop_force"] = bool(rng.integers(2))
                o["closed_loop_moment"] = bool(rng.integers(2))
                o["in_compliant_frame"] = bool(rng.integers(2))
            if rng.random() < 0.25:
                o["enforce_type_1"] = True
            if rng.random() < 0.15:
                o["enforce_handling"] = False
        else:
            o.update(kp=float(rng.uniform(20, 200)), kv=float(rng.uniform(5, 30)))
            if rng.random() < 0.4:
                o["ki"] = float(rng.uniform(1, 10))
            if rng.random() < 0.4:
                o["velocity_saturation"] = float(rng.uniform(0.3, 1.5))
        opts.append(o)
    return opts


def _configs(make, tasks, opts, otg):
    cfgs = make(tasks)
    for c, o in zip(cfgs, opts):
        cases.apply_opts(c, o)
        c.use_internal_otg = int(otg)
    return cfgs


# SAI2B_FUZZ_SEEDS=<n> widens the sweep for an exploratory run (600 seeds were run clean when this was written)
@pytest.mark.parametrize("seed", range(int(os.environ.get("SAI2B_FUZZ_SEEDS", "24"))))
def test_random_configuration_closed_loop(seed):
    rng = np.random.default_rng(9000 + seed)
    name = sorted(SHAPES)[seed % len(SHAPES)]
    tasks = SHAPES[name]
    B = 192
    inp = _custom_inputs(tasks, B, seed=zlib.crc32(name.encode()) % 1000 + seed, singular_fraction=0.1)
    opts = _draw_opts(rng, tasks)
    otg = bool(rng.integers(2))
    gravity = bool(rng.integers(2))
    introspection = bool(rng.integers(2))
    o = ol.Oracle(ol.panda_model(), _configs(ol.task_configs, tasks, opts, otg), B, threads=8)
    g = pkg.Controller(pkg.panda_model(), _configs(pkg.task_configs, tasks, opts, otg), B, introspection=introspection)
    wrench = {k: rng.normal(0, s, size=(3, B)) for k, s in (("f", 3.0), ("m", 0.5), ("sf", 3.0), ("sm", 0.5))}
    for c in (o, g):
        ol.load_inputs(c, inp)
        c.enable_gravity_compensation(gravity)
        for t, (kind, _) in enumerate(tasks):
            if kind == "mft" and "force_space_dimension" in opts[t]:
                c.set_mft_goal_wrench(t, wrench["f"], wrench["m"])
                c.set_mft_sensed_wrench(t, wrench["sf"], wrench["sm"])
    what = (seed, name, opts, otg, gravity, introspection)
    worst = 0.0
    for period in range(6):
        tau_o, tau_g = o.tick(), g.tick()
        regular = np.ones(B, dtype=bool)
        for t, (kind, _) in enumerate(tasks):
            if kind == "mft":
                _, _, ro = o.get_mft_singularity(t)
                regular &= ro == (o.tasks[t].pos_range + o.tasks[t].ori_range)
        den = np.maximum(np.abs(tau_o).max(axis=0), 1e-9)
        e = np.abs(tau_g - tau_o).max(axis=0) / den
        # regular robots: the parity bar with the margin test_certified_generic_path_matches_oracle explains
        # (unfiltered random poses); robots inside a singularity-blending region: Jacobi-SVD vectors of
        # near-degenerate subspaces differ at 1e-7 between two correct FP64 implementations
        assert e[regular].max() < 1e-8, (what, period, float(e[regular].max()))
        if (~regular).any():
            assert e[~regular].max() < 1e-5, (what, period, float(e[~regular].max()))
        worst = max(worst, float(e[regular].max()))
        o.sim_step(tau_o, 0.001, 1, with_gravity=gravity)
        g.sim_step(tau_g, 0.001, 1, with_gravity=gravity)
    qo, dqo = o.get_state()
    qg, dqg = g.get_state()
    assert np.abs(qg - qo).max() < 1e-9 and np.abs(dqg - dqo).max() < 1e-6, what
