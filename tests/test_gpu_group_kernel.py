"""The generic kernel with a robot spread over 16 or 8 lanes (sai2b_group.hip), forced for every hierarchy
(no SVD-free path, no introspection), against the golden fixtures and the CPU oracle: 1e-10 per robot in the
fully non-singular branch, 1e-6 inside a singularity-blending region, branch bookkeeping equal."""
import numpy as np
import pytest

import cases
import oracle_lib as ol
import sai2_primitives_perso_amd as pkg

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _err(tau, ref):
    return np.abs(tau - ref).max(axis=0) / np.maximum(np.abs(ref).max(axis=0), 1.0)


@pytest.fixture(params=[16, 8])
def lanes(request, monkeypatch):
    monkeypatch.setenv("SAI2B_GENERIC_LANES", str(request.param))
    monkeypatch.setenv("SAI2B_NO_FAST_PATH", "1")
    return request.param


def _pair(inp, opts=None):
    go, gg = ol.task_configs(inp["tasks"]), pkg.task_configs(inp["tasks"])
    for cfgs in (go, gg):
        for c, o in zip(cfgs, opts or []):
            cases.apply_opts(c, o)
    return ol.Oracle(ol.panda_model(), go, inp["B"], threads=8), pkg.Controller(pkg.panda_model(), gg, inp["B"])


@pytest.mark.parametrize("name", list(cases.case_table()))
def test_group_kernel_matches_oracle_and_golden(name, lanes):
    inp, opts, kw, z = cases.load_case(name)
    o, g = _pair(inp, opts)
    tau_o = cases.run_case_on(o, inp, kw, z)
    tau_g = cases.run_case_on(g, inp, kw, z)
    singular = np.zeros(inp["B"], dtype=bool)
    for t, (kind, _) in enumerate(inp["tasks"]):
        if kind == "mft":
            _, _, ro = o.get_mft_singularity(t)
            singular |= ro < (o.tasks[t].pos_range + o.tasks[t].ori_range)
            assert np.array_equal(g.get_singularity_types_count(t) > 0, ro < (o.tasks[t].pos_range + o.tasks[t].ori_range))
    ok = ~singular
    e = _err(tau_g, tau_o)
    assert e[ok].max() < TOL, e[ok].max()
    if singular.any():
        assert e[singular].max() < 1e-6, e[singular].max()
    e = _err(tau_g, z["out_tau"])
    assert e[ok].max() < TOL, e[ok].max()


@pytest.mark.parametrize("config,B", [(2, 1000), (3, 4096), (4, 4096)])
def test_group_kernel_on_seeded_batches(config, B, lanes):
    """ragged batch sizes too (B not a multiple of the robots per wavefront), three ticks: state carried"""
    inp = pkg.workloads.make_inputs(config, B=B, seed=3000 + config)
    o, g = _pair(inp)
    ol.load_inputs(o, inp)
    ol.load_inputs(g, inp)
    rank0 = o.tasks[0].pos_range + o.tasks[0].ori_range
    for tick in range(3):
        tau_o, tau_g = o.tick(), g.tick()
        _, _, ro = o.get_mft_singularity(0)
        ok = ro == rank0
        assert np.array_equal(g.get_singularity_types_count(0) > 0, ~ok)
        e = _err(tau_g, tau_o)
        assert e[ok].max() < TOL, (tick, e[ok].max())
        if (~ok).any():
            assert e[~ok].max() < 1e-6, (tick, e[~ok].max())


def test_group_kernel_split_api_and_no_compensation(lanes):
    B = 512
    inp = pkg.workloads.make_inputs(4, B=B, seed=11)
    o, g = _pair(inp)
    ol.load_inputs(o, inp)
    ol.load_inputs(g, inp)
    rank0 = o.tasks[0].pos_range + o.tasks[0].ori_range
    for comp in (True, False):
        for c in (o, g):
            c.update_task_models()
        tau_o, tau_g = o.compute_control_torques(comp), g.compute_control_torques(comp)
        _, _, ro = o.get_mft_singularity(0)
        ok = ro == rank0
        e = _err(tau_g, tau_o)
        assert e[ok].max() < TOL and e.max() < 1e-6
