"""-m gpu: the singular-vector sign settings (enum sai2b_singular_vector_sign, include/sai2b.h) on the device. The
reference's classifySingularity perturbs along V_s as Eigen's JacobiSVD left it (reference
src/tasks/SingularityHandler.cpp:78-81,253-265); neither side can reproduce that sign, so the convention is a
setting and the GPU must agree with the oracle under EVERY setting, in every kernel family: the one-lane generic
kernel (introspection), the lanes-per-robot generic kernel behind the SVD-free kernels' work lists (C3: sai2b_fast,
C4: sai2b_cert) and on its own. tests/test_svd_sign.py holds the CPU half and the measured sign dependence."""
import numpy as np
import pytest

import cases
import oracle_lib as ol
import sai2_primitives_perso_amd as pkg
from test_svd_sign import SIGNS, sign_report

pytestmark = pytest.mark.gpu
make_golden = cases.make_golden


def _both(inp, sign, introspection, ticks=1):
    go, gg = ol.task_configs(inp["tasks"]), pkg.task_configs(inp["tasks"])
    go[0].singular_vector_sign = gg[0].singular_vector_sign = sign
    o = ol.Oracle(ol.panda_model(), go, inp["B"], threads=8)
    g = pkg.Controller(pkg.panda_model(), gg, inp["B"], introspection=introspection)
    for c in (o, g):
        ol.load_inputs(c, inp)
    for _ in range(ticks):
        tau_o, tau_g = o.tick(), g.tick()
    _, c1o, c2o = o.get_mft_sh_state(0)
    ng, c1g, c2g = g.get_mft_singularity_state(0)
    _, _, ns = o.get_mft_singularity(0)
    return (c1o.astype(int), c2o.astype(int), ns), (c1g, c2g, ng), tau_o, tau_g


@pytest.mark.parametrize("sign", SIGNS)
@pytest.mark.parametrize("introspection", [True, False])
def test_singular_fixture_classified_like_the_oracle_under_every_sign(sign, introspection):
    inp = make_golden.make_singular(pkg.workloads.make_inputs(3, B=48))
    (c1o, c2o, _), (c1g, c2g, _), tau_o, tau_g = _both(inp, sign, introspection, ticks=3)
    assert np.array_equal(c1o, c1g) and np.array_equal(c2o, c2g)
    scale = np.maximum(np.abs(tau_o).max(axis=0), 1.0)
    assert (np.abs(tau_g - tau_o).max(axis=0) / scale).max() < 1e-6  # every robot is inside a blending region


@pytest.mark.parametrize("sign", SIGNS)
def test_c4_workload_singular_robots_classified_like_the_oracle_under_every_sign(sign):
    """the ~3 600 robots of the C4 bench workload inside a blending region, through the kernels the bench runs (the
    SVD-free kernel for general hierarchies declines them, the lanes-per-robot generic kernel classifies them)"""
    inp = pkg.workloads.make_inputs(4, B=65536)
    (c1o, c2o, ns), (c1g, c2g, ng), tau_o, tau_g = _both(inp, sign, introspection=False)
    rank = 3
    sing = ns < rank
    assert sing.sum() > 3000
    assert np.array_equal(c1o, c1g) and np.array_equal(c2o, c2g)
    assert np.array_equal(ng > 0, sing)
    scale = np.maximum(np.abs(tau_o).max(axis=0), 1.0)
    e = np.abs(tau_g - tau_o).max(axis=0) / scale
    assert e[~sing].max() < 1e-10 and e[sing].max() < 1e-6


def test_sign_dependence_measured_on_the_device_equals_the_oracles():
    """the count DESIGN.md §2 quotes (robots whose type flips with the sign), taken from the GPU's own classification"""
    inp = pkg.workloads.make_inputs(4, B=65536)
    ref = sign_report(inp)
    c1 = {}
    for sign in SIGNS:
        _, (c1g, _, ng), _, _ = _both(inp, sign, introspection=False)
        c1[sign] = c1g
    assert int((c1[pkg.SV_SIGN_V_MAX_POSITIVE] != c1[pkg.SV_SIGN_V_MAX_NEGATIVE]).sum()) == ref["type_depends_on_sign"]
    assert int(c1[pkg.SV_SIGN_EITHER].sum()) == ref["type1_either"] and int(c1[pkg.SV_SIGN_BOTH].sum()) == ref["type1_both"]


def test_bad_sign_setting_is_refused():
    cfg = pkg.task_configs([("mft", {}), ("jt", {})])
    cfg[0].singular_vector_sign = 7
    with pytest.raises(ValueError, match="singular_vector_sign"):
        pkg.Controller(pkg.panda_model(), cfg, 64)
