"""The product's OTG device code (csrc/sai2b_otg_core.hpp) compiled for the HOST, test-only
(tests/cpp/otg_core_test.cpp): planner and wrappers against the reference-generated fixtures, with no
GPU. Built without FMA contraction, like the kernel that uses it (csrc/Makefile: sai2b_otg.o), so
the planner is compared bit for bit."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))

import make_otg_golden as mog  # noqa: E402
import otg_np  # noqa: E402
import otg_scenarios  # noqa: E402
from test_otg_oracle import _Lagged, _OracleCartesian, _OracleJoints  # noqa: E402

GOLDEN = os.path.join(HERE, "golden")


@pytest.fixture(scope="module")
def core(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("otgcore") / "libotg_core_test.so")
    subprocess.run(
        ["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-Wall",
         "-I", os.path.join(ROOT, "sai2-primitives-perso_amd", "csrc"),
         os.path.join(HERE, "cpp", "otg_core_test.cpp"), "-o", out],
        check=True,
    )
    return C.CDLL(out)


def test_core_planner_matches_reference_ruckig_fixture(core):
    """phase-synchronised rows of the fixture (the product always asks for Synchronization::Phase, as the
    wrappers do: OTG_joints.cpp:23)"""
    z = np.load(os.path.join(GOLDEN, "otg_ruckig_calc.npz"))
    rows = mog.random_calc_inputs(len(z["n"]))
    n = 0
    for i, row in enumerate(rows):
        if row[1] != otg_np.SYNC_PHASE:
            continue
        r, T, times, p, v, a = mog.calc_with(core.otg_test_calculate_and_sample, row)
        assert r == z["result"][i] and T == z["duration"][i], i
        assert np.array_equal(p, z["p"][i]) and np.array_equal(v, z["v"][i]) and np.array_equal(a, z["a"][i]), i
        n += 1
    assert n == 300


@pytest.mark.skipif(not otg_np.ref_available(), reason="oracle/_ref/libruckig_ref.so not built")
def test_core_planner_matches_reference_ruckig_live(core):
    ref = otg_np.load_ref()
    for row in mog.random_calc_inputs(4000, seed=321):
        if row[1] != otg_np.SYNC_PHASE:
            continue
        a = mog.calc_with(ref.rref_calculate_and_sample, row)
        b = mog.calc_with(core.otg_test_calculate_and_sample, row)
        assert a[0] == b[0] and a[1] == b[1]
        assert all(np.array_equal(x, y) for x, y in zip(a[2:], b[2:]))


@pytest.mark.parametrize("name", list(otg_scenarios.scenarios().keys()))
def test_core_wrappers_follow_fixture(core, name):
    z = np.load(os.path.join(GOLDEN, "otg_wrappers.npz"))
    rec = otg_scenarios.run(otg_scenarios.scenarios()[name], lambda x0, dt: _Lagged(_OracleJoints(x0, dt, core)),
                            lambda p, R, dt: _Lagged(_OracleCartesian(p, R, dt, core)))
    want = z[name]
    assert rec.shape == want.shape
    assert np.array_equal(rec[:, :3], want[:, :3]), "tick / goal_reached / result flags differ"
    assert np.abs(rec[:, 3:] - want[:, 3:]).max() < 1e-12
