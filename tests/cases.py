"""Shared description of the golden-fixture cases (tests/golden/make_golden.py:cases) and how their
options map onto sai2b_task_config fields."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden  # noqa: E402

import sai2_primitives_perso_amd as pkg  # noqa: E402

GOLDEN_DIR = os.path.join(HERE, "golden")


def case_table():
    return {name: (config, B, opts, kw) for name, config, B, opts, kw in make_golden.cases()}


def load_case(name):
    config, B, opts, kw = case_table()[name]
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    inp = pkg.workloads.make_inputs(config, B=B)
    if kw.get("prepare") == "singular":
        inp = make_golden.make_singular(inp)
    # the fixture carries its own inputs: check the regenerated workload is the committed one
    assert np.array_equal(inp["q"], z["q"]) and np.array_equal(inp["dq"], z["dq"]), "workload drifted from fixture"
    for t, (kind, _) in enumerate(inp["tasks"]):
        for k in inp[f"{kind}{t}"]:
            inp[f"{kind}{t}"][k] = z[f"in_{kind}{t}_{k}"]
    return inp, opts, kw, z


def apply_opts(cfg, opts):
    """mutate a TaskConfig according to a make_golden option dict"""
    if not opts:
        return cfg
    for k, v in opts.items():
        if k == "decoupling":
            cfg.dynamic_decoupling_type = v
        elif k == "bie_threshold":
            cfg.bie_threshold = v
        elif k == "velocity_saturation":
            cfg.use_velocity_saturation = 1
            if cfg.type == pkg.MOTION_FORCE_TASK:
                cfg.linear_saturation_velocity, cfg.angular_saturation_velocity = v
            else:
                for i in range(len(cfg.saturation_velocity)):  # every slot: robots of up to SAI2B_MAX_DOF joints
                    cfg.saturation_velocity[i] = v
        elif k in ("kp_pos", "kv_pos", "ki_pos", "kp_ori", "kv_ori", "ki_ori"):
            for i in range(3):
                getattr(cfg, k)[i] = v
        elif k in ("kp", "kv", "ki"):
            for i in range(len(getattr(cfg, k))):
                getattr(cfg, k)[i] = v
        elif k in ("force_space_dimension", "moment_space_dimension"):
            setattr(cfg, k, v)
        elif k in ("force_axis", "moment_axis"):
            a = np.asarray(v, dtype=float)
            a = a / np.linalg.norm(a)
            for i in range(3):
                getattr(cfg, k)[i] = a[i]
        elif k in ("closed_loop_force", "closed_loop_moment"):
            setattr(cfg, k, int(v))
        elif k == "in_compliant_frame":
            cfg.parametrization_in_compliant_frame = int(v)
        elif k == "passivity":
            cfg.passivity_enabled = int(v)
        elif k == "enforce_type_1":
            cfg.enforce_type_1_strategy = int(v)
        elif k == "enforce_handling":
            cfg.enforce_handling_strategy = int(v)
        elif k == "sv_sign":
            cfg.singular_vector_sign = int(v)
        else:
            raise KeyError(k)
    return cfg


def run_case_on(ctrl, inp, kw, z):
    """drive an Oracle / Controller through a fixture case; returns tau"""
    from oracle_lib import load_inputs

    load_inputs(ctrl, inp)
    if "in_wrench_f" in z:
        B = inp["B"]
        ctrl.set_mft_goal_wrench(0, z["in_wrench_f"][:, :B], z["in_wrench_m"][:, :B])
        ctrl.set_mft_sensed_wrench(0, z["in_wrench_sf"][:, :B], z["in_wrench_sm"][:, :B])
    if kw.get("gravity_comp"):
        ctrl.enable_gravity_compensation(True)
    tau = None
    for tick in range(kw.get("ticks", 1)):
        if kw.get("tick_inputs") == "popc":
            dq_t, sf_t, sm_t = make_golden.popc_tick_inputs(tick, inp["B"], z["in_wrench_f"])
            ctrl.set_state(inp["q"], dq_t)
            ctrl.set_mft_sensed_wrench(0, sf_t, sm_t)
        ctrl.update_task_models()
        tau = ctrl.compute_control_torques(with_compensation=kw.get("with_comp", True))
    return tau


def rel_err(a, b):
    """max over robots of ||a-b||_inf / max(||b||_inf, 1e-300) per robot (arrays [C][B])"""
    a, b = np.atleast_2d(a), np.atleast_2d(b)
    den = np.maximum(np.abs(b).max(axis=0), 1e-12)
    return float((np.abs(a - b).max(axis=0) / den).max())
