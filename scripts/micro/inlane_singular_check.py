#!/usr/bin/env python3
"""Diagnostic: torque error against the oracle of the robots inside a blending region, through the in-lane singular
branch of tick_cert_kernel<3> (default) and through the generic kernel's work list (SAI2B_NO_INLANE_SINGULAR=1)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np

import cases
import oracle_lib as ol
import sai2_primitives_perso_amd as pkg


def err(tau, ref):
    return np.abs(tau - ref).max(axis=0) / np.maximum(np.abs(ref).max(axis=0), 1.0)


B = 8192
for dec in (0, 1, 2):
    inp = pkg.workloads.make_inputs(4, B=B, seed=4100 + dec)
    opts = [{"decoupling": dec, "ki": 40.0, "ki_pos": 40.0, "ki_ori": 40.0} for _ in inp["tasks"]]
    go, gg = ol.task_configs(inp["tasks"]), pkg.task_configs(inp["tasks"])
    for cfgs in (go, gg):
        for c, o_ in zip(cfgs, opts):
            cases.apply_opts(c, o_)
    o = ol.Oracle(ol.panda_model(), go, B, threads=8)
    g = pkg.Controller(pkg.panda_model(), gg, B)
    os.environ["SAI2B_NO_INLANE_SINGULAR"] = "1"
    h = pkg.Controller(pkg.panda_model(), gg, B)
    del os.environ["SAI2B_NO_INLANE_SINGULAR"]
    for c in (o, g, h):
        ol.load_inputs(c, inp)
    for tick in range(4):
        to, tg, th = o.tick(), g.tick(), h.tick()
        _, _, ro = o.get_mft_singularity(0)
        sing = ro < 3
        eg, eh = err(tg, to), err(th, to)
        so = o.get_mft_sh_state(0)
        sg = g.get_mft_singularity_state(0)
        sh = h.get_mft_singularity_state(0)
        same = all(np.array_equal(a, b) for a, b in zip(sg, sh))
        print(f"dec {dec} tick {tick}: singular {sing.sum()}  in-lane fb {g.fallback_count()} err sing {eg[sing].max():.2e} reg {eg[~sing].max():.2e} | "
              f"group fb {h.fallback_count()} err sing {eh[sing].max():.2e} reg {eh[~sing].max():.2e} | state equal {same}"
              + (f" oracle-equal {all(np.array_equal(a, b) for a, b in zip(sg, so))}" if so is not None else ""))
