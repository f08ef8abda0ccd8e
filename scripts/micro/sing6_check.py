#!/usr/bin/env python3
"""Diagnostic for an in-lane singular branch of 6-row tasks (tick_cert_kernel<6>, SAI2B_PREFER_CERT=1 on the C3 hierarchy with
one pose in ten near a singularity): error against the oracle, robots through the work list, time per tick."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np

import oracle_lib as ol
import sai2_primitives_perso_amd as pkg
import test_gpu_parity as tp


def err(tau, ref):
    return np.abs(tau - ref).max(axis=0) / np.maximum(np.abs(ref).max(axis=0), 1.0)


if len(sys.argv) > 1 and sys.argv[1] == "prefer_cert":
    os.environ["SAI2B_PREFER_CERT"] = "1"
for dec in (0, 1, 2):
    B = 4096
    tasks = [("mft", {"partial": None}), ("jt", {"selection": None})]
    inp = tp._custom_inputs(tasks, B, seed=700 + dec, singular_fraction=0.1)
    go, gg = ol.task_configs(inp["tasks"]), pkg.task_configs(inp["tasks"])
    for cfgs in (go, gg):
        for c in cfgs:
            c.dynamic_decoupling_type = dec
    o = ol.Oracle(ol.panda_model(), go, B, threads=8)
    g = pkg.Controller(pkg.panda_model(), gg, B)
    for c in (o, g):
        ol.load_inputs(c, inp)
    for tick in range(3):
        to, tg = o.tick(), g.tick()
    _, _, ro = o.get_mft_singularity(0)
    e = err(tg, to)
    _, c1o, c2o = o.get_mft_sh_state(0)
    n, c1, c2 = g.get_mft_singularity_state(0)
    print(f"dec {dec}: ranks {dict(zip(*np.unique(ro, return_counts=True)))} fallback {g.fallback_count()} err regular {e[ro == 6].max():.1e} "
          f"one direction {e[ro == 5].max() if (ro == 5).any() else 0:.1e} more {e[ro < 5].max() if (ro < 5).any() else 0:.1e} state equal "
          f"{np.array_equal(c1, c1o) and np.array_equal(c2, c2o)}")
B = 65536
for frac in (0.0, 0.1):
    inp = tp._custom_inputs([("mft", {"partial": None}), ("jt", {"selection": None})], B, seed=9, singular_fraction=frac)
    g = pkg.Controller(pkg.panda_model(), pkg.task_configs(inp["tasks"]), B)
    pkg.workloads.load_inputs(g, inp)
    for _ in range(5):
        g.tick(want_output=False)
    g.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        g.tick(want_output=False)
    g.synchronize()
    dt = (time.perf_counter() - t0) / 50
    k, fb = g.profile_tick(20)
    print(f"65536 robots, singular fraction {frac}: {dt * 1e6:.1f} us per tick (first kernel {k * 1e3:.1f} us, pass {fb * 1e3:.1f} us over {g.fallback_count()} robots)")
