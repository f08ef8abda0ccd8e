#!/usr/bin/env python3
"""Where the 6-row kernel with the in-lane singular branch (tick_cert_kernel<6, S6>) starts to beat the headline kernel + its
work-list pass: [MFT(6), JT(7)] on 65 536 robots, a growing share of the poses unfiltered (55 % of those are inside a blending
region). SAI2B_NO_SING6=1 against SAI2B_FORCE_SING6=1."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np

import sai2_primitives_perso_amd as pkg
import test_gpu_parity as tp

B = 65536
tasks = [("mft", {"partial": None}), ("jt", {"selection": None})]
base = pkg.workloads.make_inputs(3, B=B)
wild = tp._custom_inputs(tasks, B, seed=3, singular_fraction=0.0)
for share in (0.0, 0.05, 0.1, 0.2, 0.3, 0.4, 0.6, 1.0):
    inp = dict(base)
    n = int(share * B)
    q = base["q"].copy()
    q[:, :n] = wild["q"][:, :n]
    inp["q"] = q
    row = []
    for env in ("SAI2B_NO_SING6", "SAI2B_FORCE_SING6"):
        os.environ[env] = "1"
        g = pkg.Controller(pkg.panda_model(), pkg.task_configs(inp["tasks"]), B)
        del os.environ[env]
        pkg.workloads.load_inputs(g, inp)
        for _ in range(20):
            g.tick(want_output=False)
        g.synchronize()
        t0 = time.perf_counter()
        for _ in range(60):
            g.tick(want_output=False)
        g.synchronize()
        row.append(((time.perf_counter() - t0) / 60 * 1e6, g.fallback_count()))
    print(f"unfiltered share {share:4.2f}: headline kernel + pass {row[0][0]:6.1f} us ({row[0][1]:6d} declined) | 6-row kernel with the branch {row[1][0]:6.1f} us ({row[1][1]:6d} declined)")
