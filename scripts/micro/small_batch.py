#!/usr/bin/env python3
"""Time per tick of the C3 / C2 hierarchies at small batches: the one-lane SVD-free kernel (+ its empty work-list pass) against
the lanes-per-robot generic kernel for the whole batch (SAI2B_NO_FAST_PATH=1, 16 lanes per robot)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import numpy as np

import sai2_primitives_perso_amd as pkg

for config in (3, 2):
    for B in (256, 1024, 2048, 4096, 8192, 16384):
        row = []
        for env in ({}, {"SAI2B_NO_FAST_PATH": "1", "SAI2B_GENERIC_LANES": "16"}, {"SAI2B_NO_FAST_PATH": "1", "SAI2B_GENERIC_LANES": "8"}):
            for k, v in env.items():
                os.environ[k] = v
            inp = pkg.workloads.make_inputs(config, B=B)
            c = pkg.Controller(pkg.panda_model(), pkg.task_configs(inp["tasks"]), B)
            for k in env:
                del os.environ[k]
            pkg.workloads.load_inputs(c, inp)
            for _ in range(20):
                c.tick(want_output=False)
            c.synchronize()
            t0 = time.perf_counter()
            for _ in range(300):
                c.tick(want_output=False)
            c.synchronize()
            row.append((time.perf_counter() - t0) / 300 * 1e6)
        print(f"C{config} B={B:6d}: one lane per robot {row[0]:6.1f} us | 16 lanes per robot (generic) {row[1]:6.1f} us | 8 lanes {row[2]:6.1f} us")
