#!/usr/bin/env python3
"""Experiment: shard the 65 536-robot C3 batch over K contexts (K HIP streams) on ONE GPU, so that the
load phase of one shard can overlap the compute phase of another."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch  # noqa: F401

import sai2_primitives_perso_amd as pkg

B = 65536
for K in (1, 2, 4, 8):
    ctrls = []
    for k in range(K):
        inp = pkg.workloads.make_inputs(3, B=B // K, rank=k)
        c = pkg.Controller(pkg.panda_model(), pkg.task_configs(inp["tasks"]), B // K)
        pkg.workloads.load_inputs(c, inp)
        ctrls.append(c)
    for _ in range(20):
        for c in ctrls:
            c.tick(want_output=False)
    for c in ctrls:
        c.synchronize()
    steps = 300
    t0 = time.perf_counter()
    for _ in range(steps):
        for c in ctrls:
            c.tick(want_output=False)
    for c in ctrls:
        c.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"K={K}: {dt * 1e6:.1f} us per 65536-robot tick  {B / dt / 1e6:.0f} Mticks/s")
    for c in ctrls:
        c.close()
