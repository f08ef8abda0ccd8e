#!/usr/bin/env python3
"""The reference's own loop — updateControllerTaskModels(), goal setters, computeControlTorques() — against the fused
tick(), 65 536 robots, C3 and C4 hierarchies: sai2b_update_task_models() is deferred and consumed by the torque call
that follows, so the two-call form runs the same kernels as tick() (SVD-free kernel + work list)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch  # noqa: F401

import sai2_primitives_perso_amd as pkg

B = 65536
for config in (3, 4):
    inp = pkg.workloads.make_inputs(config, B=B)
    c = pkg.Controller(pkg.panda_model(), pkg.task_configs(inp["tasks"]), B)
    pkg.workloads.load_inputs(c, inp)
    out = torch.empty((7, B), dtype=torch.float64, device="cuda")

    def split():
        c.update_task_models()
        c.compute_control_torques(True, out=out)

    def fused():
        c.tick(out=out)

    for name, fn in (("update + compute", split), ("tick", fused)):
        for _ in range(10):
            fn()
        c.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            fn()
        c.synchronize()
        dt = (time.perf_counter() - t0) / 100
        print(f"C{config} {name:<16}: {dt * 1e6:7.1f} us/step  {B / dt / 1e6:8.1f} Mticks/s  (declined last step: {c.fallback_count()})")
