#!/usr/bin/env python3
"""Extra measurements for DESIGN.md: batch-size sweep of the fused tick (device-resident inputs) and the
PCIe-inclusive rate when every input comes from / every output goes to pageable host memory."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch  # noqa: F401

import sai2_primitives_perso_amd as pkg

print("batch sweep, C3 hierarchy, inputs resident in HBM")
for B in (4096, 16384, 65536, 131072, 262144, 524288):
    inp = pkg.workloads.make_inputs(3, B=B)
    c = pkg.Controller(pkg.panda_model(), pkg.task_configs(inp["tasks"]), B)
    pkg.workloads.load_inputs(c, inp)
    for _ in range(10):
        c.tick(want_output=False)
    c.synchronize()
    steps = 100
    t0 = time.perf_counter()
    for _ in range(steps):
        c.tick(want_output=False)
    c.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"  B={B:7d}: {dt * 1e6:8.1f} us/step  {B / dt / 1e6:8.1f} Mticks/s")
    if B == 65536:
        # PCIe-inclusive: state + all goals H2D, tick, torques D2H, every step (pageable numpy arrays)
        tau = np.empty((7, B))
        for _ in range(3):
            pkg.workloads.load_inputs(c, inp)
            c.tick(out=tau)
        t0 = time.perf_counter()
        for _ in range(10):
            pkg.workloads.load_inputs(c, inp)
            c.tick(out=tau)
        dt = (time.perf_counter() - t0) / 10
        nbytes = (14 + 24 + 21 + 7) * 8 * B
        print(f"  PCIe-inclusive (host in, host out): {dt * 1e3:.2f} ms/step  {B / dt / 1e6:.1f} Mticks/s  ({nbytes / dt / 1e9:.1f} GB/s over the link)")
    c.close()
