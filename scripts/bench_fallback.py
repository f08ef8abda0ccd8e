#!/usr/bin/env python3
"""Single-pass latency of the generic kernel (the fallback behind the SVD-free path) by case: which
hierarchy, and whether the robots are inside the singularity-blending region. One wavefront per SIMD at
B = 65 536, so the time of a launch is the critical path of one wavefront."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
os.environ["SAI2B_NO_FAST_PATH"] = "1"
import numpy as np
import torch  # noqa: F401

import make_golden
import sai2_primitives_perso_amd as pkg

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for label, config, singular in (("[MFT]      regular", 2, False), ("[MFT]      singular", 2, True),
                                ("[MFT, JT]  regular", 3, False), ("[MFT, JT]  singular", 3, True)):
    inp = pkg.workloads.make_inputs(config, B=B)
    if singular:
        inp = make_golden.make_singular(inp)
    c = pkg.Controller(pkg.panda_model(), pkg.task_configs(inp["tasks"]), B)
    pkg.workloads.load_inputs(c, inp)
    for _ in range(5):
        c.tick(want_output=False)
    c.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        c.tick(want_output=False)
    c.synchronize()
    dt = (time.perf_counter() - t0) / 30
    print(f"generic kernel, {label}: {dt * 1e6:7.1f} us/launch")
    c.close()
