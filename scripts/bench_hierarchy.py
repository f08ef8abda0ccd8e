#!/usr/bin/env python3
"""Times the C4 hierarchy [MFT(3), JT(2), JT(7)] on regular poses (no injected singular robots) and on
the C4 workload: shows what the SVD-free certificates of the generic kernel buy when whole wavefronts
are regular."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch  # noqa: F401

import sai2_primitives_perso_amd as pkg

B = 65536
for label, frac in (("regular poses", 0.0), ("1 in 1000 near-singular", 0.001), ("C4 workload (10 % near-singular)", 0.10)):
    inp = pkg.workloads.make_inputs(4, B=B)
    rng = np.random.default_rng(1)
    q = pkg.workloads.sample_poses(rng, B, reject_ratio=0.1 if frac == 0 else None, singular_fraction=frac)
    inp["q"] = np.ascontiguousarray(q.T)
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
    k_ms, fb_ms = c.profile_tick(30)
    print(f"{label}: {dt * 1e6:.1f} us/step  {B / dt / 1e6:.1f} Mticks/s   (first kernel {k_ms * 1e3:.1f} us, work-list pass "
          f"{fb_ms * 1e3:.1f} us over {c.fallback_count()} robots)")
