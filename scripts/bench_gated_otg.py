#!/usr/bin/env python3
"""C4 hierarchy [MFT(3), JT(2), JT(7)] with the tasks' internal OTG off / on. With it on, the 2-joint task's
generator is gated per robot (its range can be empty: DESIGN.md 8b), which costs a model-only pass ahead of the
generator kernels: the SVD-free cascade for the robots it can certify, the generic kernel for the others
(SAI2B_NO_CERT_PATH=1: the generic kernel for the whole batch, as before round 2's range_cert_kernel)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
import torch  # noqa: F401

import sai2_primitives_perso_amd as pkg

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
inp = pkg.workloads.make_inputs(4, B=B)
for otg in (False, True):
    cfgs = pkg.task_configs(inp["tasks"])
    for c in cfgs:
        c.use_internal_otg = int(otg)
    c = pkg.Controller(pkg.panda_model(), cfgs, B)
    pkg.workloads.load_inputs(c, inp)
    for _ in range(5):
        c.tick(want_output=False)
    c.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        c.tick(want_output=False)
    c.synchronize()
    dt = (time.perf_counter() - t0) / 30
    print(f"C4, OTG {'on ' if otg else 'off'}: {dt * 1e6:7.1f} us/step  {B / dt / 1e6:7.1f} Mticks/s")
    c.close()
