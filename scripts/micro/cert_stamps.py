#!/usr/bin/env python3
"""Where one wavefront's cycles go in the SVD-free kernel for general hierarchies: run with SAI2B_LIB pointing at a
library whose sai2b_cert.hip was compiled with -DSAI2B_CERT_STAMP (lane 0 of workgroup 0 records (mark, cycle
counter) at every CSTAMP of sai2b_cert.hpp). Usage: cert_stamps.py [config] [singular]: with `singular` the workload's
own poses (C4: one in ten near a singularity, so the first wavefront goes through the in-lane singular branch)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch  # noqa: F401

import sai2_primitives_perso_amd as pkg

NAMES = {0: "start", 1: "q loaded", 2: "fk", 3: "crba", 4: "gravity, factors", 10: "mft begin", 11: "mft q dq loaded", 12: "pose+velocity sweep",
         13: "law", 14: "jacobian sweep", 20: "level: Y, Jp", 21: "level: certificate", 22: "level: direct + bounded-inertia terms",
         23: "level: Gram-Schmidt + Lambda term", 24: "level: Q downdate", 30: "jt begin", 31: "jt law", 40: "task end",
         50: "level: certificate (failed: singular branch)", 51: "singular: one-sided Jacobi", 52: "singular: sort, split, U^T F, Y W",
         53: "singular: Lambda_s (and bounded-inertia forms)", 54: "singular: torques, Gram-Schmidt of the regular columns, Q'",
         55: "singular: state loads, classification (perturbed kinematics)", 56: "singular: history ring",
         57: "singular: posture columns Q' L^-1 V_s", 58: "singular: joint strategy torques, blend", 59: "singular: posture Gram-Schmidt, Q''"}
config = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = 4096
inp = pkg.workloads.make_inputs(config, B=B)
rng = np.random.default_rng(1)
if len(sys.argv) < 3 or sys.argv[2] != "singular":
    inp["q"] = np.ascontiguousarray(pkg.workloads.sample_poses(rng, B, reject_ratio=0.1).T)
c = pkg.Controller(pkg.panda_model(), pkg.task_configs(inp["tasks"]), B)
pkg.workloads.load_inputs(c, inp)
lib = c.lib
for _ in range(3):
    c.tick(want_output=False)
c.synchronize()
lib.sai2b_debug_read_cstamps.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 1024)()
lib.sai2b_debug_reset_cstamps()
c.tick(want_output=False)
c.synchronize()
n = lib.sai2b_debug_read_cstamps(buf, 1024)
prev = None
tot = 0
for k in range(n):
    i, t = buf[2 * k], buf[2 * k + 1]
    if prev is not None:
        print(f"-> {NAMES.get(i, i):<42} {t - prev:>8} cycles")
        tot += t - prev
    prev = t
print("total", tot, "cycles between the first and the last mark")
