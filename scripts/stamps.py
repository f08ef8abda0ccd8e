#!/usr/bin/env python3
"""Diagnostic: per-phase s_memtime shares of tick_fast_kernel (variant-5 build). Never quote this
build's run time; read the shares."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ["SAI2B_STAMPS"] = "1"
import numpy as np

import oracle_lib as ol
import sai2_primitives_perso_amd as pkg

B = 65536
inp = pkg.workloads.make_inputs(3, B=B)
c = pkg.Controller(pkg.panda_model(), pkg.task_configs(inp["tasks"]), B)
ol.load_inputs(c, inp)
for _ in range(20):
    c.tick(want_output=False)
c.synchronize()
W = B // 64
buf = np.zeros((16, W), dtype=np.uint64)
c.lib.sai2b_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
assert c.lib.sai2b_debug_stamps(c.h, buf.ctypes.data) == 0
t = buf[:8].astype(np.int64)
t0 = t[0].min()
names = ["start->loads issued", "FK+J", "law", "CRBA", "certify+vote", "rest (Cholesky, nullspace, JT)", "stores drained"]
print("wave start skew (cycles): p50 %d p99 %d" % (np.percentile(t[0] - t0, 50), np.percentile(t[0] - t0, 99)))
for k in range(7):
    d = t[k + 1] - t[k]
    print("%-34s median %7d  p90 %7d cycles (100 MHz ticks x?)" % (names[k], np.median(d), np.percentile(d, 90)))
print("total median", np.median(t[7] - t[0]))
