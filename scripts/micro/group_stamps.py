#!/usr/bin/env python3
"""Where one robot's cycles go in the lanes-per-robot generic kernel: builds nothing itself — run with
SAI2B_LIB pointing at a library whose sai2b_group.hip was compiled with -DSAI2B_GROUP_STAMP, which makes lane 0 of
workgroup 0 record (mark id, cycle counter) at every phase mark (sai2b_group_tick.hpp: GMARK).
Usage: group_stamps.py <config> <lanes> [singular]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
os.environ["SAI2B_NO_FAST_PATH"] = "1"
os.environ["SAI2B_GENERIC_LANES"] = sys.argv[2] if len(sys.argv) > 2 else "16"
import numpy as np
import torch  # noqa: F401

import sai2_primitives_perso_amd as pkg

NAMES = ['mft_begin', 'mft_jp_done', 'mft_cert_done', 'mft_branch_done', 'mft_lambda_done', 'mft_law_done', 'mft_sing_done', 'mft_end',
         'jt_begin', 'jt_jp_done', 'jt_range_done', 'jt_torque_done', 'jt_end', 'model_begin', 'model_fk_done', 'model_crba_done',
         'model_minv_done', 'model_end']
config = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = 4096
inp = pkg.workloads.make_inputs(config, B=B)
if len(sys.argv) > 3:
    import make_golden

    inp = make_golden.make_singular(inp)
c = pkg.Controller(pkg.panda_model(), pkg.task_configs(inp["tasks"]), B)
pkg.workloads.load_inputs(c, inp)
lib = c.lib
hip = C.CDLL("libamdhip64.so")
for _ in range(3):
    c.tick(want_output=False)
c.synchronize()


lib.sai2b_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 512)()
lib.sai2b_debug_reset_stamps()
c.tick(want_output=False)
c.synchronize()
n = lib.sai2b_debug_read_stamps(buf, 512)
prev = None
tot = 0
for k in range(n):
    i, t = buf[2 * k], buf[2 * k + 1]
    if prev is not None:
        print(f"{NAMES[pi]:<18} -> {NAMES[i]:<18} {t - prev:>8} cycles")
        tot += t - prev
    prev, pi = t, i
print("total", tot, "cycles between the first and the last mark")
