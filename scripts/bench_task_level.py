#!/usr/bin/env python3
"""A manual hierarchy (examples 04 / 18: MotionForceTask, then JointTask in its nullspace, tasks driven through the
TemplateTask calls) on 65 536 robots: the nullspace chained through the host as the reference's Eigen code does it,
chained on the device (sai2b_device_buffer(SAI2B_BUF_TASK_N_TOTAL)), and the same hierarchy as one fused tick."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch  # noqa: F401

import sai2_primitives_perso_amd as pkg

B = 65536
inp = pkg.workloads.make_inputs(3, B=B)
c = pkg.Controller(pkg.panda_model(), pkg.task_configs(inp["tasks"]), B)
pkg.workloads.load_inputs(c, inp)
t0_dev, t1_dev = (torch.empty((7, B), dtype=torch.float64, device="cuda") for _ in range(2))


def host_chain():
    c.task_update_model(0, None)
    c.task_update_model(1, c.task_nullspaces(0)[2])
    return c.task_compute_torques(0) + c.task_compute_torques(1)


def device_chain():
    c.task_update_model(0, None)
    c.task_update_model_behind(1, 0)
    c.task_compute_torques(0, out=t0_dev)
    c.task_compute_torques(1, out=t1_dev)
    return t0_dev + t1_dev


def fused():
    return c.tick(out=t0_dev)


for name, fn, n in (("nullspace through the host", host_chain, 5), ("nullspace on the device", device_chain, 30), ("fused tick", fused, 100)):
    fn()
    c.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    c.synchronize()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{name:<28}: {dt * 1e6:9.1f} us per period  {B / dt / 1e6:8.1f} M robot-periods/s")
