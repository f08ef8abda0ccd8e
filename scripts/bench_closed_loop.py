#!/usr/bin/env python3
"""Closed-loop throughput: controller tick + one simulation step per control period,
nothing leaving the device — the examples' two threads (examples/05-...cpp:143-196, :215-236) for 65 536
robots at once. Goals are re-drawn every 500 periods, so with the internal OTG on the robots are
moving most of the time."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch  # noqa: F401

import sai2_primitives_perso_amd as pkg

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
inp = pkg.workloads.make_inputs(3, B=B)
for otg in (False, True):
    cfg = [pkg.motion_force_task_config("m", internal_otg=otg), pkg.joint_task_config("j", internal_otg=otg)]
    c = pkg.Controller(pkg.panda_model(), cfg, B)
    c.set_state(inp["q"], np.zeros_like(inp["q"]))
    c.reinitialize()
    goals = []
    for k in range(2):
        g = pkg.workloads.make_inputs(3, B=B, seed=100 + k)
        goals.append((torch.as_tensor(inp["mft0"]["pos"] + 0.04 * (k + 1), device="cuda"), torch.as_tensor(inp["mft0"]["rot"], device="cuda"),
                      torch.as_tensor(inp["q"] + 0.1 * (k + 1), device="cuda")))

    def run(n):
        for i in range(n):
            if i % 500 == 0:
                p, R, q = goals[(i // 500) % 2]
                c.set_mft_goals(0, p, R, None, None, None, None)
                c.set_jt_goals(1, q, None, None)
            c.tick(want_output=False)
            c.sim_step(None, 0.001, 1)
        c.synchronize()

    run(50)
    t0 = time.perf_counter()
    run(STEPS)
    dt = (time.perf_counter() - t0) / STEPS
    q, dq = c.get_state()
    print(f"  robots in the generic kernel at the last period: {c.fallback_count()} of {B}")
    print(f"closed loop, OTG {'on ' if otg else 'off'}: {dt * 1e6:7.1f} us per control period  {B / dt / 1e9:.2f} G robot-periods/s  "
          f"(simulated time / wall time per robot = {0.001 / dt:.1f}x, max|dq| = {np.abs(dq).max():.3f})")
    c.close()
