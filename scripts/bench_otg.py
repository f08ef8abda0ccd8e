#!/usr/bin/env python3
"""Cost of the tasks' internal OTG on the C3 workload (65 536 Panda, MotionForceTask + JointTask),
inputs resident in HBM: the fused tick with the generators off (BASELINE's definition), on but idle
(goals reached), on with every robot mid-trajectory, and re-planning every tick (goals change each
tick: the worst case). `python scripts/bench_otg.py 65536 jerk`: the same phases with both generators JERK-LIMITED
(enableInternalOtgJerkLimited: ruckig's third-order interface, csrc/sai2b_otg3_core.hpp; one lane plans one robot)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch  # noqa: F401

import sai2_primitives_perso_amd as pkg

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ONLY_MOVING = len(sys.argv) > 2 and sys.argv[2] == "moving"  # profiling runs: just the all-moving phase
JERK = len(sys.argv) > 2 and sys.argv[2] == "jerk"
inp = pkg.workloads.make_inputs(3, B=B)


def timed(c, steps, before=None):
    c.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        if before:
            before(k)
        c.tick(want_output=False)
    c.synchronize()
    return (time.perf_counter() - t0) / steps


def make(otg):
    cfg = [pkg.motion_force_task_config("m", internal_otg=otg), pkg.joint_task_config("j", internal_otg=otg)]
    if otg and JERK:  # the reference's default jerk limits (MotionForceTask.h:73-74, JointTask.h:42)
        cfg[0].internal_otg_jerk_limited = cfg[1].internal_otg_jerk_limited = 1
    c = pkg.Controller(pkg.panda_model(), cfg, B)
    c.set_state(inp["q"], inp["dq"])
    c.reinitialize()
    return c


if not ONLY_MOVING:
    c = make(False)
    pkg.workloads.load_inputs(c, inp)
    timed(c, 20)
    off = timed(c, 200)
    print(f"OTG off            : {off * 1e6:7.1f} us/step  {B / off / 1e9:.2f} G ticks/s" + ("   (generators below: jerk-limited)" if JERK else ""))
    c.close()

c = make(True)
timed(c, 20)  # goals = current pose: every generator finishes at once
if not ONLY_MOVING:
    idle = timed(c, 200)
    print(f"OTG on, idle       : {idle * 1e6:7.1f} us/step  {B / idle / 1e9:.2f} G ticks/s   reached={c.get_otg_status(1)[0].mean():.2f}")

# far goals: ~1 s trajectories, every robot moving during the timed region
far_q = inp["q"] + 0.8 * np.sign(np.random.default_rng(0).normal(size=inp["q"].shape))
pos = inp["mft0"]["pos"] + 0.25
qd = torch.as_tensor(far_q, device="cuda")
c.set_jt_goals(1, far_q, None, None)
c.set_mft_goals(0, pos, inp["mft0"]["rot"], None, None, None, None)
timed(c, 10)
mov = timed(c, 200)
r = c.get_otg_status(1)[0].mean(), c.get_otg_status(0)[0].mean()
print(f"OTG on, all moving : {mov * 1e6:7.1f} us/step  {B / mov / 1e9:.2f} G ticks/s   reached jt/mft={r[0]:.2f}/{r[1]:.2f}")

# goals change on the device every tick: every robot re-plans both generators every tick
G = [torch.as_tensor(far_q + 0.01 * k, device="cuda") for k in range(4)]
P = [torch.as_tensor(pos + 0.001 * k, device="cuda") for k in range(4)]


def regoal(k):
    c.set_jt_goals(1, G[k % 4], None, None)
    c.set_mft_goals(0, P[k % 4], None, None, None, None, None)


if ONLY_MOVING:
    c.close()
    sys.exit(0)

# 1 % of the robots get a new goal each tick (a different 1 % every time)
rng = np.random.default_rng(1)
S = []
for k in range(8):
    q = far_q.copy()
    idx = rng.choice(B, B // 100, replace=False)
    q[:, idx] += 0.05
    S.append(torch.as_tensor(q, device="cuda"))
    far_q = q


def sparse(k):
    c.set_jt_goals(1, S[k % 8], None, None)


timed(c, 8, sparse)
sp = timed(c, 96, sparse)
print(f"OTG on, 1% re-goal : {sp * 1e6:7.1f} us/step  {B / sp / 1e9:.2f} G ticks/s")

timed(c, 10, regoal)
rep = timed(c, 100, regoal)
print(f"OTG on, re-planning: {rep * 1e6:7.1f} us/step  {B / rep / 1e9:.2f} G ticks/s")
c.close()
