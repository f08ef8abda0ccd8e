#!/usr/bin/env python3
"""Times one tick of 65 536 robots for the two other robots of the reference's examples — the 8-joint Panda on a
prismatic base with example 06's hierarchy [partial JointTask(2), MotionForceTask(6), JointTask(8)] and the planar 4R
with example 11's [planar MotionForceTask(3), JointTask(4)] — with the SVD-free kernel for general hierarchies
(sai2b_cert.hip) in front of the generic kernel, and with the generic kernel alone (SAI2B_NO_CERT_PATH=1)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch  # noqa: F401

import robots
import sai2_primitives_perso_amd as pkg

B = 65536


def make(robot):
    m, links = pkg.model_from_urdf(robots.TEXT[robot](), is_file=False)
    n = m.dof
    rng = np.random.default_rng(5)
    lo, hi = np.array(list(m.q_lower)[:n]), np.array(list(m.q_upper)[:n])
    q = 0.5 * (lo + hi)[:, None] + 0.3 * (hi - lo)[:, None] * rng.uniform(-1, 1, (n, B))
    dq = rng.normal(0, 0.2, (n, B))
    if robot == "sliding_base":
        link, fpos, frot = pkg.resolve_link_frame(links, "end-effector", (0.0, 0.0, 0.07))
        sel = np.zeros((2, n))
        sel[0, 0] = sel[1, 7] = 1
        cfgs = [pkg.joint_task_config("partial_joint_task", sel, internal_otg=False, robot_dof=n),
                pkg.motion_force_task_config("motion_force_task", link, fpos, frot, internal_otg=False, robot_dof=n),
                pkg.joint_task_config("joint_task", None, internal_otg=False, robot_dof=n)]
    else:
        link, fpos, frot = pkg.resolve_link_frame(links, "link4", (0.5, 0.0, 0.0))
        partial = (np.array([[1.0, 0, 0], [0, 1.0, 0]]), np.array([[0, 0, 1.0]]))
        cfgs = [pkg.motion_force_task_config("motion_force_task", link, fpos, frot, partial, internal_otg=False, robot_dof=n),
                pkg.joint_task_config("joint_task", None, internal_otg=False, robot_dof=n)]
    c = pkg.Controller(m, cfgs, B)
    c.set_state(q, dq)
    c.reinitialize()
    return c


for robot in robots.TEXT:
    for no_cert in ("0", "1"):
        os.environ["SAI2B_NO_CERT_PATH"] = no_cert
        c = make(robot)
        for _ in range(5):
            c.tick(want_output=False)
        c.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            c.tick(want_output=False)
        c.synchronize()
        dt = (time.perf_counter() - t0) / 30
        k_ms, fb_ms = c.profile_tick(30)
        what = "generic kernel alone" if no_cert == "1" else f"SVD-free kernel {k_ms * 1e3:.1f} us + work-list pass {fb_ms * 1e3:.1f} us over {c.fallback_count()} robots"
        print(f"{robot}: {dt * 1e6:.1f} us/tick  {B / dt / 1e6:.1f} Mticks/s   ({what})")
