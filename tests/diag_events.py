"""Per-period breakdown of test_random_runtime_events_closed_loop for a seed: python tests/diag_events.py 14"""
import sys, os
HERE = os.path.dirname(os.path.abspath(__file__)); sys.path[:0] = [os.path.dirname(HERE), HERE, os.path.join(HERE, "golden")]
import numpy as np
import test_gpu_fuzz as F
np.set_printoptions(linewidth=220, precision=8)
seed = int(sys.argv[1]); upto = int(sys.argv[2]) if len(sys.argv) > 2 else 40
watch = int(sys.argv[3]) if len(sys.argv) > 3 else -1
rng, name, tasks, otg, o, g = F._event_run_setup(seed, introspection=None if os.environ.get("DIAG_DRAWN") else True)
env = {"gravity": False}
print(seed, name, "otg", otg)
for period in range(upto):
    entry = F._event(rng, o, g, tasks, period, env)
    tau_o, tau_g = F._control(rng, o, g)
    e = np.abs(tau_g - tau_o).max(axis=0) / np.maximum(np.abs(tau_o).max(axis=0), 1e-9)
    b = int(e.argmax()) if watch < 0 else watch
    print("period", entry, "worst robot", b, "err", e[b], "robots above 1e-6:", np.nonzero(e > 1e-6)[0][:20])
    for t, (kind, _) in enumerate(tasks):
        do, dg = (o.get_mft_desired(t), g.get_mft_desired(t)) if kind == "mft" else (o.get_jt_desired(t), g.get_jt_desired(t))
        so, sg = o.get_otg_status(t), g.get_otg_status(t)
        print("   task", t, kind, "desired diffs (robot b)", [float(np.abs(a[..., b] - c[..., b]).max()) for a, c in zip(do, dg)],
              "status o", [x[b] for x in so], "g", [x[b] for x in sg])
        if watch >= 0:
            print("      task torque (oracle)", o.get_task_torques(t)[:, b], "gpu diff", np.abs(o.get_task_torques(t)[:, b] - g.get_task_torques(t)[:, b]).max())
            if kind == "mft":
                fo, fg = o.get_mft_task_forces(t), g.get_mft_task_forces(t)
                print("      F_unit  oracle", fo[0][:, b], "gpu", fg[0][:, b])
                print("      F_force oracle", fo[1][:, b], "gpu", fg[1][:, b])
                import torch
                class _Raw:
                    __cuda_array_interface__ = {"data": (int(g.device_buffer(F.pkg._abi.BUF_STATE, t)), False), "shape": (12, g.B), "typestr": "<f8", "version": 2}
                g.synchronize()
                ig = torch.as_tensor(_Raw(), device="cuda").cpu().numpy()[:, b]
                io = o.get_mft_integrators(t)[:, b]
                print("      integrators oracle", io, "\n      integrators gpu   ", ig)
                c = o.tasks[t]
                print("      cfg: fdim", c.force_space_dimension, "mdim", c.moment_space_dimension, "cl", c.closed_loop_force, c.closed_loop_moment, "frame", c.parametrization_in_compliant_frame, "passivity", c.passivity_enabled)
        if e[b] > 1e-6 or watch >= 0:
            print("      oracle desired", [a[..., b] for a in do])
            print("      gpu    desired", [a[..., b] for a in dg])
    o.sim_step(tau_o, 0.001, 1, with_gravity=env["gravity"]); g.sim_step(tau_g, 0.001, 1, with_gravity=env["gravity"])
    g.set_state(*o.get_state())  # as the test does
