"""Per-period breakdown of tests/test_gpu_fuzz.py for given seeds (worst robot: singular values, ranks,
per-task torques GPU vs oracle): python tests/diag_fuzz.py 150 591"""
import sys, os
HERE = os.path.dirname(os.path.abspath(__file__)); sys.path[:0] = [os.path.dirname(HERE), HERE, os.path.join(HERE, "golden")]
import numpy as np, zlib
import test_gpu_fuzz as F
import oracle_lib as ol, sai2_primitives_perso_amd as pkg
np.set_printoptions(linewidth=200, precision=6)
for seed in [int(a) for a in sys.argv[1:]] or [0]:
    rng = np.random.default_rng(9000 + seed)
    name = sorted(F.SHAPES)[seed % len(F.SHAPES)]
    tasks = F.SHAPES[name]; B = 192
    inp = F._custom_inputs(tasks, B, seed=zlib.crc32(name.encode()) % 1000 + seed, singular_fraction=0.1)
    opts = F._draw_opts(rng, tasks)
    otg = bool(rng.integers(2)); gravity = bool(rng.integers(2)); intro = bool(rng.integers(2))
    print(seed, name, opts, otg, gravity, intro)
    o = ol.Oracle(ol.panda_model(), F._configs(ol.task_configs, tasks, opts, otg), B, threads=8)
    g = pkg.Controller(pkg.panda_model(), F._configs(pkg.task_configs, tasks, opts, otg), B, introspection=True)
    wrench = {k: rng.normal(0, s, size=(3, B)) for k, s in (("f", 3.0), ("m", 0.5), ("sf", 3.0), ("sm", 0.5))}
    for c in (o, g):
        ol.load_inputs(c, inp); c.enable_gravity_compensation(gravity)
        for t, (kind, _) in enumerate(tasks):
            if kind == "mft" and "force_space_dimension" in opts[t]:
                c.set_mft_goal_wrench(t, wrench["f"], wrench["m"]); c.set_mft_sensed_wrench(t, wrench["sf"], wrench["sm"])
    for period in range(6):
        tau_o, tau_g = o.tick(), g.tick()
        den = np.maximum(np.abs(tau_o).max(axis=0), 1e-9)
        e = np.abs(tau_g - tau_o).max(axis=0) / den
        b = int(np.argmax(e))
        so, ao, ro = o.get_mft_singularity(0); sg, ag, rg = g.get_mft_singularity(0)
        print(" period", period, "worst robot", b, "err", e[b], "rank o/g", ro[b], rg[b], "alpha", ao[b], ag[b])
        print("   sigma o", so[:, b]); print("   sigma g", sg[:, b])
        for t in range(len(tasks)):
            to, tg = o.get_task_torques(t)[:, b], g.get_task_torques(t)[:, b]
            print("   task", t, "tau_o", to, "diff", np.abs(to - tg).max())
        No, Ng = o.get_task_nullspace(1)[:, b], g.get_task_nullspace(1)[:, b]
        print("   N after task1 diff", np.abs(No - Ng).max(), "|N|", np.abs(No).max())
        o.sim_step(tau_o, 0.001, 1, with_gravity=gravity); g.sim_step(tau_g, 0.001, 1, with_gravity=gravity)
