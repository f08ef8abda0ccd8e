import sys, zlib
sys.path[:0]=['/root/repo','/root/repo/tests']
import numpy as np
import oracle_lib as ol
import sai2_primitives_perso_amd as pkg
import test_gpu_parity as tp
B=2048
for name in tp.HIERARCHIES:
    inp = tp._custom_inputs(tp.HIERARCHIES[name], B, seed=zlib.crc32(name.encode()) % 1000, singular_fraction=0.05)
    o, g = tp._pair(inp, introspection=False)
    for c in (o, g):
        ol.load_inputs(c, inp)
    for tick in range(2):
        tau_o, tau_g = o.tick(), g.tick()
    regular = np.ones(B, dtype=bool)
    for t, (kind, _) in enumerate(inp["tasks"]):
        if kind == "mft":
            _, _, ro = o.get_mft_singularity(t)
            regular &= ro == (o.tasks[t].pos_range + o.tasks[t].ori_range)
    e = tp._err(tau_g, tau_o)
    print(f"{name:28s} singular {int((~regular).sum()):4d} fallback {g.fallback_count():5d}  err regular {e[regular].max():.1e} singular {e[~regular].max() if (~regular).any() else 0:.1e}")
