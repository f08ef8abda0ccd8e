import sys, time, os
sys.path[:0]=['/root/repo']
import numpy as np, torch
import sai2_primitives_perso_amd as pkg
B=65536
inp = pkg.workloads.make_inputs(3, B=B)
c = pkg.Controller(pkg.panda_model(), pkg.task_configs(inp["tasks"]), B)
pkg.workloads.load_inputs(c, inp)
for _ in range(5): c.tick(want_output=False)
c.synchronize(); torch.cuda.synchronize()
def t(f, n=200):
    t0=time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter()-t0)/n*1e6
print("idle ctrl.synchronize us", t(c.synchronize))
print("idle torch.cuda.synchronize us", t(torch.cuda.synchronize))
print("sharding.barrier us", t(pkg.sharding.barrier))
# enqueue cost
t0=time.perf_counter()
for _ in range(200): c.tick(want_output=False)
t1=time.perf_counter(); c.synchronize(); t2=time.perf_counter()
print("enqueue per tick us", (t1-t0)/200*1e6, "drain us", (t2-t1)*1e6)
for steps in (20, 20, 20, 200):
    c.synchronize(); torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(steps): c.tick(want_output=False)
    c.synchronize()
    t1=time.perf_counter()
    torch.cuda.synchronize()
    t2=time.perf_counter()
    print(steps, "per step us", (t1-t0)/steps*1e6, "incl torch sync", (t2-t0)/steps*1e6)
