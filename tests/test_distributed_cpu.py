"""world_size-2 gloo test of the N > 1 path: batch sharding with no data-path
collective. Each rank builds its own shard of the C3 workload, the oracle stands in for the GPU tick
(CPU only here), and the timing protocol (barrier + MAX over ranks) is exercised for real."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import oracle_lib as ol
    import sai2_primitives_perso_amd as pkg

    r, w = pkg.sharding.init("gloo")
    assert (r, w) == (rank, world)
    B = 96
    inp = pkg.workloads.make_inputs(5, B=B, rank=rank)  # C5 = C3 sharded
    o = ol.Oracle(ol.panda_model(), ol.task_configs(inp["tasks"]), B)
    ol.load_inputs(o, inp)
    pkg.sharding.barrier()
    tau = o.tick()
    elapsed = 1.0 + rank  # deterministic "timings": the MAX must be the slowest rank's
    (mx,) = pkg.sharding.max_over_ranks([elapsed])
    assert mx == float(world)
    value = pkg.sharding.node_throughput(B, world, steps=10, elapsed_max=mx)
    assert value == B * world * 10 / world
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), q=inp["q"], tau=tau)
    pkg.sharding.finalize()


def test_two_rank_gloo_sharding(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a, b = (np.load(tmp_path / f"rank{r}.npz") for r in range(world))
    assert not np.array_equal(a["q"], b["q"]), "ranks must own different robots"
    assert np.isfinite(a["tau"]).all() and np.isfinite(b["tau"]).all()
    # a rank's shard does not depend on the world size or on the other ranks (no exchange step)
    sys.path.insert(0, ROOT)
    import oracle_lib as ol
    import sai2_primitives_perso_amd as pkg

    inp = pkg.workloads.make_inputs(5, B=96, rank=1)
    o = ol.Oracle(ol.panda_model(), ol.task_configs(inp["tasks"]), 96)
    ol.load_inputs(o, inp)
    assert np.array_equal(o.tick(), b["tau"])


def test_shard_bounds_cover_the_batch():
    import sai2_primitives_perso_amd as pkg

    for total, world in ((524288, 8), (1000, 3), (7, 8)):
        cuts = [pkg.sharding.shard_bounds(total, world, r) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == total
        assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
