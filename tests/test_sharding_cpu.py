"""CPU, world_size 2 over gloo: the launch-sharding host logic (launch ownership, SEED chain,
SUM/MAX reduction).  The per-launch kernels are stood in for by the oracle (tests may do that);
the product's sharding module does the partitioning and the collective."""
import os
import sys

import numpy as np
import pytest

from conftest import GLB, ROOT, ROUTE


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as g
    g.load_package()
    from uvrt_amd import sharding
    orc = g.load_oracle()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = orc.Scene(GLB)
    r = orc.load_route(ROUTE)
    lamps = r["lamps"][:3]
    iters, n = 2, 20000
    comp = orc.Computation(s, lamps, n * len(lamps), r["lightHeight"], r["lightLength"], r["lightIntensity"], nthreads=2)
    comp.reset()
    pos = [tuple(float(x) for x in comp.lamp_world_pos(l)) for l in lamps]
    seeds, final = sharding.seed_chain(pos, r["lightLength"], iters)
    for k, li in sharding.launches(pos, iters):
        if sharding.owner(k, world) != rank:
            continue
        comp.SEED = seeds[k]
        comp.single_light(lamps[li])
    sum_t = torch.from_numpy(comp.photonMap)
    max_t = torch.from_numpy(comp.maxPhotonMap)
    sharding.reduce_maps(sum_t, max_t)
    q.put((rank, comp.photonMap.copy(), comp.maxPhotonMap.copy(), final))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one(orc, oscene, oroute):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    lamps = oroute["lamps"][:3]
    comp = orc.Computation(oscene, lamps, 60000, oroute["lightHeight"], oroute["lightLength"], oroute["lightIntensity"])
    comp.reset()
    comp.iteration()
    comp.iteration()
    for rank, pm, mx, final in res:
        assert final == comp.SEED
        assert np.array_equal(pm, comp.photonMap)
        assert np.array_equal(mx, comp.maxPhotonMap)


def test_seed_chain_host_walk_matches_oracle(pkg, orc, oscene, oroute):
    from uvrt_amd import sharding
    comp = orc.Computation(oscene, oroute["lamps"], 1 << 16, oroute["lightHeight"], oroute["lightLength"],
                           oroute["lightIntensity"])
    pos = [tuple(float(x) for x in comp.lamp_world_pos(l)) for l in oroute["lamps"]]
    seeds, final = sharding.seed_chain(pos, oroute["lightLength"], 2)
    s = 0
    for k, li in sharding.launches(pos, 2):
        assert seeds[k] == s
        _, s = orc.generate(0, 1, pos[li], oroute["lightLength"], s)
    assert final == s
    assert seeds[1] == 0x79044923 and seeds[2] == 0xce0db3eb
