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


def _worker_planes(rank, world, port, q):
    """Ray-range sharding: per launch a private int32 plane, ONE all-reduce of all planes, accumulate replayed."""
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
    lamps = r["lamps"][:2]
    iters, n = 2, 30001
    comp = orc.Computation(s, lamps, n * len(lamps), r["lightHeight"], r["lightLength"], r["lightIntensity"], nthreads=2)
    comp.reset()
    pos = [tuple(float(x) for x in comp.lamp_world_pos(l)) for l in lamps]
    seeds, final = sharding.seed_chain(pos, r["lightLength"], iters)
    first, mine = sharding.ray_range(rank, world, n)
    planes = np.zeros((iters * len(lamps), s.T), dtype=np.int32)
    for k, li in sharding.launches(pos, iters):
        rays, _ = orc.generate(first, mine, pos[li], r["lightLength"], seeds[k])
        orc.extend(planes[k], s.tris, rays, s.nodes, s.triIdx, 2)
    t = torch.from_numpy(planes)
    sharding.reduce_planes(t)                               # the one collective
    for k, li in sharding.launches(pos, iters):             # replay accumulate.cl per launch, in order
        orc.accumulate(comp.photonMap, comp.maxPhotonMap, planes[k], lamps[li][2])
    q.put((rank, comp.photonMap.copy(), comp.maxPhotonMap.copy(), final, first, mine))
    dist.barrier()
    dist.destroy_process_group()


def test_ray_range_shards_with_one_plane_reduction_equal_one(orc, oscene, oroute):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_planes, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    lamps = oroute["lamps"][:2]
    comp = orc.Computation(oscene, lamps, 60002, oroute["lightHeight"], oroute["lightLength"], oroute["lightIntensity"])
    assert comp.photonsPerLight == 30001 - 1        # (60002 / 2) & ~1: the workers trace one photon more per launch
    comp.photonsPerLight = 30001
    comp.reset()
    comp.iteration()
    comp.iteration()
    assert sorted((f, m) for _, _, _, _, f, m in res) == [(0, 15001), (15001, 15000)]
    for rank, pm, mx, final, _, _ in res:
        assert final == comp.SEED
        assert np.array_equal(pm, comp.photonMap) and pm.any()
        assert np.array_equal(mx, comp.maxPhotonMap)


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
