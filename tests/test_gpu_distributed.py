"""-m gpu: the N>1 plan end to end on ONE GPU: 2 ranks (2 processes, both on cuda:0, gloo transport staged through host
memory because RCCL refuses two ranks on one device) run tpch.q3_distributed_plan over their shards -- device hash
partition, all-to-all per column, partitioned joins, Partial -> shuffle -> FinalPartitioned, per-rank sort, gather -- and
the gathered result must equal the CPU oracle row for row.  Only the transport differs from the 8-GPU RCCL run."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, sf, q, plan_name="q3_distributed_plan"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    try:
        import faulthandler
        faulthandler.dump_traceback_later(150, exit=True)      # a rank stuck in a collective must not hold the GPU box
        import torch
        import torch.distributed as dist
        import dfgpu
        from dfgpu import exchange, physical_plan as ops, tpch
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        # odd ranks run on a private stream: the exchange must then synchronise around the collectives by itself
        ctx = dfgpu.Context(0, stream=torch.cuda.current_stream().cuda_stream) if rank % 2 == 0 else dfgpu.Context(0)
        host = tpch.gen_host(sf)
        nc, no = len(host["c_custkey"]), len(host["o_orderkey"])
        c0, c1, o0, o1 = nc * rank // world, nc * (rank + 1) // world, no * rank // world, no * (rank + 1) // world
        lo_key, hi_key = host["o_orderkey"][o0], (host["o_orderkey"][o1] if o1 < no else np.iinfo(np.int64).max)
        lsel = (host["l_orderkey"] >= lo_key) & (host["l_orderkey"] < hi_key)         # lineitems stay with their orders
        shard = {k: (v[c0:c1] if k.startswith("c_") else v[o0:o1] if k.startswith("o_") else v[lsel]) for k, v in host.items()}
        tables = tpch.upload(ctx, shard)
        tc = ops.TaskContext(ctx, batch_size=8192)
        plan = getattr(tpch, plan_name)(tables)
        local = list(plan.execute(0, tc))
        mine = ops.concat_batches(local[0].schema, local) if local else None
        gathered = exchange.gather_batches(ctx, None, mine, 0, names=["l_orderkey", "revenue", "o_orderdate", "o_shippriority"])
        if rank == 0:
            q.put((rank, tpch.q3_result_to_numpy([gathered])))
        else:
            q.put((rank, "ok"))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # noqa
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("world,plan_name", [(2, "q3_distributed_plan"), (3, "q3_distributed_plan"), (2, "q3_broadcast_plan"), (3, "q3_broadcast_plan"),
                                             (2, "q3_colocated_plan"), (3, "q3_colocated_plan")])
def test_q3_distributed_two_ranks_one_gpu_matches_oracle(world, plan_name):
    import torch.multiprocessing as mp
    from dfgpu import tpch
    from oracle import pyoracle as po
    from test_gpu_q3 import canon
    sf = 0.05
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 1000) + world + {"q3_distributed_plan": 0, "q3_broadcast_plan": 10, "q3_colocated_plan": 20}[plan_name]
    procs = [ctx.Process(target=_worker, args=(r, world, port, sf, q, plan_name)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=170) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for r in range(1, world):
        assert results[r] == "ok", results[r]
    assert isinstance(results[0], dict), results[0]
    host = tpch.gen_host(sf)
    want = canon(po.tpch_q3(host, tpch.SEGMENTS.index(tpch.Q3_SEGMENT), tpch.Q3_DATE, 4))
    got = canon(results[0])
    assert len(got["l_orderkey"]) == len(want["l_orderkey"]) > 0
    for k in want:
        assert np.array_equal(got[k], want[k]), k
