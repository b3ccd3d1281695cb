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
        if plan_name == "Q3ColocatedStaged":
            local = list(plan.execute(0, tc))                    # built once, executed again: the second run must not see the first one's state
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
                                             (2, "q3_colocated_plan"), (3, "q3_colocated_plan"), (2, "Q3ColocatedStaged"), (3, "Q3ColocatedStaged")])
def test_q3_distributed_two_ranks_one_gpu_matches_oracle(world, plan_name):
    import torch.multiprocessing as mp
    from dfgpu import tpch
    from oracle import pyoracle as po
    from test_gpu_q3 import canon
    sf = 0.05
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 1000) + world + {"q3_distributed_plan": 0, "q3_broadcast_plan": 10, "q3_colocated_plan": 20, "Q3ColocatedStaged": 30}[plan_name]
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


def _utf8_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    try:
        import faulthandler
        faulthandler.dump_traceback_later(120, exit=True)
        import pyarrow as pa
        import torch
        import torch.distributed as dist
        import dfgpu
        from dfgpu import exchange, physical_plan as ops
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        ctx = dfgpu.Context(0, stream=torch.cuda.current_stream().cuda_stream) if rank == 0 else dfgpu.Context(0)
        rng = np.random.default_rng(100 + rank)
        n = [0, 3000, 1777][rank % 3] if world == 3 else [2500, 1300][rank]                    # one rank of three holds no rows at all
        k = rng.integers(0, 500, n).astype(np.int64)
        s_col = pa.array([None if v % 9 == 0 else f"rank{rank}-" + "x" * int(v % 23) for v in k], type=pa.utf8())
        t_col = pa.array([f"{v:05d}" for v in k], type=pa.utf8())
        tab = pa.table({"k": pa.array(k), "s": s_col, "t": t_col})
        C = ops.Column
        outs = {}
        if n:
            b = ops.batch_from_arrow(ctx, tab)
            src = ops.MemoryExec([[b]], b.schema)
        else:
            b0 = ops.batch_from_arrow(ctx, pa.table({"k": pa.array([], type=pa.int64()), "s": pa.array([], type=pa.utf8()), "t": pa.array([], type=pa.utf8())}))
            src = ops.MemoryExec([[b0]], b0.schema)
        tc = ops.TaskContext(ctx, 8192)
        for name, node in (("shuffle", exchange.ShuffleExec(src, [C("k", 0)])), ("broadcast", exchange.BroadcastExec(src))):
            got = [x for x in node.execute(0, tc)]
            rows = []
            for x in got:
                cols = [c.to_arrow().to_pylist() for c in x.columns]
                rows += list(zip(*cols))
            outs[name] = rows
        q.put((rank, {"mine": list(zip(k.tolist(), s_col.to_pylist(), t_col.to_pylist())), **outs}))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # noqa
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("world", [2, 3])
def test_exchange_of_utf8_and_nullable_utf8_columns(world):
    """ShuffleExec / BroadcastExec over batches with Utf8 columns (one nullable): offsets + value bytes travel inside the packed
    per-destination message; a rank without rows still takes part.  Shuffle: every row arrives exactly once, rows with equal keys on
    one rank; broadcast: every rank holds every rank's rows in rank order."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 900) + world
    procs = [ctx.Process(target=_utf8_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=150) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for r in range(world):
        assert isinstance(results[r], dict), results[r]
    everything = [row for r in range(world) for row in results[r]["mine"]]
    shuffled = [row for r in range(world) for row in results[r]["shuffle"]]
    assert sorted(shuffled, key=repr) == sorted(everything, key=repr)
    owner = {}
    for r in range(world):
        for row in results[r]["shuffle"]:
            assert owner.setdefault(row[0], r) == r, "equal keys must land on one rank"
    for r in range(world):
        assert results[r]["broadcast"] == everything
