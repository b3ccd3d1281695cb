"""-m gpu: the C entry point dfgpu_exchange (csrc/exchange.hip) -- RepartitionExec's exchange between ranks.

Several ranks share ONE GPU (RCCL refuses two ranks on one device), so the transport is the caller-provided one: callbacks over
torch.distributed's gloo backend that stage the device buffers through the host (exchange.Comm).  Everything else is the production path:
one-pass partition of all columns (dfgpu_partition_columns), the metadata all-gather, one collective per column lane, validity bytes back
to bitmaps.  Checked: every row reaches the rank hash % world names (the oracle's BatchPartitioner restatement decides), rows arrive ordered
by source rank and in input order inside a source, nullable columns keep their NULLs, a rank without any batch takes part, and the
world-1 RCCL communicator (ncclCommInitRank inside libdfgpu.so) moves data through the same entry point."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_rows(rank, n):
    import pyarrow as pa
    rng = np.random.default_rng(700 + rank)
    k = rng.integers(0, 10**6, n).astype(np.int64)
    a = pa.array(rng.integers(-50, 50, n).astype(np.int32), mask=(rng.random(n) < 0.2) if rank != 1 else None)      # nullable on some ranks only
    d = pa.array(rng.random(n))
    import decimal
    m = pa.array([decimal.Decimal(int(v)).scaleb(-2) for v in rng.integers(-10**9, 10**9, n)], type=pa.decimal128(15, 2))
    return pa.table({"k": pa.array(k), "a": a, "d": d, "m": m})


def _worker(rank, world, port, q, sizes, via_plan):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    try:
        import faulthandler
        faulthandler.dump_traceback_later(150, exit=True)
        import pyarrow as pa
        import torch
        import torch.distributed as dist
        import dfgpu
        from dfgpu import exchange, physical_plan as ops
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        ctx = dfgpu.Context(0, stream=torch.cuda.current_stream().cuda_stream) if rank % 2 == 0 else dfgpu.Context(0)
        tab = make_rows(rank, sizes[rank])
        if via_plan:
            tc = ops.TaskContext(ctx, 8192)
            if sizes[rank]:
                b = ops.batch_from_arrow(ctx, tab); src = ops.MemoryExec([[b]], b.schema)
            else:
                b = ops.batch_from_arrow(ctx, tab.slice(0, 0)); src = ops.MemoryExec([[b]], b.schema)
            node = exchange.ShuffleExec(src, [ops.Column("k", 0)], native=True)
            got = [x for x in node.execute(0, tc)]
            out = pa.concat_tables([x.to_arrow() for x in got]) if got else tab.slice(0, 0)
            q.put((rank, {c: out[c].to_pylist() for c in out.column_names}))
        else:
            comm = exchange.Comm(ctx, None)
            assert comm.kind == "callbacks"
            cols = [ctx.from_arrow(tab[c].combine_chunks()) for c in tab.column_names] if sizes[rank] else None
            got, sent, recv = comm.exchange([cols[0]] if cols else None, cols, 4, None)
            res = {"sent": sent, "recv": recv}
            if got is not None:
                for name, arr in zip(tab.column_names, got):
                    res[name] = arr.to_arrow().to_pylist()
            q.put((rank, res))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # noqa
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("world,sizes,via_plan", [(2, [30000, 12345], False), (3, [20000, 0, 7777], False), (3, [5000, 9000, 1], True), (2, [0, 4000], True)],
                         ids=["2-ranks", "3-ranks-one-empty", "3-ranks-through-ShuffleExec", "2-ranks-through-ShuffleExec-one-empty"])
def test_native_exchange_routes_rows_like_the_oracle(world, sizes, via_plan):
    import pyarrow as pa
    import torch.multiprocessing as mp
    from oracle import pyoracle as po
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 900) + world * 3 + int(via_plan)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, sizes, via_plan)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=170) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for r in range(world):
        assert isinstance(results[r], dict), results[r]
    # expectation: per destination, the rows of rank 0, then rank 1, ... each in input order (BatchPartitioner keeps it)
    want = {d: {c: [] for c in ("k", "a", "d", "m")} for d in range(world)}
    for src in range(world):
        if not sizes[src]:
            continue
        t = make_rows(src, sizes[src])
        idx, counts = po.hash_partition([t["k"].combine_chunks()], world)
        off = 0
        for d in range(world):
            part = t.take(pa.array(idx[off:off + counts[d]])); off += counts[d]
            for c in want[d]:
                want[d][c] += part[c].to_pylist()
    for d in range(world):
        got = results[d]
        for c in ("k", "a", "d", "m"):
            assert got.get(c, []) == want[d][c], (d, c)
        if not via_plan:
            assert sum(got["recv"]) == len(want[d]["k"])


def test_rccl_communicator_world_one(ctx):
    """the RCCL transport itself (ncclCommInitRank / grouped ncclSend + ncclRecv inside libdfgpu.so) at world size 1: rows come back grouped as
    one partition, in input order"""
    import pyarrow as pa
    import torch
    import torch.distributed as dist
    from dfgpu import exchange
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(29650 + os.getpid() % 300))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        comm = exchange.Comm(ctx, None)
        assert comm.kind == "rccl"
        t = make_rows(0, 50000)
        cols = [ctx.from_arrow(t[c].combine_chunks()) for c in t.column_names]
        got, sent, recv = comm.exchange([cols[0]], cols, 4, None)
        assert sent == [50000] and recv == [50000]
        for name, arr in zip(t.column_names, got):
            assert arr.to_arrow().equals(t[name].combine_chunks()), name
    finally:
        dist.destroy_process_group()
