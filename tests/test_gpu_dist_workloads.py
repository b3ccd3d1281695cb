"""-m gpu: the N > 1 forms of BASELINE configs 4 and 5 (dist_workloads.py) on ONE GPU with 2 and 4 ranks over gloo: TPC-H Q5 as the fully partitioned
plan (11 exchanges) and the ClickBench Q28 shape (partial states shuffled on a Utf8 key).  The data set is the same at every world size (8 virtual
shards), so the gathered result must equal a direct numpy evaluation of the query over ALL shards -- exact Decimal128 sums, exact counts / max,
AVG within 1e-9 relative -- whatever the number of ranks; the native exchange (dfgpu_exchange under ShuffleExec) must give the same rows."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLOAT_RTOL = 1e-9


def q5_expected(tt):
    n = lambda k: tt[k].cpu().numpy()
    from dfgpu import dist_workloads as dw
    okey, ocust, odate = n("o_orderkey"), n("o_custkey"), n("o_orderdate")
    sel = (odate >= dw.Q5_DATE_LO) & (odate < dw.Q5_DATE_HI)
    cnat = np.zeros(int(n("c_custkey").max()) + 1, dtype=np.int64); cnat[n("c_custkey")] = n("c_nationkey")
    snat = np.zeros(int(n("s_suppkey").max()) + 1, dtype=np.int64); snat[n("s_suppkey")] = n("s_nationkey")
    order_nat = dict(zip(okey[sel].tolist(), cnat[ocust[sel]].tolist()))
    lkey, lsupp, price, disc = n("l_orderkey"), n("l_suppkey"), n("l_price")[:, 0], n("l_disc")[:, 0]
    keep = np.isin(lkey, okey[sel])
    lk, ls, lp, ld = lkey[keep], lsupp[keep], price[keep], disc[keep]
    onat = np.array([order_nat[k] for k in lk.tolist()], dtype=np.int64)
    m = (snat[ls] == onat) & (onat % 5 == 2)               # supplier in the customer's nation; nation in region ASIA (r_regionkey 2)
    rev = lp[m].astype(object) * (100 - ld[m].astype(object))
    out = {}
    for nat, r in zip(onat[m].tolist(), rev.tolist()):
        out[f"NATION{nat:02d}"] = out.get(f"NATION{nat:02d}", 0) + r
    return out


def _worker(rank, world, port, q, what, native):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    try:
        import faulthandler
        faulthandler.dump_traceback_later(170, exit=True)
        import torch
        import torch.distributed as dist
        import dfgpu
        from dfgpu import dist_workloads as dw, exchange, physical_plan as ops
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        ctx = dfgpu.Context(0, stream=torch.cuda.current_stream().cuda_stream)
        tc = ops.TaskContext(ctx, 8192)
        if what == "q5":
            tables = dw.q5_tables(ctx, dw.q5_tensors(0.05, rank, world), rank)
            plan = dw.q5_plan(tables, native=native); names = dw.Q5_OUTPUT
        else:
            ids, length, w = dw.clickbench_tensors(300000, 5000, rank, world)
            plan = dw.clickbench_plan(dw.clickbench_batch(ctx, ids, length, w, 5000)); names = dw.CLICKBENCH_OUTPUT
        local = list(plan.execute(0, tc))
        mine = ops.concat_batches(local[0].schema, local) if local else None
        gathered = exchange.gather_batches(ctx, None, mine, 0, names=names)
        if rank == 0:
            rows = gathered.to_arrow().to_pylist() if gathered.num_rows else []
            if what == "q5":
                want = q5_expected(dw.q5_tensors(0.05, 0, 1))
            else:
                ids, length, w = (t.cpu().numpy() for t in dw.clickbench_tensors(300000, 5000, 0, 1))
                want = (ids, length, w)
            q.put((rank, (rows, want, sum(nd.bytes_sent for nd in dw.shuffle_nodes(plan)))))
        else:
            q.put((rank, "ok"))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # noqa
        import traceback
        q.put((rank, traceback.format_exc()))


def run(world, what, native=False):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29300 + (os.getpid() % 600) + world * 7 + (3 if what == "q5" else 0) + int(native)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, what, native)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=175) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for r in range(world):
        assert not isinstance(results[r], str) or results[r] == "ok", results[r]
    return results[0]


@pytest.mark.parametrize("world,native", [(2, False), (4, False), (2, True)], ids=["2-ranks", "4-ranks", "2-ranks-native-exchange"])
def test_q5_partitioned_plan_equals_direct_evaluation(world, native):
    rows, want, sent = run(world, "q5", native)
    got = {r["n_name"]: int(r["revenue"].scaleb(4)) for r in rows}
    assert got == want and len(got) == 5
    assert sent > 0


@pytest.mark.parametrize("world", [2, 4])
def test_clickbench_shape_equals_direct_evaluation(world):
    rows, (ids, length, w), _ = run(world, "clickbench")
    keep = ids != 0
    cnt = np.bincount(ids[keep], minlength=5000); s = np.bincount(ids[keep], weights=length[keep].astype(np.float64), minlength=5000)
    mx = np.zeros(5000, dtype=np.int64); np.maximum.at(mx, ids[keep], w[keep])
    name = lambda k: f"https://site{k}.example/{k * 7919 % 1000}"
    want = sorted(((s[k] / cnt[k], name(k), int(cnt[k]), int(mx[k])) for k in range(1, 5000) if cnt[k] > 3), key=lambda t: (-t[0], t[1]))
    # each rank keeps its top 25; the global top 25 are among them
    got = sorted(((r["l"], r["k"], r["c"], r["m"]) for r in rows), key=lambda t: (-t[0], t[1]))[:25]
    assert len(got) == 25
    for g, e in zip(got, want[:25]):
        assert g[1:] == e[1:] and abs(g[0] - e[0]) <= FLOAT_RTOL * abs(e[0])


def test_both_workloads_over_the_rccl_backend_at_world_one(ctx):
    """the same plans with torch.distributed's nccl backend (RCCL) instead of gloo: every collective of the exchanges (metadata all-gather, all-to-all(v) of
    fixed-width, Utf8 and validity lanes, the gather to rank 0) runs on device buffers; one rank, so the result is the direct evaluation"""
    import torch
    import torch.distributed as dist
    from dfgpu import dist_workloads as dw, exchange, physical_plan as ops
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(29150 + os.getpid() % 300))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        tc = ops.TaskContext(ctx, 8192)
        for native in (False, True):
            tt = dw.q5_tensors(0.05, 0, 1)
            plan = dw.q5_plan(dw.q5_tables(ctx, tt, 0), native=native)
            local = list(plan.execute(0, tc))
            gathered = exchange.gather_batches(ctx, None, ops.concat_batches(local[0].schema, local), 0, names=dw.Q5_OUTPUT)
            got = {r["n_name"]: int(r["revenue"].scaleb(4)) for r in gathered.to_arrow().to_pylist()}
            assert got == q5_expected(tt)
        ids, length, w = dw.clickbench_tensors(200000, 3000, 0, 1)
        plan = dw.clickbench_plan(dw.clickbench_batch(ctx, ids, length, w, 3000))
        local = list(plan.execute(0, tc))
        gathered = exchange.gather_batches(ctx, None, ops.concat_batches(local[0].schema, local), 0, names=dw.CLICKBENCH_OUTPUT)
        assert gathered.num_rows == 25
        idn = ids.cpu().numpy(); keep = idn != 0
        cnt = np.bincount(idn[keep], minlength=3000)
        for r in gathered.to_arrow().to_pylist():
            k = int(r["k"].split("site")[1].split(".")[0])
            assert r["c"] == cnt[k]
    finally:
        dist.destroy_process_group()
