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


def _q5_tables(seed=77):
    import decimal
    import pyarrow as pa
    rng = np.random.default_rng(seed)
    dec = lambda v: pa.array([decimal.Decimal(int(x)).scaleb(-2) for x in v], type=pa.decimal128(15, 2))
    nc, no, nl, ns = 600, 3000, 12000, 80
    customer = pa.table({"c_custkey": pa.array(np.arange(nc, dtype=np.int64)), "c_nationkey": pa.array(rng.integers(0, 25, nc).astype(np.int64))})
    orders = pa.table({"o_orderkey": pa.array(np.arange(no, dtype=np.int64) * 4 + 1), "o_custkey": pa.array(rng.integers(0, nc, no).astype(np.int64)),
                       "o_orderdate": pa.array(rng.integers(8766, 9500, no).astype(np.int32)).cast(pa.date32())})
    line = pa.table({"l_orderkey": pa.array(rng.integers(0, no, nl).astype(np.int64) * 4 + 1), "l_suppkey": pa.array(rng.integers(0, ns, nl).astype(np.int64)),
                     "l_extendedprice": dec(rng.integers(90000, 10494951, nl)), "l_discount": dec(rng.integers(0, 11, nl))})
    supplier = pa.table({"s_suppkey": pa.array(np.arange(ns, dtype=np.int64)), "s_nationkey": pa.array(rng.integers(0, 25, ns).astype(np.int64))})
    nation = pa.table({"n_nationkey": pa.array(np.arange(25, dtype=np.int64)), "n_name": pa.array([f"NATION{i:02d}" for i in range(25)]), "n_regionkey": pa.array((np.arange(25) % 5).astype(np.int64))})
    region = pa.table({"r_regionkey": pa.array(np.arange(5, dtype=np.int64)), "r_name": pa.array(["AFRICA", "AMERICA", "ASIA", "EUROPE", "MIDDLE EAST"])})
    return dict(customer=customer, orders=orders, line=line, supplier=supplier, nation=nation, region=region)


def _q5_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    try:
        import decimal
        import faulthandler
        faulthandler.dump_traceback_later(140, exit=True)
        import pyarrow as pa
        import torch
        import torch.distributed as dist
        import dfgpu
        from dfgpu import exchange, physical_plan as ops
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        ctx = dfgpu.Context(0, stream=torch.cuda.current_stream().cuda_stream)
        tc = ops.TaskContext(ctx, 8192)
        t = _q5_tables()
        shard = lambda tab: tab.slice(tab.num_rows * rank // world, tab.num_rows * (rank + 1) // world - tab.num_rows * rank // world)      # round-robin file groups
        def mem(tab):
            b = ops.batch_from_arrow(ctx, shard(tab))
            return ops.MemoryExec([[b]], b.schema)
        C, L, B = ops.Column, ops.Literal, ops.BinaryExpr
        rep = lambda plan, keys: ops.CoalesceBatchesExec(exchange.ShuffleExec(plan, keys), 8192)         # RepartitionExec Hash(keys, world) across ranks
        hj = lambda l, r, on: ops.CoalesceBatchesExec(ops.HashJoinExec(l, r, on, None, "Inner", "Partitioned"), 8192)
        fo = ops.FilterExec(B(B(C("o_orderdate", 2), ">=", L(8766, pa.date32())), "AND", B(C("o_orderdate", 2), "<", L(9131, pa.date32()))), mem(t["orders"]))
        j1 = hj(rep(mem(t["customer"]), [C("c_custkey", 0)]), rep(fo, [C("o_custkey", 1)]), [(C("c_custkey", 0), C("o_custkey", 1))])
        p1 = ops.ProjectionExec([(C("c_nationkey", 1), "c_nationkey"), (C("o_orderkey", 2), "o_orderkey")], j1)
        j2 = hj(rep(p1, [C("o_orderkey", 1)]), rep(mem(t["line"]), [C("l_orderkey", 0)]), [(C("o_orderkey", 1), C("l_orderkey", 0))])
        p2 = ops.ProjectionExec([(C("c_nationkey", 0), "c_nationkey"), (C("l_suppkey", 3), "l_suppkey"), (C("l_extendedprice", 4), "l_extendedprice"), (C("l_discount", 5), "l_discount")], j2)
        j3 = hj(rep(mem(t["supplier"]), [C("s_suppkey", 0), C("s_nationkey", 1)]), rep(p2, [C("l_suppkey", 1), C("c_nationkey", 0)]),
                [(C("s_suppkey", 0), C("l_suppkey", 1)), (C("s_nationkey", 1), C("c_nationkey", 0))])                      # TWO-key partitioned join
        p3 = ops.ProjectionExec([(C("s_nationkey", 1), "s_nationkey"), (C("l_extendedprice", 4), "l_extendedprice"), (C("l_discount", 5), "l_discount")], j3)
        fr = ops.FilterExec(B(C("r_name", 1), "=", L("ASIA", pa.utf8())), mem(t["region"]))
        jn = hj(rep(ops.ProjectionExec([(C("r_regionkey", 0), "r_regionkey")], fr), [C("r_regionkey", 0)]), rep(mem(t["nation"]), [C("n_regionkey", 2)]), [(C("r_regionkey", 0), C("n_regionkey", 2))])
        pn = ops.ProjectionExec([(C("n_nationkey", 1), "n_nationkey"), (C("n_name", 2), "n_name")], jn)          # n_name: Utf8 through the exchange
        j4 = hj(rep(pn, [C("n_nationkey", 0)]), rep(p3, [C("s_nationkey", 0)]), [(C("n_nationkey", 0), C("s_nationkey", 0))])
        rev = B(C("l_extendedprice", 3), "*", B(L(decimal.Decimal(1), pa.decimal128(20, 0)), "-", C("l_discount", 4)))
        aggs = [ops.AggregateFunctionExpr("SUM", rev, "revenue", input_field=ops.Field("r", dfgpu.capi.DECIMAL128, 38, 4))]
        partial = ops.AggregateExec("Partial", [(C("n_name", 1), "n_name")], aggs, j4)
        final = ops.AggregateExec("FinalPartitioned", [(C("n_name", 0), "n_name")], aggs, rep(partial, [C("n_name", 0)]))       # Utf8 group key + nullable Decimal128 state shuffled
        local = [b for b in final.execute(0, tc)]
        mine = ops.concat_batches(local[0].schema, local) if local else None
        g = exchange.gather_batches(ctx, None, mine, 0, names=["n_name", "revenue"])
        rows = sorted(zip(*[c.to_arrow().to_pylist() for c in g.columns])) if rank == 0 and g.num_rows else []
        q.put((rank, rows))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # noqa
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("world", [2, 3])
def test_q5_partitioned_joins_over_all_to_all(world):
    """BASELINE config 4's shape (tpch/q5.slt.part): every join input and the partial aggregate hash-repartitioned across ranks
    (ShuffleExec = device hash partition + all-to-all), one join on TWO keys, a Utf8 column and a Utf8 group key through the exchange;
    the gathered result must equal the single-process oracle."""
    import decimal
    import pyarrow as pa
    import pyarrow.compute as pc
    import torch.multiprocessing as mp
    from oracle import pyoracle as po
    from test_gpu_workloads import oracle_agg, oracle_join
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29300 + (os.getpid() % 600) + world
    procs = [ctx.Process(target=_q5_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=160) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for r in range(world):
        assert isinstance(results[r], list), results[r]
    t = _q5_tables()
    okeep = pc.and_(pc.greater_equal(t["orders"]["o_orderdate"].cast(pa.int32()), 8766), pc.less(t["orders"]["o_orderdate"].cast(pa.int32()), 9131))
    x = oracle_join(t["customer"], t["orders"].filter(okeep), ["c_custkey"], ["o_custkey"]).select(["c_nationkey", "o_orderkey"])
    x = oracle_join(x, t["line"], ["o_orderkey"], ["l_orderkey"]).select(["c_nationkey", "l_suppkey", "l_extendedprice", "l_discount"])
    x = oracle_join(t["supplier"], x, ["s_suppkey", "s_nationkey"], ["l_suppkey", "c_nationkey"]).select(["s_nationkey", "l_extendedprice", "l_discount"])
    nt = oracle_join(t["region"].filter(pc.equal(t["region"]["r_name"], "ASIA")).select(["r_regionkey"]), t["nation"], ["r_regionkey"], ["n_regionkey"]).select(["n_nationkey", "n_name"])
    x = oracle_join(nt, x, ["n_nationkey"], ["s_nationkey"])
    r = po.binary("*", x["l_extendedprice"], po.binary("-", pa.array([decimal.Decimal(1)], type=pa.decimal128(20, 0)), x["l_discount"], l_scalar=True))
    want = oracle_agg(x.append_column("rev", r), ["n_name"], [("SUM", "rev")])
    want_rows = sorted(zip(want["c0"].to_pylist(), want["c1"].to_pylist()))
    assert len(want_rows) == 5 and results[0] == want_rows


def _click_rows(seed=91, n=60000, card=900, zipf=True):
    rng = np.random.default_rng(seed)
    ids = (rng.zipf(1.1, n) % card) if zipf else rng.integers(0, card, n)
    words = np.array([""] + [f"https://site{k}.example/{k * 7919 % 1000}" for k in range(1, card)], dtype=object)
    key = words[ids]
    key_null = rng.random(n) < 0.01                                      # NULL keys are one group of their own (primitive.rs:118-122)
    length = rng.integers(0, 500, n).astype(np.int32)
    w = rng.integers(0, 10**6, n).astype(np.int64)
    return key, key_null, length, w


def _click_worker(rank, world, port, q, zipf):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    try:
        import faulthandler
        faulthandler.dump_traceback_later(140, exit=True)
        import pyarrow as pa
        import torch
        import torch.distributed as dist
        import dfgpu
        from dfgpu import capi, exchange, physical_plan as ops
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        ctx = dfgpu.Context(0, stream=torch.cuda.current_stream().cuda_stream)
        tc = ops.TaskContext(ctx, 8192)
        key, key_null, length, w = _click_rows(zipf=zipf)
        n = len(key)
        lo, hi = n * rank // world, n * (rank + 1) // world
        # every rank (≙ every parquet file group) carries its OWN dictionary: the distinct values of its rows in a rank-specific order,
        # so equal codes on two ranks mean different strings and the exchange has to move values, not codes
        keys = pa.array(list(key[lo:hi]), type=pa.utf8(), mask=key_null[lo:hi])
        enc = keys.dictionary_encode()
        perm = np.random.default_rng(rank + 5).permutation(len(enc.dictionary))
        inv = np.empty_like(perm); inv[perm] = np.arange(len(perm))
        codes = pa.array(inv[enc.indices.fill_null(0).to_numpy()].astype(np.int32), mask=key_null[lo:hi])
        dkeys = pa.DictionaryArray.from_arrays(codes, enc.dictionary.take(pa.array(perm)))
        tab = pa.table({"key": dkeys, "len": pa.array(length[lo:hi]), "w": pa.array(w[lo:hi])})
        halves = [ops.batch_from_arrow(ctx, tab.slice(0, tab.num_rows // 2)), ops.batch_from_arrow(ctx, tab.slice(tab.num_rows // 2))]
        src = ops.MemoryExec([halves], halves[0].schema)
        C, L, B, F = ops.Column, ops.Literal, ops.BinaryExpr, ops.Field
        f = ops.CoalesceBatchesExec(ops.FilterExec(B(C("key", 0), "!=", L("", pa.utf8())), src), 8192)            # NULL <> '' is NULL: those rows go too
        proj = ops.ProjectionExec([(C("key", 0), "key"), (ops.CastExpr(C("len", 1), capi.FLOAT64), "lenf"), (C("w", 2), "w")], f)
        aggs = [ops.AggregateFunctionExpr("AVG", C("lenf", 1), "l", input_field=F("x", capi.FLOAT64)), ops.AggregateFunctionExpr("COUNT", None, "c"),
                ops.AggregateFunctionExpr("MAX", C("w", 2), "m", input_field=F("x", capi.INT64))]
        partial = ops.AggregateExec("Partial", [(C("key", 0), "k")], aggs, proj)
        shuffled = ops.CoalesceBatchesExec(exchange.ShuffleExec(partial, [C("k", 0)]), 8192)                       # Utf8 group key + AVG (count, sum) + COUNT + MAX states
        final = ops.AggregateExec("FinalPartitioned", [(C("k", 0), "k")], aggs, shuffled)
        having = ops.FilterExec(B(C("c", 2), ">", L(3, pa.int64())), final)
        order = [ops.PhysicalSortExpr(C("l", 1), True, True), ops.PhysicalSortExpr(C("k", 0), False, False)]
        local = [b for b in ops.SortExec(order, having, fetch=25).execute(0, tc)]                                 # per-rank TopK, merged on rank 0
        mine = ops.concat_batches(local[0].schema, local) if local else None
        g = exchange.gather_batches(ctx, None, mine, 0, names=["k", "l", "c", "m"])
        rows = []
        if rank == 0 and g is not None and g.num_rows:
            merged = [b for b in ops.SortExec(order, ops.MemoryExec([[g]], g.schema), fetch=25).execute(0, tc)]
            rows = list(zip(*[c.to_arrow().to_pylist() for c in ops.concat_batches(merged[0].schema, merged).columns]))
        q.put((rank, rows))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # noqa
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("world,zipf", [(2, True), (3, False)])
def test_clickbench_group_by_dictionary_key_across_ranks(world, zipf):
    """BASELINE config 5's shape (ClickBench Q28, queries.sql:29): filter -> GROUP BY a dictionary-encoded Utf8 key -> AVG/COUNT/MAX ->
    HAVING -> ORDER BY .. LIMIT 25, as Partial -> hash shuffle on the group key -> FinalPartitioned -> per-rank TopK -> merge.  Each rank's
    rows use a different dictionary, so the group key crosses the exchange by value; Zipf keys put one heavy group on one rank."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 28100 + (os.getpid() % 700) + world
    procs = [ctx.Process(target=_click_worker, args=(r, world, port, q, zipf)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=160) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for r in range(world):
        assert isinstance(results[r], list), results[r]
    key, key_null, length, w = _click_rows(zipf=zipf)
    groups = {}
    for k, isnull, ln, wt in zip(key, key_null, length, w):
        if isnull or k == "":
            continue
        g = groups.setdefault(k, [0, 0.0, -1])
        g[0] += 1; g[1] += float(ln); g[2] = max(g[2], int(wt))
    want = sorted(((k, s / c, c, m) for k, (c, s, m) in groups.items() if c > 3), key=lambda r: (-r[1], r[0]))[:25]
    got = results[0]
    assert len(got) == len(want) == 25
    for a, b in zip(got, want):
        assert a[0] == b[0] and a[2] == b[2] and a[3] == b[3] and abs(a[1] - b[1]) <= 1e-9 * abs(b[1]), (a, b)
