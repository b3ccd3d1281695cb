#!/usr/bin/env python3
"""bench_workloads.py -- the other target plan shapes of SURVEY.md section 8(d), measured like bench.py measures Q3: synthetic
TPC-H-shaped columns resident in HBM, one step = one pass of the reference's physical plan through the dfgpu operator layer,
rows/s = input rows / wall time; per-kernel device time from one fully bracketed step.  One JSON line per workload.

  q1_decimal   tpch/q1.slt.part: Filter l_shipdate <= d -> Projection (disc_price once) -> Aggregate (2 dictionary keys, 4 groups)
               4 SUM + 3 AVG + COUNT(*) over Decimal128(15,2) money -> Sort          [algorithmic bytes 70 B/row]
  q1_float64   the same with Float64 money columns ("fp64 accumulators")                  [38 B/row]
  q18_groups   the Q18 subquery: GROUP BY l_orderkey (one Int64 key, 1/4 rows distinct) SUM(l_quantity) -> Filter SUM > 300
               (GroupValuesPrimitive at 150 M groups for SF100)                            [24 B/row]
  hash_join    HashJoinExec on sparse random Int64 keys (no rank index, no bitmap): 150 K x sf build rows, 1.5 M x sf probe rows, 20 % match,
               SUM + COUNT over the join output  [8 B/build row + 16 B/probe row]
  groupby_int64  GROUP BY an unclustered Int64 key (hash table path), 1 M and 20 M groups over 1 M x sf rows, SUM + COUNT, top 10  [16 B/row]
  sort         SortExec over 1 M x sf rows: ORDER BY l_extendedprice DESC, l_shipdate, three columns materialised in the new order
               [28 B/row in + 28 B/row out]
  partition    RepartitionExec Hash([l_orderkey], 8) over the four Q3 lineitem columns (Int64, 2 x Decimal128, Date32), every partition's
               columns materialised: the per-rank work of one shuffle of the 8-GPU plan  [44 B/row in + 44 B/row out]
  clickbench   ClickBench Q28 shape: filter key <> '' -> GROUP BY a dictionary-encoded Utf8 key -> AVG(Int32 as f64), COUNT(*),
               MAX(Int64) -> HAVING -> ORDER BY avg DESC LIMIT 25; uniform and Zipf(1.1) keys  [code 4 + 4 + 8 B/row]
This is NOT the driver's bench contract (bench.py is); it exists so that DESIGN.md can quote measured numbers for these shapes.
"""
import argparse
import decimal
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0


def dec_tensor(torch, n, lo, hi, g):
    v = torch.zeros((n, 2), dtype=torch.int64, device="cuda")
    v[:, 0] = torch.randint(lo, hi, (n,), generator=g, device="cuda", dtype=torch.int64)
    return v


def wrap_dict(ctx, capi, keys, dictionary, key_type):
    import ctypes as C
    d = capi.ArrayDesc()
    dd = dictionary.describe()
    d.type, d.key_type, d.length, d.null_count = capi.DICTIONARY, key_type, keys.numel(), 0
    d.values = keys.data_ptr()
    d.dictionary = C.pointer(dd)
    return ctx.wrap_device(d, keepalive=(keys, dictionary, dd))


KEEP = [False]             # True only while the last timed step runs: that step's output is what the result check reads
STEP_MS = [None]           # wall time of every timed step of the most recent workload (reported next to the mean)
SYNC_CAUSES = [None]       # the last measured workload's host read-backs by cause (from the bracketed step)
LAST_OUT = [None]          # batches of the most recent plan step: what the result checks read (outside the timed region)


def result_columns(batches):
    """result batches -> one list of numpy columns: Decimal128 as Python ints (unscaled), dictionaries / strings as Python strings, the rest as numpy arrays"""
    import numpy as np
    import pyarrow as pa
    if not batches:
        return []
    cols = []
    for i in range(batches[0].num_columns):
        a = pa.concat_arrays([b.columns[i].to_arrow() for b in batches]) if len(batches) > 1 else batches[0].columns[i].to_arrow()
        if pa.types.is_dictionary(a.type):
            a = a.cast(a.type.value_type)
        if pa.types.is_decimal(a.type):
            raw = np.frombuffer(a.buffers()[1], dtype=np.uint64, count=2 * len(a), offset=a.offset * 16).reshape(-1, 2)
            cols.append(([int(lo) | (int(hi) << 64) for lo, hi in raw.tolist()], a.type.scale))
        elif pa.types.is_string(a.type) or pa.types.is_large_string(a.type):
            cols.append(a.to_pylist())
        elif pa.types.is_date32(a.type):
            cols.append(np.asarray(a.cast(pa.int32())))
        else:
            cols.append(np.asarray(a))
    return cols


M64 = (1 << 64) - 1


class _DevView:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False), "version": 2}


def dev_tensor(torch, arr, typestr, cols=1):
    """a dfgpu Array's values buffer as a torch tensor over the same HBM (no copy; `arr` must stay alive while the tensor is used)"""
    d = arr.describe()
    return torch.as_tensor(_DevView(d.values, (d.length,) if cols == 1 else (d.length, cols), typestr), device="cuda")


def wrap64(x):
    """Python int (possibly negative / wider than 64 bits) -> its low 64 bits, what a wrapping Int64 torch sum holds"""
    return int(x) & M64


def check(name, ok, detail):
    """every workload's result of the last timed step against a plain-torch recomputation over the same tensors: asserted, and reported in the JSON line"""
    assert ok, f"{name}: result differs from the torch recomputation: {detail}"
    return {"ok": True, "what": detail}


def time_plan(ctx, ops, tc, template, steps, warmup):
    def step():
        out = [b for b in ops.with_fresh_state(template).execute(0, tc)]
        ctx.synchronize()
        LAST_OUT[0] = out if KEEP[0] else None
        return sum(b.num_rows for b in out)
    breakdown = None
    gc.collect(); gc.disable()          # no cyclic-GC pass between here and the end of the timed loop (one landed in a single partition step: 82 ms against 17)
    for w in range(max(warmup, 1)):
        if w == max(warmup, 1) - 1:
            ctx.profile_select(None); ctx.profile_enable(True); ctx.profile_read()
            rows = step()
            breakdown = ctx.profile_read(); ctx.profile_enable(False)
        else:
            step()
    ctx.synchronize()
    t0 = time.perf_counter(); per = []
    for it in range(steps):
        KEEP[0] = it == steps - 1
        t1 = time.perf_counter(); rows = step(); per.append(round((time.perf_counter() - t1) * 1e3, 3))
    KEEP[0] = False
    dt = (time.perf_counter() - t0) / steps
    gc.enable()
    STEP_MS[0] = per
    kern = {k: round(v[1], 3) for k, v in sorted(breakdown.items(), key=lambda kv: -kv[1][1]) if not k.startswith("sync:")}
    syncs = sum(v[0] for k, v in breakdown.items() if k.startswith("sync:"))
    SYNC_CAUSES[0] = {k[5:]: v[0] for k, v in breakdown.items() if k.startswith("sync:")}
    return dt, rows, kern, syncs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sf", type=float, default=100.0)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--only", default="")
    ap.add_argument("--option", action="append", default=[], metavar="KEY=VALUE", help="ctx option set before the run (dfgpu_ctx_set_option), for A/B runs of one workload")
    args = ap.parse_args()
    run(args)


def run(args, ctx=None, emit=True):
    """args: .sf .steps .warmup .only (comma separated workload names, "" = all).  Returns the list of result dicts (bench.py nests them
    in its JSON line); emit = also print one JSON line per workload."""
    import numpy as np
    import pyarrow as pa
    import torch
    import dfgpu
    from dfgpu import capi, physical_plan as ops
    if ctx is None:
        torch.cuda.set_device(0)
        ctx = dfgpu.Context(0, stream=torch.cuda.current_stream().cuda_stream)
        for kv in getattr(args, "option", []) or []:
            k, v = kv.split("=", 1); ctx.set_option(k, int(v))
    tc = ops.TaskContext(ctx, batch_size=8192)
    C, L, B, F = ops.Column, ops.Literal, ops.BinaryExpr, ops.Field
    g = torch.Generator(device="cuda"); g.manual_seed(20241024)
    n_orders = int(1_500_000 * args.sf)
    want = set(args.only.split(",")) if args.only else None
    lines = torch.randint(1, 8, (n_orders,), generator=g, device="cuda", dtype=torch.int64)
    i = torch.arange(n_orders, dtype=torch.int64, device="cuda")
    l_orderkey = torch.repeat_interleave((i // 8) * 32 + (i % 8) + 1, lines)
    n = l_orderkey.numel()
    del lines, i
    out = []

    def report(name, dt, rows_in, rows_out, bytes_per_row, kern, syncs, extra=None):
        line = {"workload": name, "sf": args.sf, "input_rows": rows_in, "result_rows": rows_out, "ms_per_step": round(dt * 1e3, 3), "rows_per_s": round(rows_in / dt, 1),
                "algorithmic_GBps": round(bytes_per_row * rows_in / dt / 1e9, 1), "frac_of_hbm_peak": round(bytes_per_row * rows_in / dt / 1e9 / HBM_PEAK_GBS, 4),
                "algorithmic_bytes_per_row": bytes_per_row, "kernel_ms_per_step": kern, "host_syncs_per_step": syncs}
        if extra:
            line.update(extra)
        if SYNC_CAUSES[0] is not None:
            line["host_syncs_by_cause"] = SYNC_CAUSES[0]; SYNC_CAUSES[0] = None
        if STEP_MS[0]:
            line["step_ms"] = STEP_MS[0]; line["ms_per_step_median"] = sorted(STEP_MS[0])[len(STEP_MS[0]) // 2]; STEP_MS[0] = None       # ms_per_step is the mean over the timed loop; one slow step of three shows here
        KEEP[0] = False
        live0 = ctx.get_option("live_bytes"); gc.collect()            # device arrays held only by reference cycles of the Python wrappers go back here, not in the middle of a later step
        line["memory_GB"] = {"dfgpu_live_before_gc": round(live0 / 1e9, 2), "dfgpu_live": round(ctx.get_option("live_bytes") / 1e9, 2), "dfgpu_cached": round(ctx.get_option("cached_bytes") / 1e9, 2), "torch_reserved": round(torch.cuda.memory_reserved() / 1e9, 2)}
        ctx.set_option("trim_cache", 1)
        if kern:
            dk = next(iter(kern))
            line["roofline"] = {"bound": "hbm", "kernel": dk, "kernel_ms_per_step": kern[dk], "frac": line["frac_of_hbm_peak"], "unit": "GB/s", "achieved": line["algorithmic_GBps"], "peak": HBM_PEAK_GBS,
                                "note": "achieved = the workload's algorithmic bytes / step time (the plan's whole pass, not one kernel); kernel = largest device time of the bracketed step"}
        if emit:
            print(json.dumps(line), flush=True)
        out.append(line)

    # ------------------------------------------------------------------ Q1
    for money in ("decimal", "float64"):
        name = "q1_" + money
        if want and name not in want:
            continue
        if money == "decimal":
            cols = {k: dec_tensor(torch, n, lo, hi, g) for k, lo, hi in (("l_quantity", 100, 5001), ("l_extendedprice", 90000, 10494951), ("l_discount", 0, 11), ("l_tax", 0, 9))}
            wrap = lambda t: ctx.wrap_tensor(t, capi.DECIMAL128, 15, 2)
            one = L(decimal.Decimal(1), pa.decimal128(20, 0))
            m = lambda p, s: F("x", capi.DECIMAL128, p, s)
            bpr = 16 * 4 + 1 + 1 + 4
        else:
            cols = {k: torch.randint(lo, hi, (n,), generator=g, device="cuda", dtype=torch.int64).to(torch.float64) / 100.0
                    for k, lo, hi in (("l_quantity", 100, 5001), ("l_extendedprice", 90000, 10494951), ("l_discount", 0, 11), ("l_tax", 0, 9))}
            wrap = lambda t: ctx.wrap_tensor(t, capi.FLOAT64)
            one = L(1.0, pa.float64())
            m = lambda p, s: F("x", capi.FLOAT64)
            bpr = 8 * 4 + 1 + 1 + 4
        rf = torch.randint(0, 3, (n,), generator=g, device="cuda", dtype=torch.int8)
        ls = torch.randint(0, 2, (n,), generator=g, device="cuda", dtype=torch.int8)
        shipdate = torch.randint(8035, 10560, (n,), generator=g, device="cuda", dtype=torch.int32)
        torch.cuda.synchronize()
        rfd, lsd = ctx.from_arrow(pa.array(["A", "N", "R"])), ctx.from_arrow(pa.array(["F", "O"]))
        names = ["l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_returnflag", "l_linestatus", "l_shipdate"]
        arrays = [wrap(cols[k]) for k in names[:4]] + [wrap_dict(ctx, capi, rf, rfd, capi.INT8), wrap_dict(ctx, capi, ls, lsd, capi.INT8), ctx.wrap_tensor(shipdate, capi.DATE32)]
        batch = ops.RecordBatch.from_arrays(ctx, names, arrays)
        src = ops.MemoryExec([[batch]], batch.schema)
        f = ops.CoalesceBatchesExec(ops.FilterExec(B(C("l_shipdate", 6), "<=", L(10471, pa.date32())), src), 8192)
        proj = ops.ProjectionExec([(B(C("l_extendedprice", 1), "*", B(one, "-", C("l_discount", 2))), "disc_price"), (C("l_quantity", 0), "l_quantity"), (C("l_extendedprice", 1), "l_extendedprice"),
                                   (C("l_discount", 2), "l_discount"), (C("l_tax", 3), "l_tax"), (C("l_returnflag", 4), "l_returnflag"), (C("l_linestatus", 5), "l_linestatus")], f)
        charge = B(C("disc_price", 0), "*", B(one, "+", C("l_tax", 4)))
        aggs = [ops.AggregateFunctionExpr("SUM", C("l_quantity", 1), "sum_qty", input_field=m(15, 2)), ops.AggregateFunctionExpr("SUM", C("l_extendedprice", 2), "sum_base_price", input_field=m(15, 2)),
                ops.AggregateFunctionExpr("SUM", C("disc_price", 0), "sum_disc_price", input_field=m(38, 4)), ops.AggregateFunctionExpr("SUM", charge, "sum_charge", input_field=m(38, 6)),
                ops.AggregateFunctionExpr("AVG", C("l_quantity", 1), "avg_qty", input_field=m(15, 2)), ops.AggregateFunctionExpr("AVG", C("l_extendedprice", 2), "avg_price", input_field=m(15, 2)),
                ops.AggregateFunctionExpr("AVG", C("l_discount", 3), "avg_disc", input_field=m(15, 2)), ops.AggregateFunctionExpr("COUNT", None, "count_order")]
        agg = ops.AggregateExec("Single", [(C("l_returnflag", 5), "l_returnflag"), (C("l_linestatus", 6), "l_linestatus")], aggs, proj)
        plan = ops.SortExec([ops.PhysicalSortExpr(C("l_returnflag", 0), False, False), ops.PhysicalSortExpr(C("l_linestatus", 1), False, False)], agg)
        dt, rows, kern, syncs = time_plan(ctx, ops, tc, plan, args.steps, args.warmup)
        # result check: 6 groups (l_returnflag, l_linestatus) in key order; sums / counts against torch index_add over the same tensors
        res = result_columns(LAST_OUT[0]); LAST_OUT[0] = None
        sel = shipdate <= 10471
        gid = (rf.to(torch.int64) * 2 + ls.to(torch.int64))[sel]
        gmask = [gid == k for k in range(6)]               # six groups: masked sums (an index_add of 6e8 rows into six words serialises on its atomics)
        cnt = torch.stack([m.sum() for m in gmask])
        got_keys = list(zip(res[0], res[1])); exp_keys = [(a, b) for a in ("A", "N", "R") for b in ("F", "O")]
        ok = got_keys == exp_keys and [int(x) for x in res[9]] == cnt.tolist()
        if money == "decimal":
            q, pr, di, tx = (cols[k][:, 0][sel] for k in ("l_quantity", "l_extendedprice", "l_discount", "l_tax"))
            acc = lambda v: [int(v[m].sum().item()) for m in gmask]
            dp = pr * (100 - di)
            expect = [acc(q), acc(pr), acc(dp), acc(dp * (100 + tx))]
            for ci, w in zip((2, 3, 4, 5), expect):
                ok = ok and [wrap64(v) for v in res[ci][0]] == [wrap64(v) for v in w]
            for ci, w in zip((6, 7, 8), (expect[0], expect[1], acc(di))):            # AVG = SUM / COUNT, Decimal128 result: compared as a ratio to 1e-9
                sc = 10 ** res[ci][1]
                ok = ok and all(abs(g / sc - (wv / 100) / c) <= 1e-9 * abs((wv / 100) / c) + 10 / sc for g, wv, c in zip(res[ci][0], w, cnt.tolist()))
            what = "6 groups in key order: COUNT(*), wrapping Int64 (low word of Decimal128) SUM(qty), SUM(price), SUM(disc_price), SUM(charge) exact; 3 AVGs to 1e-9"
        else:
            q, pr, di, tx = (cols[k][sel] for k in ("l_quantity", "l_extendedprice", "l_discount", "l_tax"))
            acc = lambda v: [float(v[m].sum().item()) for m in gmask]
            dp = pr * (1.0 - di)
            expect = [acc(q), acc(pr), acc(dp), acc(dp * (1.0 + tx))]
            close = lambda g, w: abs(g - w) <= 1e-9 * abs(w)
            for ci, w in zip((2, 3, 4, 5), expect):
                ok = ok and all(close(float(g), wv) for g, wv in zip(res[ci], w))
            for ci, w in zip((6, 7, 8), (expect[0], expect[1], acc(di))):
                ok = ok and all(close(float(g), wv / c) for g, wv, c in zip(res[ci], w, cnt.tolist()))
            what = "6 groups in key order: COUNT(*) exact; Float64 SUMs and AVGs within 1e-9 relative of torch index_add"
        del sel, gid, gmask, q, pr, di, tx, dp
        report(name, dt, n, rows, bpr, kern, syncs, {"result_check": check(name, ok, what)})
        del cols, rf, ls, shipdate, arrays, batch, src, plan
        torch.cuda.empty_cache()

    # ------------------------------------------------------------------ Q18 subquery: 1 Int64 key, n/4 groups
    if not want or "q18_groups" in want:
        qty = dec_tensor(torch, n, 100, 5001, g)
        torch.cuda.synchronize()
        batch = ops.RecordBatch.from_arrays(ctx, ["l_orderkey", "l_quantity"], [ctx.wrap_tensor(l_orderkey, capi.INT64), ctx.wrap_tensor(qty, capi.DECIMAL128, 15, 2)])
        src = ops.MemoryExec([[batch]], batch.schema)
        sub = ops.AggregateExec("Single", [(C("l_orderkey", 0), "l_orderkey")], [ops.AggregateFunctionExpr("SUM", C("l_quantity", 1), "SUM(l_quantity)", input_field=F("q", capi.DECIMAL128, 15, 2))], src)
        plan = ops.CoalesceBatchesExec(ops.FilterExec(B(C("SUM(l_quantity)", 1), ">", L(decimal.Decimal(300), pa.decimal128(25, 2))), sub), 8192)
        dt, rows, kern, syncs = time_plan(ctx, ops, tc, plan, args.steps, args.warmup)
        res = result_columns(LAST_OUT[0]); LAST_OUT[0] = None
        kk = l_orderkey - 1; oidx = (kk // 32) * 8 + (kk % 32); del kk
        sums = torch.zeros(n_orders, dtype=torch.int64, device="cuda").index_add_(0, oidx, qty[:, 0]); big = sums > 30000
        ok = rows == int(big.sum().item()) and wrap64(sum(int(x) for x in res[0])) == wrap64(int(((torch.nonzero(big).flatten() // 8) * 32 + torch.nonzero(big).flatten() % 8 + 1).sum().item())) and \
            wrap64(sum(res[1][0])) == wrap64(int(sums[big].sum().item()))
        del oidx, sums, big
        report("q18_groups", dt, n, rows, 8 + 16, kern, syncs, {"groups": n_orders, "result_check": check("q18_groups", ok, "rows with SUM(l_quantity) > 300, wrapping sums of their keys and of their sums == torch index_add per order")})
        del qty, batch, src, plan
        torch.cuda.empty_cache()

    # ------------------------------------------------------------------ Q5: 6 tables, 5 hash joins (one on two keys), SUM by n_name (Utf8)
    if not want or "q5" in want:
        n_cust, n_supp = int(150_000 * args.sf), max(10, int(10_000 * args.sf))
        ii = torch.arange(n_orders, dtype=torch.int64, device="cuda")
        o_orderkey = (ii // 8) * 32 + (ii % 8) + 1
        del ii
        o_custkey = torch.randint(1, n_cust + 1, (n_orders,), generator=g, device="cuda", dtype=torch.int64)
        o_orderdate = torch.randint(8035, 10441, (n_orders,), generator=g, device="cuda", dtype=torch.int32)
        c_custkey = torch.arange(1, n_cust + 1, dtype=torch.int64, device="cuda")
        c_nationkey = torch.randint(0, 25, (n_cust,), generator=g, device="cuda", dtype=torch.int64)
        s_suppkey = torch.arange(1, n_supp + 1, dtype=torch.int64, device="cuda")
        s_nationkey = torch.randint(0, 25, (n_supp,), generator=g, device="cuda", dtype=torch.int64)
        l_suppkey = torch.randint(1, n_supp + 1, (n,), generator=g, device="cuda", dtype=torch.int64)
        price, disc = dec_tensor(torch, n, 90000, 10494951, g), dec_tensor(torch, n, 0, 11, g)
        torch.cuda.synchronize()
        W = lambda t, ty=capi.INT64: ctx.wrap_tensor(t, ty)
        mk = lambda names, arrays: (lambda b: ops.MemoryExec([[b]], b.schema))(ops.RecordBatch.from_arrays(ctx, names, arrays))
        customer = mk(["c_custkey", "c_nationkey"], [W(c_custkey), W(c_nationkey)])
        orders = mk(["o_orderkey", "o_custkey", "o_orderdate"], [W(o_orderkey), W(o_custkey), W(o_orderdate, capi.DATE32)])
        line = mk(["l_orderkey", "l_suppkey", "l_extendedprice", "l_discount"], [W(l_orderkey), W(l_suppkey), ctx.wrap_tensor(price, capi.DECIMAL128, 15, 2), ctx.wrap_tensor(disc, capi.DECIMAL128, 15, 2)])
        supplier = mk(["s_suppkey", "s_nationkey"], [W(s_suppkey), W(s_nationkey)])
        nat = pa.table({"n_nationkey": pa.array(list(range(25)), type=pa.int64()), "n_name": pa.array([f"NATION{i:02d}" for i in range(25)]), "n_regionkey": pa.array([i % 5 for i in range(25)], type=pa.int64())})
        reg = pa.table({"r_regionkey": pa.array(list(range(5)), type=pa.int64()), "r_name": pa.array(["AFRICA", "AMERICA", "ASIA", "EUROPE", "MIDDLE EAST"])})
        nation = (lambda b: ops.MemoryExec([[b]], b.schema))(ops.batch_from_arrow(ctx, nat))
        region = (lambda b: ops.MemoryExec([[b]], b.schema))(ops.batch_from_arrow(ctx, reg))
        cb = lambda p: ops.CoalesceBatchesExec(p, 8192)
        hj = lambda l, r, on: cb(ops.HashJoinExec(l, r, on, None, "Inner", "Partitioned"))
        fo = cb(ops.FilterExec(B(B(C("o_orderdate", 2), ">=", L(8766, pa.date32())), "AND", B(C("o_orderdate", 2), "<", L(9131, pa.date32()))), orders))
        j1 = hj(customer, fo, [(C("c_custkey", 0), C("o_custkey", 1))])
        p1 = ops.ProjectionExec([(C("c_nationkey", 1), "c_nationkey"), (C("o_orderkey", 2), "o_orderkey")], j1)
        j2 = hj(p1, line, [(C("o_orderkey", 1), C("l_orderkey", 0))])
        p2 = ops.ProjectionExec([(C("c_nationkey", 0), "c_nationkey"), (C("l_suppkey", 3), "l_suppkey"), (C("l_extendedprice", 4), "l_extendedprice"), (C("l_discount", 5), "l_discount")], j2)
        j3 = hj(supplier, p2, [(C("s_suppkey", 0), C("l_suppkey", 1)), (C("s_nationkey", 1), C("c_nationkey", 0))])
        p3 = ops.ProjectionExec([(C("s_nationkey", 1), "s_nationkey"), (C("l_extendedprice", 4), "l_extendedprice"), (C("l_discount", 5), "l_discount")], j3)
        fr = cb(ops.FilterExec(B(C("r_name", 1), "=", L("ASIA", pa.utf8())), region))
        jn = hj(ops.ProjectionExec([(C("r_regionkey", 0), "r_regionkey")], fr), nation, [(C("r_regionkey", 0), C("n_regionkey", 2))])
        pn = ops.ProjectionExec([(C("n_nationkey", 1), "n_nationkey"), (C("n_name", 2), "n_name")], jn)
        j4 = hj(pn, p3, [(C("n_nationkey", 0), C("s_nationkey", 0))])
        rev = B(C("l_extendedprice", 3), "*", B(L(decimal.Decimal(1), pa.decimal128(20, 0)), "-", C("l_discount", 4)))
        agg = ops.AggregateExec("Single", [(C("n_name", 1), "n_name")], [ops.AggregateFunctionExpr("SUM", rev, "revenue", input_field=F("r", capi.DECIMAL128, 38, 4))], j4)
        plan = ops.SortExec([ops.PhysicalSortExpr(C("revenue", 1), True, True)], agg)
        rows_in = n + n_orders + n_cust + n_supp + 30
        dt, rows, kern, syncs = time_plan(ctx, ops, tc, plan, args.steps, args.warmup)
        bytes_total = n * (8 + 8 + 16 + 16) + n_orders * (8 + 8 + 4) + n_cust * 16 + n_supp * 16
        # result check: revenue per ASIA nation (n_regionkey = n_nationkey % 5 == 2), descending, against torch gathers over the dense keys
        res = result_columns(LAST_OUT[0]); LAST_OUT[0] = None
        o_ok = (o_orderdate >= 8766) & (o_orderdate < 9131)
        o_nat = c_nationkey[o_custkey - 1]
        kk = l_orderkey - 1; oidx = (kk // 32) * 8 + (kk % 32); del kk
        s_nat = s_nationkey[l_suppkey - 1]
        lsel = o_ok[oidx] & (s_nat == o_nat[oidx]) & (s_nat % 5 == 2)
        rev = torch.zeros(25, dtype=torch.int64, device="cuda").index_add_(0, s_nat[lsel], (price[:, 0] * (100 - disc[:, 0]))[lsel]).tolist()
        expect = sorted(((r, f"NATION{i:02d}") for i, r in enumerate(rev) if i % 5 == 2 and r), key=lambda t: -t[0])
        ok = [(wrap64(v), nm) for v, nm in zip(res[1][0], res[0])] == [(wrap64(r), nm) for r, nm in expect]
        del o_ok, o_nat, oidx, s_nat, lsel
        report("q5", dt, rows_in, rows, round(bytes_total / rows_in, 2), kern, syncs, {"result_check": check("q5", ok, "SUM(l_extendedprice * (1 - l_discount)) per ASIA nation, order included, == torch gathers + index_add (exact, low word of Decimal128)")})
        del customer, orders, line, supplier, plan, price, disc, l_suppkey, o_custkey, o_orderdate, o_orderkey
        torch.cuda.empty_cache()

    # ------------------------------------------------------------------ Q18: sub-aggregate + HAVING -> LeftSemi against customer-orders-lineitem, 5-key group-by
    if not want or "q18" in want:
        n_cust = int(150_000 * args.sf)
        ii = torch.arange(n_orders, dtype=torch.int64, device="cuda")
        o_orderkey = (ii // 8) * 32 + (ii % 8) + 1
        del ii
        o_custkey = torch.randint(1, n_cust + 1, (n_orders,), generator=g, device="cuda", dtype=torch.int64)
        o_orderdate = torch.randint(8035, 10441, (n_orders,), generator=g, device="cuda", dtype=torch.int32)
        o_totalprice = dec_tensor(torch, n_orders, 10**5, 5 * 10**7, g)
        c_custkey = torch.arange(1, n_cust + 1, dtype=torch.int64, device="cuda")
        qty = dec_tensor(torch, n, 100, 5001, g)
        torch.cuda.synchronize()
        W = lambda t, ty=capi.INT64: ctx.wrap_tensor(t, ty)
        mk = lambda names, arrays: (lambda b: ops.MemoryExec([[b]], b.schema))(ops.RecordBatch.from_arrays(ctx, names, arrays))
        customer = mk(["c_custkey"], [W(c_custkey)])          # c_name is a function of c_custkey in TPC-H ("Customer#%09d"); carried as the key itself
        orders = mk(["o_orderkey", "o_custkey", "o_totalprice", "o_orderdate"], [W(o_orderkey), W(o_custkey), ctx.wrap_tensor(o_totalprice, capi.DECIMAL128, 15, 2), W(o_orderdate, capi.DATE32)])
        line = mk(["l_orderkey", "l_quantity"], [W(l_orderkey), ctx.wrap_tensor(qty, capi.DECIMAL128, 15, 2)])
        qf = F("q", capi.DECIMAL128, 15, 2)
        sub = ops.AggregateExec("Single", [(C("l_orderkey", 0), "l_orderkey")], [ops.AggregateFunctionExpr("SUM", C("l_quantity", 1), "SUM(l_quantity)", input_field=qf)], line)
        having = ops.CoalesceBatchesExec(ops.FilterExec(B(C("SUM(l_quantity)", 1), ">", L(decimal.Decimal(300), pa.decimal128(25, 2))), sub), 8192)
        j1 = ops.HashJoinExec(customer, orders, [(C("c_custkey", 0), C("o_custkey", 1))], None, "Inner", "CollectLeft")          # c_custkey,o_orderkey,o_custkey,o_totalprice,o_orderdate
        j2 = ops.HashJoinExec(j1, line, [(C("o_orderkey", 1), C("l_orderkey", 0))], None, "Inner", "CollectLeft")                 # + l_orderkey,l_quantity
        semi = ops.HashJoinExec(j2, ops.ProjectionExec([(C("l_orderkey", 0), "l_orderkey")], having), [(C("o_orderkey", 1), C("l_orderkey", 0))], None, "LeftSemi", "CollectLeft")
        gby = [(C("c_custkey", 0), "c_custkey"), (C("o_orderkey", 1), "o_orderkey"), (C("o_orderdate", 4), "o_orderdate"), (C("o_totalprice", 3), "o_totalprice")]
        agg = ops.AggregateExec("Single", gby, [ops.AggregateFunctionExpr("SUM", C("l_quantity", 6), "SUM(l_quantity)", input_field=qf)], semi)
        plan = ops.SortExec([ops.PhysicalSortExpr(C("o_totalprice", 3), True, True), ops.PhysicalSortExpr(C("o_orderdate", 2), False, False)], agg)
        rows_in = 2 * n + n_orders + n_cust
        dt, rows, kern, syncs = time_plan(ctx, ops, tc, plan, args.steps, args.warmup)
        bytes_total = 2 * n * (8 + 16) + n_orders * (8 + 8 + 16 + 4) + n_cust * 8
        # result check: one row per order whose lineitems sum to > 300, ordered by o_totalprice DESC, o_orderdate
        res = result_columns(LAST_OUT[0]); LAST_OUT[0] = None
        kk = l_orderkey - 1; oidx = (kk // 32) * 8 + (kk % 32); del kk
        sums = torch.zeros(n_orders, dtype=torch.int64, device="cuda").index_add_(0, oidx, qty[:, 0]); big = torch.nonzero(sums > 30000).flatten()
        tp, od = o_totalprice[:, 0][big], o_orderdate[big].to(torch.int64)
        order = torch.argsort((tp.max() - tp) * 4096 + (od - 8035), stable=True)
        big = big[order]
        ok = rows == big.numel() and np.array_equal(np.asarray(res[1]).astype(np.int64), o_orderkey[big].cpu().numpy()) and np.array_equal(np.asarray(res[0]).astype(np.int64), o_custkey[big].cpu().numpy()) and \
            [wrap64(v) for v in res[4][0]] == [wrap64(v) for v in sums[big].tolist()] and [wrap64(v) for v in res[3][0]] == [wrap64(v) for v in o_totalprice[:, 0][big].tolist()]
        del oidx, sums, big, tp, od, order
        report("q18", dt, rows_in, rows, round(bytes_total / rows_in, 2), kern, syncs, {"result_check": check("q18", ok, "every result row (c_custkey, o_orderkey, o_totalprice, SUM(l_quantity)) in output order == torch index_add + stable argsort")})
        del customer, orders, line, plan, qty, o_totalprice, o_custkey, o_orderdate, o_orderkey
        torch.cuda.empty_cache()

    # ------------------------------------------------------------------ general-case operators: keys that are neither clustered nor dense
    # ------------------------------------------------------------------ unique build keys over a dense domain, in no order (a primary-key side after a hash repartition -- the N > 1 shuffle plan's builds)
    if not want or "hash_join" in want or "hash_join_dense_unsorted" in want:
        nbd, npd = int(150_000 * args.sf), int(1_500_000 * args.sf)
        bkd = torch.randperm(nbd * 4, generator=g, device="cuda")[:nbd].to(torch.int64) + 1
        pkd = torch.randint(1, nbd * 4 + 1, (npd,), generator=g, device="cuda", dtype=torch.int64)
        pvd = torch.randint(0, 10**6, (npd,), generator=g, device="cuda", dtype=torch.int64)
        torch.cuda.synchronize()
        leftd = ops.RecordBatch.from_arrays(ctx, ["k"], [ctx.wrap_tensor(bkd, capi.INT64)])
        rightd = ops.RecordBatch.from_arrays(ctx, ["k", "v"], [ctx.wrap_tensor(pkd, capi.INT64), ctx.wrap_tensor(pvd, capi.INT64)])
        jd = ops.HashJoinExec(ops.MemoryExec([[leftd]], leftd.schema), ops.MemoryExec([[rightd]], rightd.schema), [(C("k", 0), C("k", 0))], None, "Inner", "CollectLeft")
        pland = ops.AggregateExec("Single", [], [ops.AggregateFunctionExpr("SUM", C("v", 2), "s", input_field=F("v", capi.INT64)), ops.AggregateFunctionExpr("COUNT", None, "c")], jd)
        dt, rows, kern, syncs = time_plan(ctx, ops, tc, pland, args.steps, args.warmup)
        res = result_columns(LAST_OUT[0]); LAST_OUT[0] = None
        member = torch.zeros(nbd * 4 + 2, dtype=torch.bool, device="cuda"); member[bkd] = True; hitd = member[pkd]
        ok = int(res[1][0]) == int(hitd.sum().item()) and wrap64(int(res[0][0])) == wrap64(int(pvd[hitd].sum().item()))
        per_first = list(STEP_MS[0])
        ctx.set_option("join_rank_index_unsorted", 0)
        try:
            dt0, _, kern0, _ = time_plan(ctx, ops, tc, pland, args.steps, args.warmup)
            res0 = result_columns(LAST_OUT[0]); LAST_OUT[0] = None
        finally:
            ctx.set_option("join_rank_index_unsorted", 1)
        ok = ok and int(res0[1][0]) == int(res[1][0]) and wrap64(int(res0[0][0])) == wrap64(int(res[0][0]))
        STEP_MS[0] = per_first
        report("hash_join_dense_unsorted", dt, nbd + npd, rows, round((nbd * 8 + npd * 16) / (nbd + npd), 2), kern, syncs, {"build_rows": nbd, "probe_rows": npd, "match_fraction": 0.25,
               "hash_table_ms_per_step": round(dt0 * 1e3, 3), "hash_table_kernel_ms_per_step": kern0,
               "result_check": check("hash_join_dense_unsorted", ok, "COUNT(*) and SUM(v) over the join output == a torch membership table over the same keys, and == the same plan through the bitmap + hash table")})
        del bkd, pkd, pvd, leftd, rightd, jd, pland, member, hitd
        torch.cuda.empty_cache()
    # on request only (--only hash_join_big_build): a build side beyond 2048 x 14 000 rows (40 M x 150 M at sf 100) -- the partitioned join with up to 4096 partitions against the
    # global table it used to decline to (--option join_partitioned_big=0)
    if want and "hash_join_big_build" in want:
        nbb, npb = int(400_000 * args.sf), int(1_500_000 * args.sf)
        bkb = torch.randint(0, 2**62, (nbb,), generator=g, device="cuda", dtype=torch.int64)
        pkb = torch.cat([bkb[torch.randint(0, nbb, (npb // 5,), generator=g, device="cuda")], torch.randint(0, 2**62, (npb - npb // 5,), generator=g, device="cuda", dtype=torch.int64)])
        pkb = pkb[torch.randperm(npb, generator=g, device="cuda")]
        pvb = torch.randint(0, 10**6, (npb,), generator=g, device="cuda", dtype=torch.int64)
        torch.cuda.synchronize()
        leftb = ops.RecordBatch.from_arrays(ctx, ["k"], [ctx.wrap_tensor(bkb, capi.INT64)])
        rightb = ops.RecordBatch.from_arrays(ctx, ["k", "v"], [ctx.wrap_tensor(pkb, capi.INT64), ctx.wrap_tensor(pvb, capi.INT64)])
        jb = ops.HashJoinExec(ops.MemoryExec([[leftb]], leftb.schema), ops.MemoryExec([[rightb]], rightb.schema), [(C("k", 0), C("k", 0))], None, "Inner", "CollectLeft")
        planb = ops.AggregateExec("Single", [], [ops.AggregateFunctionExpr("SUM", C("v", 2), "s", input_field=F("v", capi.INT64)), ops.AggregateFunctionExpr("COUNT", None, "c")], jb)
        dt, rows, kern, syncs = time_plan(ctx, ops, tc, planb, args.steps, args.warmup)
        res = result_columns(LAST_OUT[0]); LAST_OUT[0] = None
        sb = torch.sort(bkb).values; pos = torch.searchsorted(sb, pkb).clamp_(max=nbb - 1); hit = sb[pos] == pkb
        ok = int(res[1][0]) == int(hit.sum().item()) and wrap64(int(res[0][0])) == wrap64(int(pvb[hit].sum().item()))
        del sb, pos, hit
        report("hash_join_big_build", dt, nbb + npb, rows, round((nbb * 8 + npb * 16) / (nbb + npb), 2), kern, syncs, {"build_rows": nbb, "probe_rows": npb, "match_fraction": 0.2,
               "result_check": check("hash_join_big_build", ok, "COUNT(*) and SUM(v) over the join output == torch sort + searchsorted over the same keys")})
        del bkb, pkb, pvb, leftb, rightb, jb, planb
        torch.cuda.empty_cache()
    if not want or "hash_join" in want or "hash_join_plain" in want or "hash_join_fk5" in want or "hash_join_two_keys" in want:          # hash_join = every variant; _plain / _fk5 / _two_keys = one of them (profiling passes)
        do_plain, do_fk5 = (not want or "hash_join" in want or "hash_join_plain" in want), (not want or "hash_join" in want or "hash_join_fk5" in want)
        do_two = not want or "hash_join" in want or "hash_join_two_keys" in want
        nb, npr = int(150_000 * args.sf), int(1_500_000 * args.sf)
        bk = torch.randint(0, 2**62, (nb,), generator=g, device="cuda", dtype=torch.int64)
        pk = torch.cat([bk[torch.randint(0, nb, (npr // 5,), generator=g, device="cuda")], torch.randint(0, 2**62, (npr - npr // 5,), generator=g, device="cuda", dtype=torch.int64)])
        pk = pk[torch.randperm(npr, generator=g, device="cuda")]
        pv = torch.randint(0, 10**6, (npr,), generator=g, device="cuda", dtype=torch.int64)
        torch.cuda.synchronize()
        left = ops.RecordBatch.from_arrays(ctx, ["k"], [ctx.wrap_tensor(bk, capi.INT64)])
        right = ops.RecordBatch.from_arrays(ctx, ["k", "v"], [ctx.wrap_tensor(pk, capi.INT64), ctx.wrap_tensor(pv, capi.INT64)])
        j = ops.HashJoinExec(ops.MemoryExec([[left]], left.schema), ops.MemoryExec([[right]], right.schema), [(C("k", 0), C("k", 0))], None, "Inner", "CollectLeft")
        plan = ops.AggregateExec("Single", [], [ops.AggregateFunctionExpr("SUM", C("v", 2), "s", input_field=F("v", capi.INT64)), ops.AggregateFunctionExpr("COUNT", None, "c")], j)
        if do_plain:
            dt, rows, kern, syncs = time_plan(ctx, ops, tc, plan, args.steps, args.warmup)
            res = result_columns(LAST_OUT[0]); LAST_OUT[0] = None
            sb = torch.sort(bk).values; pos = torch.searchsorted(sb, pk).clamp_(max=nb - 1); hit = sb[pos] == pk         # build keys are distinct (62-bit random): a match is one pair
            ok = int(res[1][0]) == int(hit.sum().item()) and wrap64(int(res[0][0])) == wrap64(int(pv[hit].sum().item()))
            del sb, pos, hit
            report("hash_join_sparse_keys", dt, nb + npr, rows, round((nb * 8 + npr * 16) / (nb + npr), 2), kern, syncs, {"build_rows": nb, "probe_rows": npr, "match_fraction": 0.2,
                   "result_check": check("hash_join_sparse_keys", ok, "COUNT(*) and SUM(v) over the join output == torch sort + searchsorted over the same keys")})
        # the same join with a foreign-key build side: every build key five times (75 M build rows at SF100 would leave the 2048-partition range: 3 M distinct keys x 5)
        if do_fk5:
            nd = nb // 5
            bk5 = bk[:nd].repeat(5)[torch.randperm(nd * 5, generator=g, device="cuda")]
            torch.cuda.synchronize()
            left5 = ops.RecordBatch.from_arrays(ctx, ["k"], [ctx.wrap_tensor(bk5, capi.INT64)])
            j5 = ops.HashJoinExec(ops.MemoryExec([[left5]], left5.schema), ops.MemoryExec([[right]], right.schema), [(C("k", 0), C("k", 0))], None, "Inner", "CollectLeft")
            plan5 = ops.AggregateExec("Single", [], [ops.AggregateFunctionExpr("SUM", C("v", 2), "s", input_field=F("v", capi.INT64)), ops.AggregateFunctionExpr("COUNT", None, "c")], j5)
            dt, rows, kern, syncs = time_plan(ctx, ops, tc, plan5, args.steps, args.warmup)
            res = result_columns(LAST_OUT[0]); LAST_OUT[0] = None
            sb = torch.sort(bk[:nd]).values; pos = torch.searchsorted(sb, pk).clamp_(max=nd - 1); hit = sb[pos] == pk
            ok = int(res[1][0]) == 5 * int(hit.sum().item()) and wrap64(int(res[0][0])) == wrap64(5 * int(pv[hit].sum().item()))
            del sb, pos, hit
            report("hash_join_sparse_keys_fk5", dt, nd * 5 + npr, rows, round((nd * 5 * 8 + npr * 16) / (nd * 5 + npr), 2), kern, syncs, {"build_rows": nd * 5, "probe_rows": npr, "rows_per_build_key": 5,
                   "result_check": check("hash_join_sparse_keys_fk5", ok, "COUNT(*) and SUM(v) over the join output (5 pairs per matching probe row) == torch")})
        # the same join on TWO Int64 key columns whose ranges do not pack into one word: the partitioned join's hashed mode (64-bit keyset hashes through the LDS tables, every
        # emitted pair verified in the columns); next to it the same plan with that mode off = the global open-addressing table it replaces
        if do_two:
            bk1 = torch.randint(0, 2**62, (nb,), generator=g, device="cuda", dtype=torch.int64)
            src = torch.randint(0, nb, (npr,), generator=g, device="cuda")
            hit2 = torch.rand(npr, generator=g, device="cuda") < 0.2
            pk0 = torch.where(hit2, bk[src], torch.randint(0, 2**62, (npr,), generator=g, device="cuda", dtype=torch.int64))
            pk1 = torch.where(torch.rand(npr, generator=g, device="cuda") < 0.95, bk1[src], pk0)          # 5 % of the first-column matches differ in the second column
            hit2 = hit2 & (pk1 == bk1[src])
            del src
            torch.cuda.synchronize()
            left2 = ops.RecordBatch.from_arrays(ctx, ["k", "j"], [ctx.wrap_tensor(bk, capi.INT64), ctx.wrap_tensor(bk1, capi.INT64)])
            right2 = ops.RecordBatch.from_arrays(ctx, ["k", "j", "v"], [ctx.wrap_tensor(pk0, capi.INT64), ctx.wrap_tensor(pk1, capi.INT64), ctx.wrap_tensor(pv, capi.INT64)])
            j2k = ops.HashJoinExec(ops.MemoryExec([[left2]], left2.schema), ops.MemoryExec([[right2]], right2.schema), [(C("k", 0), C("k", 0)), (C("j", 1), C("j", 1))], None, "Inner", "CollectLeft")
            plan2 = ops.AggregateExec("Single", [], [ops.AggregateFunctionExpr("SUM", C("v", 4), "s", input_field=F("v", capi.INT64)), ops.AggregateFunctionExpr("COUNT", None, "c")], j2k)
            dt, rows, kern, syncs = time_plan(ctx, ops, tc, plan2, args.steps, args.warmup)
            res = result_columns(LAST_OUT[0]); LAST_OUT[0] = None
            ok = int(res[1][0]) == int(hit2.sum().item()) and wrap64(int(res[0][0])) == wrap64(int(pv[hit2].sum().item()))
            per_first = list(STEP_MS[0])
            ctx.set_option("join_partitioned_hashed", 0)
            try:
                dt0, _, kern0, _ = time_plan(ctx, ops, tc, plan2, args.steps, args.warmup)
                res0 = result_columns(LAST_OUT[0]); LAST_OUT[0] = None
            finally:
                ctx.set_option("join_partitioned_hashed", 1)
            ok = ok and int(res0[1][0]) == int(res[1][0]) and wrap64(int(res0[0][0])) == wrap64(int(res[0][0]))
            STEP_MS[0] = per_first
            report("hash_join_two_keys", dt, nb + npr, rows, round((nb * 16 + npr * 24) / (nb + npr), 2), kern, syncs, {"build_rows": nb, "probe_rows": npr, "match_fraction": round(float(hit2.float().mean().item()), 3),
                   "global_table_ms_per_step": round(dt0 * 1e3, 3), "global_table_kernel_ms_per_step": kern0,
                   "result_check": check("hash_join_two_keys", ok, "COUNT(*) and SUM(v) over the join output == the matches by construction (torch), and == the same plan through the global table")})
            del bk1, pk0, pk1, hit2, left2, right2, j2k, plan2
        del bk, pk, pv, left, right, plan
        torch.cuda.empty_cache()
    if not want or "groupby_int64" in want or any(w.startswith("groupby_int64_unclustered_") for w in want):
        ng = int(1_000_000 * args.sf)
        for total in (10**6, 2 * 10**7):
            if want and "groupby_int64" not in want and f"groupby_int64_unclustered_{total}" not in want:          # one cardinality alone (profiling passes)
                continue
            keys = torch.randint(0, total, (ng,), generator=g, device="cuda", dtype=torch.int64) * 7919
            val = torch.randint(0, 10**6, (ng,), generator=g, device="cuda", dtype=torch.int64)
            torch.cuda.synchronize()
            b = ops.RecordBatch.from_arrays(ctx, ["k", "v"], [ctx.wrap_tensor(keys, capi.INT64), ctx.wrap_tensor(val, capi.INT64)])
            agg = ops.AggregateExec("Single", [(C("k", 0), "k")], [ops.AggregateFunctionExpr("SUM", C("v", 1), "s", input_field=F("v", capi.INT64)), ops.AggregateFunctionExpr("COUNT", None, "c")],
                                    ops.MemoryExec([[b]], b.schema))
            plan = ops.SortExec([ops.PhysicalSortExpr(C("s", 1), True, True), ops.PhysicalSortExpr(C("k", 0), False, False)], agg, fetch=10)
            dt, rows, kern, syncs = time_plan(ctx, ops, tc, plan, args.steps, args.warmup)
            res = result_columns(LAST_OUT[0]); LAST_OUT[0] = None
            kid = keys // 7919
            sums = torch.zeros(total, dtype=torch.int64, device="cuda").index_add_(0, kid, val); cnts = torch.zeros(total, dtype=torch.int64, device="cuda").index_add_(0, kid, torch.ones_like(val))
            top = torch.argsort(sums * (2 * total) + (total - 1 - torch.arange(total, device="cuda")), descending=True)[:10]       # s DESC, k ASC
            ok = np.array_equal(np.asarray(res[0]).astype(np.int64), (top * 7919).cpu().numpy()) and np.array_equal(np.asarray(res[1]).astype(np.int64), sums[top].cpu().numpy()) and \
                np.array_equal(np.asarray(res[2]).astype(np.int64), cnts[top].cpu().numpy())
            del kid, sums, cnts, top
            report(f"groupby_int64_unclustered_{total}", dt, ng, rows, 16, kern, syncs, {"cardinality": total, "result_check": check(f"groupby_int64_unclustered_{total}", ok, "the 10 result rows (k, SUM, COUNT) in order == torch index_add + argsort")})
            del keys, val, b, agg, plan
            torch.cuda.empty_cache()

    # ------------------------------------------------------------------ TPC-H-typed group-by: three key columns (Int64, Date32, Int32), SUM over Decimal128, 1 M groups
    if not want or "groupby_decimal_3key" in want:
        ng, total = int(1_000_000 * args.sf), max(1000, int(10_000 * args.sf))
        gidc = torch.randint(0, total, (ng,), generator=g, device="cuda", dtype=torch.int64)
        k0 = gidc * 7919; k1 = (8035 + gidc % 2400).to(torch.int32); k2 = (gidc % 3).to(torch.int32)
        val = dec_tensor(torch, ng, 90000, 10494951, g)
        torch.cuda.synchronize()
        b = ops.RecordBatch.from_arrays(ctx, ["k", "d", "p", "v"], [ctx.wrap_tensor(k0, capi.INT64), ctx.wrap_tensor(k1, capi.DATE32), ctx.wrap_tensor(k2, capi.INT32), ctx.wrap_tensor(val, capi.DECIMAL128, 15, 2)])
        agg = ops.AggregateExec("Single", [(C("k", 0), "k"), (C("d", 1), "d"), (C("p", 2), "p")],
                                [ops.AggregateFunctionExpr("SUM", C("v", 3), "s", input_field=F("v", capi.DECIMAL128, 15, 2)), ops.AggregateFunctionExpr("COUNT", None, "c")], ops.MemoryExec([[b]], b.schema))
        plan = ops.SortExec([ops.PhysicalSortExpr(C("s", 3), True, True), ops.PhysicalSortExpr(C("k", 0), False, False)], agg, fetch=10)
        dt, rows, kern, syncs = time_plan(ctx, ops, tc, plan, args.steps, args.warmup)
        res = result_columns(LAST_OUT[0]); LAST_OUT[0] = None
        sums = torch.zeros(total, dtype=torch.int64, device="cuda").index_add_(0, gidc, val[:, 0]); cnts = torch.zeros(total, dtype=torch.int64, device="cuda").index_add_(0, gidc, torch.ones_like(gidc))
        top = torch.argsort(sums * (2 * total) + (total - 1 - torch.arange(total, device="cuda")), descending=True)[:10]       # s DESC, k ASC
        ok = np.array_equal(np.asarray(res[0]).astype(np.int64), (top * 7919).cpu().numpy()) and np.array_equal(np.asarray(res[1]).astype(np.int64), (8035 + top % 2400).cpu().numpy()) and \
            np.array_equal(np.asarray(res[2]).astype(np.int64), (top % 3).cpu().numpy()) and [wrap64(x) for x in res[3][0]] == [wrap64(x) for x in sums[top].tolist()] and \
            np.array_equal(np.asarray(res[4]).astype(np.int64), cnts[top].cpu().numpy())
        del sums, cnts, top
        report("groupby_decimal_3key", dt, ng, rows, 8 + 4 + 4 + 16, kern, syncs, {"cardinality": total, "result_check": check("groupby_decimal_3key", ok, "the 10 result rows (k, d, p, SUM Decimal128, COUNT) in order == torch index_add + argsort")})
        del gidc, k0, k1, k2, val, b, agg, plan
        torch.cuda.empty_cache()

    # ------------------------------------------------------------------ SortExec at scale: ORDER BY l_extendedprice DESC, l_shipdate over 1 M x sf rows, 3 columns out
    if not want or "sort" in want:
        ns = int(1_000_000 * args.sf)
        price = dec_tensor(torch, ns, 90000, 10494951, g)
        sdate = torch.randint(8035, 10560, (ns,), generator=g, device="cuda", dtype=torch.int32)
        okey = torch.randint(1, 6 * 10**8, (ns,), generator=g, device="cuda", dtype=torch.int64)
        torch.cuda.synchronize()
        batch = ops.RecordBatch.from_arrays(ctx, ["l_orderkey", "l_extendedprice", "l_shipdate"],
                                            [ctx.wrap_tensor(okey, capi.INT64), ctx.wrap_tensor(price, capi.DECIMAL128, 15, 2), ctx.wrap_tensor(sdate, capi.DATE32)])
        plan = ops.SortExec([ops.PhysicalSortExpr(C("l_extendedprice", 1), True, True), ops.PhysicalSortExpr(C("l_shipdate", 2), False, False)], ops.MemoryExec([[batch]], batch.schema))

        def timed_sort():
            def step():
                out = [b for b in ops.with_fresh_state(plan).execute(0, tc)]
                with ctx.deferred_flags():
                    for b in out:
                        b.columns                        # the sorted columns, not only the order
                ctx.synchronize()
                LAST_OUT[0] = out if KEEP[0] else None
                return sum(b.num_rows for b in out)
            gc.collect(); gc.disable()
            for _ in range(max(args.warmup, 1)):
                step()
            ctx.profile_enable(True); ctx.profile_read(); step(); p = ctx.profile_read(); ctx.profile_enable(False)
            t0 = time.perf_counter(); per = []
            for it in range(args.steps):
                KEEP[0] = it == args.steps - 1
                t1 = time.perf_counter(); rows = step(); per.append(round((time.perf_counter() - t1) * 1e3, 3))
            dt = (time.perf_counter() - t0) / args.steps
            gc.enable(); STEP_MS[0] = per
            kern = {k: round(v[1], 3) for k, v in sorted(p.items(), key=lambda kv: -kv[1][1]) if not k.startswith("sync:")}
            return dt, rows, kern, sum(v[0] for k, v in p.items() if k.startswith("sync:"))
        dt, rows, kern, syncs = timed_sort()
        # result check: the whole permutation against a stable torch sort of the composite key (both sorts are stable: ties keep input order)
        outb = LAST_OUT[0]; LAST_OUT[0] = None
        comp = (price[:, 0].max() - price[:, 0]) * 4096 + (sdate.to(torch.int64) - 8035)
        perm = torch.sort(comp, stable=True).indices; del comp
        got_key = torch.cat([dev_tensor(torch, b.columns[0], "<i8") for b in outb])
        got_date = torch.cat([dev_tensor(torch, b.columns[2], "<i4") for b in outb])
        ok = rows == ns and bool(torch.equal(got_key, okey[perm])) and bool(torch.equal(got_date, sdate[perm]))
        del perm, got_key, got_date, outb
        report("sort", dt, ns, rows, 2 * (8 + 16 + 4), kern, syncs, {"sort_keys": "Decimal128 DESC NULLS FIRST, Date32 ASC NULLS LAST",
               "result_check": check("sort", ok, "l_orderkey and l_shipdate of every output row == the input gathered through a stable torch sort of the composite key")})
        del price, sdate, okey, batch, plan
        torch.cuda.empty_cache()

    # ------------------------------------------------------------------ RepartitionExec: lineitem's Q3 columns hashed on l_orderkey into 8 partitions
    if not want or "partition" in want:
        price, disc = dec_tensor(torch, n, 90000, 10494951, g), dec_tensor(torch, n, 0, 11, g)
        sdate = torch.randint(8035, 10560, (n,), generator=g, device="cuda", dtype=torch.int32)
        torch.cuda.synchronize()
        batch = ops.RecordBatch.from_arrays(ctx, ["l_orderkey", "l_extendedprice", "l_discount", "l_shipdate"],
                                            [ctx.wrap_tensor(l_orderkey, capi.INT64), ctx.wrap_tensor(price, capi.DECIMAL128, 15, 2), ctx.wrap_tensor(disc, capi.DECIMAL128, 15, 2), ctx.wrap_tensor(sdate, capi.DATE32)])
        for nparts in ((8, 64) if not __import__("os").environ.get("DFGPU_BENCH_NPARTS") else (int(__import__("os").environ["DFGPU_BENCH_NPARTS"]),)):                      # 8: one node's GPUs; 64: a target_partitions-sized fan-out (the LDS-staged stable scatter)
            if want and "partition" not in want and not any(w in "partition_hash_%d" % nparts for w in want):
                continue
            plan = ops.RepartitionExec(ops.MemoryExec([[batch]], batch.schema), ops.Partitioning.Hash([C("l_orderkey", 0)], nparts))

            def timed_partition():
                def step():
                    p2 = ops.with_fresh_state(plan); rows = 0; keep = []
                    with ctx.deferred_flags():
                        for d in range(nparts):
                            for b in p2.execute(d, tc):
                                b.columns; rows += b.num_rows; keep.append((d, b))
                    ctx.synchronize()
                    LAST_OUT[0] = keep if KEEP[0] else None
                    return rows
                gc.collect(); gc.disable()
                for _ in range(max(args.warmup, 1)):
                    step()
                ctx.profile_enable(True); ctx.profile_read(); step(); pr = ctx.profile_read(); ctx.profile_enable(False)
                t0 = time.perf_counter(); per = []
                for it in range(args.steps):
                    KEEP[0] = it == args.steps - 1
                    t1 = time.perf_counter(); rows = step(); per.append(round((time.perf_counter() - t1) * 1e3, 3))
                dt = (time.perf_counter() - t0) / args.steps
                gc.enable(); STEP_MS[0] = per
                kern = {k: round(v[1], 3) for k, v in sorted(pr.items(), key=lambda kv: -kv[1][1]) if not k.startswith("sync:")}
                return dt, rows, kern, sum(v[0] for k, v in pr.items() if k.startswith("sync:"))
            dt, rows, kern, syncs = timed_partition()
            assert rows == n
            # result check: rows conserved (count, wrapping sums of key / price / date), and no key in two partitions: owner[order index] is written by every
            # partition for its rows and read back -- a key that two partitions hold fails the read-back of the one that wrote first
            keep = LAST_OUT[0]; LAST_OUT[0] = None
            owner = torch.full((n_orders,), -1, dtype=torch.int8, device="cuda"); sk = sp = sd = 0; parts = []
            for d, b in keep:
                if not b.num_rows:
                    continue
                kd = dev_tensor(torch, b.columns[0], "<i8"); kk = kd - 1; oi = (kk // 32) * 8 + (kk % 32)
                owner[oi] = d; parts.append((d, oi)); sk += int(kd.sum().item())
                sp += int(dev_tensor(torch, b.columns[1], "<i8", 2)[:, 0].sum().item()); sd += int(dev_tensor(torch, b.columns[3], "<i4").to(torch.int64).sum().item())
                del kd, kk
            ok = all(bool((owner[oi] == d).all().item()) for d, oi in parts)
            ok = ok and wrap64(sk) == wrap64(int(l_orderkey.sum().item())) and wrap64(sp) == wrap64(int(price[:, 0].sum().item())) and sd == int(sdate.to(torch.int64).sum().item())
            del keep, owner, parts, b, oi          # `b` too: a partition's batch is a slice of the partition-major output buffers, the last one left bound by the loop held all of them (35 GB)
            report("partition_hash_%d" % nparts, dt, n, rows, 2 * (8 + 16 + 16 + 4), kern, syncs, {"partitions": nparts,
                   "result_check": check("partition_hash_%d" % nparts, ok, "row count and wrapping column sums conserved; every l_orderkey value lives in exactly one output partition")})
            del plan
        del price, disc, sdate, batch
        torch.cuda.empty_cache()

    # ------------------------------------------------------------------ ParquetExec: lineitem-shaped file (written by pyarrow here) -> columns in HBM
    if not want or any("parquet" in w or "csv" in w for w in want):
        import os
        import tempfile
        import numpy as np
        import pyarrow.parquet as pq
        from dfgpu.parquet import ParquetFile
        nr = int(min(12_000_000, 120_000 * args.sf)) or 1000
        rng = np.random.default_rng(11)

        def dec(lo, hi):
            v = rng.integers(lo, hi, nr).astype(np.int64)
            buf = np.empty((nr, 2), dtype=np.int64); buf[:, 0] = v; buf[:, 1] = v >> 63
            return pa.Array.from_buffers(pa.decimal128(15, 2), nr, [None, pa.py_buffer(buf.tobytes())])
        pick = lambda words: pa.DictionaryArray.from_arrays(pa.array(rng.integers(0, len(words), nr).astype(np.int32)), pa.array(words)).cast(pa.string())
        table = pa.table({"l_orderkey": pa.array(np.sort(rng.integers(0, nr // 4 * 32, nr)).astype(np.int64)), "l_quantity": dec(100, 5001), "l_extendedprice": dec(90000, 10494951),
                          "l_discount": dec(0, 11), "l_shipdate": pa.array(rng.integers(8035, 10560, nr).astype(np.int32), type=pa.date32()),
                          "l_returnflag": pick(["A", "N", "R"]), "l_linestatus": pick(["F", "O"]), "l_shipmode": pick(["AIR", "FOB", "MAIL", "RAIL", "REG AIR", "SHIP", "TRUCK"])})
        decoded = nr * (8 + 3 * 16 + 4 + 3 * 4)

        def scan_check(cols, tab):
            """decoded device columns (Array objects) against the source table, column by column"""
            for i, c in enumerate(cols):
                a = c.to_arrow()
                if pa.types.is_dictionary(a.type):
                    a = a.cast(a.type.value_type)
                w = tab.column(i).combine_chunks().slice(0, len(a))
                if len(a) != len(w) or not a.equals(w.cast(a.type)):
                    return False
            return True
        for label, kw in (("snappy_dict", dict(compression="snappy", use_dictionary=True)), ("zstd_dict", dict(compression="zstd", use_dictionary=True)), ("plain", dict(compression="none", use_dictionary=False))):
            name = "parquet_scan_" + label
            if want and not any(w in name for w in want):
                continue
            path = os.path.join(tempfile.gettempdir(), f"dfgpu_bench_{os.getpid()}_{label}.parquet")
            pq.write_table(table, path, row_group_size=1 << 20, **kw)
            fbytes = os.path.getsize(path)
            try:
                f = ParquetFile(ctx, path=path, stage_on_device=True)

                def step():
                    cols = f.read()
                    ctx.synchronize()
                    LAST_OUT[0] = cols if KEEP[0] else None
                    return len(cols[0])
                gc.collect(); gc.disable()
                for _ in range(max(args.warmup, 1)):
                    step()
                ctx.profile_enable(True); ctx.profile_read(); step(); pr = ctx.profile_read(); ctx.profile_enable(False)
                t0 = time.perf_counter(); per = []
                for it in range(args.steps):
                    KEEP[0] = it == args.steps - 1
                    t1 = time.perf_counter(); rows = step(); per.append(round((time.perf_counter() - t1) * 1e3, 3))
                dt = (time.perf_counter() - t0) / args.steps
                gc.enable(); STEP_MS[0] = per
                scan_ok = scan_check(LAST_OUT[0], table); LAST_OUT[0] = None
                kern = {k: round(v[1], 3) for k, v in sorted(pr.items(), key=lambda kv: -kv[1][1]) if not k.startswith("sync:")}
                syncs = sum(v[0] for k, v in pr.items() if k.startswith("sync:"))
                f.close()
                fh = ParquetFile(ctx, path=path, stage_on_device=False)       # the same read with the file image in host memory: column chunks cross PCIe first
                fh.read(); ctx.synchronize(); hs = []
                for _ in range(5):
                    t0 = time.perf_counter(); fh.read(); ctx.synchronize(); hs.append(time.perf_counter() - t0)
                dth = sorted(hs)[len(hs) // 2]
                fh.close()
            finally:
                os.unlink(path)
            assert rows == nr
            report(name, dt, nr, rows, decoded // nr, kern, syncs, {"file_bytes": fbytes, "decoded_bytes": decoded, "row_groups": (nr + (1 << 20) - 1) >> 20, "decoded_GBps": round(decoded / dt / 1e9, 1),
                                                                    "file_GBps": round(fbytes / dt / 1e9, 1), "from_host_image_ms": round(dth * 1e3, 1), "from_host_image_reads_ms": [round(x * 1e3, 1) for x in hs], "from_host_image_decoded_GBps": round(decoded / dth / 1e9, 2),
                                                                    "host_image": "the file mapped and page-locked at open; column chunks cross PCIe on a copy stream while the decode kernels run (median of 5 reads)",
                                                                    "columns": "Int64 key, 3 x Decimal128(15,2) (FIXED_LEN_BYTE_ARRAY), Date32, 3 x Utf8 kept as Dictionary(Int32, Utf8)",
                                                                    "result_check": check(name, scan_ok, "every decoded column == the pyarrow table the file was written from (values compared after the D2H copy)")})
        # ---- CsvExec's per-file work: the same table as text (written by pyarrow.csv), image resident in HBM -> columns in HBM
        if not want or any(w in "csv_scan" for w in want):
            import io
            import pyarrow.csv as pcsv
            from dfgpu import capi
            from dfgpu.csv import read_csv
            nc = min(nr, 6_000_000)
            buf = io.BytesIO(); pcsv.write_csv(table.slice(0, nc), buf); img = buf.getvalue(); del buf
            dimg = torch.frombuffer(bytearray(img), dtype=torch.uint8).cuda()
            sch = [("l_orderkey", capi.INT64, 0, 0)] + [(c, capi.DECIMAL128, 15, 2) for c in ("l_quantity", "l_extendedprice", "l_discount")] + [("l_shipdate", capi.DATE32, 0, 0)] + \
                  [(c, capi.UTF8, 0, 0) for c in ("l_returnflag", "l_linestatus", "l_shipmode")]

            def step():
                cols = read_csv(ctx, dimg, sch, on_device=True)
                ctx.synchronize()
                LAST_OUT[0] = cols if KEEP[0] else None
                return len(cols[0])
            gc.collect(); gc.disable()
            for _ in range(max(args.warmup, 1)):
                step()
            ctx.profile_enable(True); ctx.profile_read(); step(); pr = ctx.profile_read(); ctx.profile_enable(False)
            t0 = time.perf_counter(); per = []
            for it in range(args.steps):
                KEEP[0] = it == args.steps - 1
                t1 = time.perf_counter(); rows = step(); per.append(round((time.perf_counter() - t1) * 1e3, 3))
            dt = (time.perf_counter() - t0) / args.steps
            gc.enable(); STEP_MS[0] = per
            kern = {k: round(v[1], 3) for k, v in sorted(pr.items(), key=lambda kv: -kv[1][1]) if not k.startswith("sync:")}
            syncs = sum(v[0] for k, v in pr.items() if k.startswith("sync:"))
            assert rows == nc
            csv_ok = scan_check(LAST_OUT[0], table.slice(0, nc)); LAST_OUT[0] = None
            report("csv_scan", dt, nc, rows, len(img) // nc, kern, syncs, {"file_bytes": len(img), "file_GBps": round(len(img) / dt / 1e9, 1),
                                                                                    "columns": "Int64 key, 3 x Decimal128(15,2), Date32, 3 x Utf8; text written by pyarrow.csv, image resident in HBM",
                                                                                    "result_check": check("csv_scan", csv_ok, "every parsed column == the pyarrow table the text was written from")})
            del dimg, img
        del table

    # ------------------------------------------------------------------ ClickBench-style string-key group-by (100 M rows at sf 100)
    nrows = int(1_000_000 * args.sf)
    for card, zipf in ((1000, False), (1_000_000, False), (20_000_000, False), (1_000_000, True)):
        card = max(10, int(card * min(1.0, args.sf / 100.0)) if card > 1000 else card)
        name = f"clickbench_{'zipf' if zipf else 'uniform'}_{card}"
        if want and not any(w in name for w in want):
            continue
        import numpy as np
        rng = np.random.default_rng(7)
        if zipf:
            ids = torch.from_numpy((rng.zipf(1.1, nrows) % card).astype(np.int32)).cuda()
        else:
            ids = torch.randint(0, card, (nrows,), generator=g, device="cuda", dtype=torch.int32)
        # dictionary values: URL-like strings of 20-40 bytes, entry 0 is the empty string (filtered out)
        words = pa.array([""] + [f"https://site{k}.example/{k * 7919 % 1000}" for k in range(1, card)], type=pa.utf8())
        dictionary = ctx.from_arrow(words)
        length = torch.randint(0, 500, (nrows,), generator=g, device="cuda", dtype=torch.int32)
        w = torch.randint(0, 10**6, (nrows,), generator=g, device="cuda", dtype=torch.int64)
        torch.cuda.synchronize()
        batch = ops.RecordBatch.from_arrays(ctx, ["key", "len", "w"], [wrap_dict(ctx, capi, ids, dictionary, capi.INT32), ctx.wrap_tensor(length, capi.INT32), ctx.wrap_tensor(w, capi.INT64)])
        src = ops.MemoryExec([[batch]], batch.schema)
        f = ops.CoalesceBatchesExec(ops.FilterExec(B(C("key", 0), "!=", L("", pa.utf8())), src), 8192)
        proj = ops.ProjectionExec([(C("key", 0), "key"), (ops.CastExpr(C("len", 1), capi.FLOAT64), "lenf"), (C("w", 2), "w")], f)
        aggs = [ops.AggregateFunctionExpr("AVG", C("lenf", 1), "l", input_field=F("x", capi.FLOAT64)), ops.AggregateFunctionExpr("COUNT", None, "c"),
                ops.AggregateFunctionExpr("MAX", C("w", 2), "m", input_field=F("x", capi.INT64))]
        agg = ops.AggregateExec("Single", [(C("key", 0), "k")], aggs, proj)
        having = ops.FilterExec(B(C("c", 2), ">", L(3, pa.int64())), agg)
        plan = ops.SortExec([ops.PhysicalSortExpr(C("l", 1), True, True), ops.PhysicalSortExpr(C("k", 0), False, False)], having, fetch=25)
        dt, rows, kern, syncs = time_plan(ctx, ops, tc, plan, args.steps, args.warmup)
        # result check: AVG / COUNT / MAX per key with torch index_add / scatter_reduce, HAVING c > 3, top 25 by (avg DESC, key ASC)
        res = result_columns(LAST_OUT[0]); LAST_OUT[0] = None
        nz = ids != 0; idl = ids[nz].to(torch.int64)
        cs = torch.zeros(card, dtype=torch.int64, device="cuda").index_add_(0, idl, torch.ones_like(idl))
        ls_ = torch.zeros(card, dtype=torch.int64, device="cuda").index_add_(0, idl, length[nz].to(torch.int64))
        mx = torch.full((card,), -1, dtype=torch.int64, device="cuda").scatter_reduce_(0, idl, w[nz], reduce="amax")
        have = torch.nonzero(cs > 3).flatten()
        avg = ls_[have].to(torch.float64) / cs[have].to(torch.float64)
        kth = torch.topk(avg, min(25, avg.numel())).values[-1] if avg.numel() else None
        cand = have[avg >= kth].tolist() if kth is not None else []                  # everything tied with the 25th average takes part in the key tie-break
        cand.sort(key=lambda k: (-(ls_[k].item() / cs[k].item()), words[k].as_py()))
        expect = [(words[k].as_py(), ls_[k].item() / cs[k].item(), int(cs[k].item()), int(mx[k].item())) for k in cand[:25]]
        got = [(k, float(a), int(c_), int(m)) for k, a, c_, m in zip(res[0], res[1], res[2], res[3])]
        ok = len(got) == len(expect) and all(g[0] == w_[0] and abs(g[1] - w_[1]) <= 1e-9 * abs(w_[1]) and g[2:] == w_[2:] for g, w_ in zip(got, expect))
        del nz, idl, cs, ls_, mx, have, avg
        report(name, dt, nrows, rows, 4 + 4 + 8, kern, syncs, {"cardinality": card, "result_check": check(name, ok, "the 25 result rows (key, AVG, COUNT, MAX) in order == torch index_add / scatter_reduce + HAVING + top 25")})
        del ids, length, w, batch, src, plan, dictionary
        torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    main()
