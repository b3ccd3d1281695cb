#!/usr/bin/env python3
"""bench.py -- TPC-H SF100 Q3 (hash-join + aggregate + sort) through the dfgpu operator layer on N MI355X.

One "step" = one full pass of the Q3 physical plan (reference plan shape: sqllogictest/test_files/tpch/q3.slt.part)
over synthetic TPC-H-shaped Arrow columns that are already resident in HBM when the clock starts (the analogue of
the reference's `--mem-table` mode, benchmarks/src/tpch/run.rs:72-74).  value = input rows (customer + orders +
lineitem) of the whole job / wall time, max over ranks.

Extra objects in the JSON line:
  roofline     dominant kernel by device time: algorithmic bytes per launch / average launch duration (HIP events
               recorded on the stream the kernel runs on, through dfgpu_profile_*), against 8 TB/s HBM3E.
  cpu_baseline the oracle's restatement of the same plan (oracle/dfo_tpch.c, "port") timed on this box's host cores
               over a bounded SF sample of the same workload; reported, never the target.
  result_check (N = 1) the step's result against a recomputation of Q3 with plain torch ops over the same tensors, outside the timed region.
  q3_general_paths (N = 1) the same plan with the rank index and the run numbering switched off: joins through the radix-partitioned /
               open-addressing hash tables, group-by through the hash table -- the paths unsorted keys take.
  workloads    (N = 1) the other target plan shapes (bench_workloads.py): Q1 (Decimal128 and Float64), Q5, Q18, the sparse-key hash join,
               unclustered group-bys, SortExec, the ClickBench Q28 shape, each with ms_per_step, rows_per_s and its roofline figures.
  plans        (N > 1) ms per step of the other two distribution plans, measured after the timed region.
Output: the LAST stdout line is one compact JSON object (< 8 KB: the contract's keys, roofline, cpu_baseline, result_check.ok, one ms_per_step
scalar per nested workload); the full result object goes to bench_detail.json next to this file and to stderr.
Launch for N > 1:  python bench.py --gpus N  starts the N ranks itself (a child torch.distributed.run, before this process touches the GPU);
under an external launcher (WORLD_SIZE set) it is one of the ranks and WORLD_SIZE must equal --gpus.
"""
import argparse
import gc
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
Q3_BYTES_PER_ROW = 39.4         # 30.2 GB algorithmic bytes / 765,037,902 input rows at SF100

# algorithmic bytes per row for the kernels that can dominate (DESIGN.md "kernels"): columns that must be read +
# results that must be written once; hash tables and other intermediates are not counted.
KERNEL_BYTES_PER_ROW = {
    "k_probe_match_bitmap": 8 + 0.125 + 0.125,   # probe key (Int64) + selection bit + match bit
    "k_probe_match_hash": 8 + 0.125 + 0.125,
    "k_compare_scalar_fast": 4 + 0.125,
    "k_join_build": 8 + 0.125 + 4,            # build key + selection bit + row->slot word
    "k_compare": 4 + 0.125,                   # Date32 operand + result bit
    "k_take_fixed": 4 + 8 + 8,                # index + gathered value + written value (8-byte column)
    "k_groups_find": 16 + 4,                  # 3 key columns (8+4+4) + group-id word
    "k_acc_update": 16 + 4,                   # Decimal128 value + group id
    "k_arith": 32 + 16,                       # two Decimal128 operands + result
}


def acero_q3(host, tpch, n_in):
    """TPC-H Q3 over the same host sample with pyarrow Acero (Table.join / group_by / sort_by, all host threads); money as Float64
    because Acero's decimal typing differs from DataFusion's.  1 warm-up + 1 timed run."""
    import numpy as np
    import pyarrow as pa
    import pyarrow.compute as pc
    cust = pa.table({"c_custkey": host["c_custkey"], "seg": host["c_mktsegment"]})
    orders = pa.table({"o_orderkey": host["o_orderkey"], "o_custkey": host["o_custkey"], "o_orderdate": host["o_orderdate"], "o_shippriority": host["o_shippriority"]})
    line = pa.table({"l_orderkey": host["l_orderkey"], "l_extendedprice": host["l_extendedprice"][:, 0].astype(np.float64) / 100.0,
                     "l_discount": host["l_discount"][:, 0].astype(np.float64) / 100.0, "l_shipdate": host["l_shipdate"]})
    seg = tpch.SEGMENTS.index(tpch.Q3_SEGMENT)

    def run():
        t = time.perf_counter()
        c = cust.filter(pc.equal(cust["seg"], seg)).select(["c_custkey"])
        o = orders.filter(pc.less(orders["o_orderdate"], tpch.Q3_DATE))
        l = line.filter(pc.greater(line["l_shipdate"], tpch.Q3_DATE))
        l = l.append_column("rev", pc.multiply(l["l_extendedprice"], pc.subtract(1.0, l["l_discount"]))).select(["l_orderkey", "rev"])
        j1 = o.join(c, keys="o_custkey", right_keys="c_custkey", join_type="inner").select(["o_orderkey", "o_orderdate", "o_shippriority"])
        j2 = l.join(j1, keys="l_orderkey", right_keys="o_orderkey", join_type="inner")
        g = j2.group_by(["l_orderkey", "o_orderdate", "o_shippriority"]).aggregate([("rev", "sum")])
        r = g.sort_by([("rev_sum", "descending"), ("o_orderdate", "ascending")])
        return time.perf_counter() - t, r.num_rows
    run()
    dt, rows = run()
    return {"value": round(n_in / dt, 1), "unit": "rows/s", "seconds": round(dt, 3), "result_rows": rows,
            "note": f"pyarrow {pa.__version__} Acero, independent engine (not DataFusion), Float64 money, same SF sample, all host threads"}


def measure_copy_bandwidth(torch):
    """Device-to-device copy of 4 GiB with torch's copy kernel: the practical HBM ceiling on this box (read + write bytes / time)."""
    a = torch.empty(1 << 30, dtype=torch.int32, device="cuda")
    b = torch.empty_like(a)
    for _ in range(2):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    del a, b
    torch.cuda.empty_cache()
    return 2 * 4 * (1 << 30) / (ms * 1e-3) / 1e9


METRIC = "rows/sec hash-join+agg, TPC-H SF100 Q3"
PARALLELISM = {"colocated": "%d GPUs, range-sharded tables: RCCL all-gather of the customer build side, partition-local orders-lineitem join + aggregation, gather of sorted partitions",
               "broadcast": "%d GPUs, CollectLeft joins: RCCL all-gather of build sides + all-to-all of partial aggregates",
               "shuffle": "%d GPUs, partitioned joins: hash partition + RCCL all-to-all per exchange"}
LINE_LIMIT = 8192       # hard cap of the stdout line (the driver's parser lost the 26.8 KB line of round 3); the target is 4 KB

ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launches_per_step", "avg_launch_ms", "algorithmic_bytes_per_launch", "measured_copy_GBps")
CPU_KEYS = ("value", "unit", "cores", "cpu_model", "kind", "sample")


def compact_line(detail):
    """The one stdout line: the contract's keys, the roofline and cpu_baseline objects cut to scalars, one ms_per_step scalar per nested
    workload.  Everything else (per-kernel tables, notes, step arrays) stays in bench_detail.json / stderr.  Same shape as the reference's
    own report: one query, its iterations averaged (benchmarks/src/tpch/run.rs:120-156)."""
    top = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: detail.get(k) for k in top}
    cfg = detail.get("config") or {}
    line["config"] = {k: cfg[k] for k in ("workload", "input_rows", "result_rows", "parallelism", "paths", "plan") if k in cfg}
    rf = detail.get("roofline")
    line["roofline"] = {k: rf.get(k) for k in ROOFLINE_KEYS} if rf else None
    if rf and isinstance(rf.get("host_syncs_per_step"), dict):
        line["host_syncs_per_step"] = sum(rf["host_syncs_per_step"].values())
    cb = detail.get("cpu_baseline")
    line["cpu_baseline"] = {k: cb.get(k) for k in CPU_KEYS} if cb else None
    if cb and isinstance(cb.get("acero"), dict) and "value" in cb["acero"]:
        line["cpu_baseline"]["acero_rows_per_s"] = cb["acero"]["value"]
    if detail.get("result_check") is not None:
        line["result_check"] = {"ok": bool(detail["result_check"].get("ok"))}
    line["ranks_seen"] = detail.get("ranks_seen", 1)
    ms = {}
    for key in ("q3_general_paths", "q3_shuffled_inputs"):
        if detail.get(key):
            ms[key] = detail[key]["ms_per_step"]
    for name, w in (detail.get("workloads") or {}).items():
        ms[name] = w.get("ms_per_step")
    for name, w in (detail.get("plans") or {}).items():
        ms["q3_plan_" + name] = w.get("ms_per_step")
    if ms:
        line["ms_per_step_other"] = ms
    line["detail"] = "bench_detail.json"
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_LIMIT:          # never lose the headline to a long string: drop the optional parts
        line.pop("ms_per_step_other", None)
        if line.get("cpu_baseline"):
            line["cpu_baseline"]["sample"] = line["cpu_baseline"]["sample"][:200]
        text = json.dumps(line, separators=(",", ":"))
    assert len(text) <= LINE_LIMIT, len(text)
    return text


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--sf", type=float, default=100.0, help="TPC-H scale factor of the whole job")
    ap.add_argument("--cpu-sf", type=float, default=30.0, help="scale factor of the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the N > 1 path on one GPU)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--no-shuffled", action="store_true", help="N = 1: skip Q3 over row-wise permuted tables")
    ap.add_argument("--no-workloads", action="store_true", help="N = 1: skip the nested other plan shapes (bench_workloads.py)")
    ap.add_argument("--workloads", default="q1_decimal,q1_float64,q5,q18,hash_join,groupby_int64,groupby_decimal_3key,sort,partition,parquet_scan,csv_scan,clickbench_uniform_1000000,clickbench_zipf_1000000",
                    help="N = 1: which plan shapes of bench_workloads.py to nest under \"workloads\"")
    ap.add_argument("--collective-deadline", type=float, default=180.0, help="N > 1: seconds without progress (no exchange started, no step finished) after which a rank reports where it stands and exits with code 3 instead of hanging the job")
    ap.add_argument("--native-exchange", action="store_true", help="N > 1 workloads: ShuffleExec through the C entry point dfgpu_exchange (RCCL inside libdfgpu.so) instead of torch.distributed collectives")
    ap.add_argument("--option", action="append", default=[], metavar="KEY=VALUE", help="ctx option set before the run (dfgpu_ctx_set_option), e.g. --option mailbox_readback=0 for an A/B run")
    ap.add_argument("--launch-check", action="store_true", help="only rendezvous the ranks (gloo, CPU) and report ranks_seen: rehearses the launcher without a GPU")
    ap.add_argument("--detail", default="", help="where the full result object goes (default: bench_detail.json next to bench.py); it is also written to stderr")
    ap.add_argument("--plan", choices=["colocated", "broadcast", "shuffle"], default="shuffle",
                    help="N > 1: colocated = customer build side broadcast, orders-lineitem join and aggregation partition-local (the shards are co-partitioned on "
                         "the order key, as TPC-H files are); broadcast = both build sides all-gathered (CollectLeft), partial aggregates shuffled; "
                         "shuffle = the reference's fully partitioned plan (hash repartition of every join / aggregate input)")
    return ap.parse_args()


class Watchdog:
    """N > 1: a rank stuck in a collective (a peer died, a lane mismatch) would otherwise hang the whole job until the driver's limit.  The main thread
    reports every exchange it starts and every step it finishes; a daemon thread exits the process (os._exit: no re-exec, no cleanup that could block
    on the GPU) with a line that says which rank stopped where once nothing has moved for `deadline` seconds."""

    def __init__(self, rank, deadline):
        import threading
        self.rank, self.deadline, self.phase, self.t, self.armed = rank, deadline, "start", time.time(), False
        threading.Thread(target=self._run, daemon=True).start()

    def beat(self, phase):
        self.phase, self.t = phase, time.time()

    def arm(self, on, phase=""):
        self.armed = on; self.beat(phase or self.phase)

    def _run(self):
        while True:
            time.sleep(1.0)
            if self.armed and time.time() - self.t > self.deadline:
                sys.stderr.write(json.dumps({"watchdog": "no progress", "rank": self.rank, "seconds": round(time.time() - self.t, 1), "last": self.phase}) + "\n"); sys.stderr.flush()
                os._exit(3)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks ourselves (one process per GPU, the reference's one task per
    partition: physical-plan/src/lib.rs:712-749) as a CHILD torch.distributed.run before this process has touched the GPU or imported torch,
    forward its output and return its exit code.  Never an exec of a process that has initialised the GPU."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench.py: launching %d ranks: %s\n" % (args.gpus, " ".join(cmd))); sys.stderr.flush()
    return subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))).returncode


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE')}: launch one rank per GPU")
    if args.launch_check:
        # launcher rehearsal (tests/test_bench_line.py): rendezvous over gloo on the CPU, count the ranks, touch no GPU
        import torch
        import torch.distributed as dist
        world, rank = args.gpus, int(os.environ.get("RANK", "0"))
        seen = 1
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            t = torch.ones(1, dtype=torch.int64); dist.all_reduce(t); seen = int(t.item())
            dist.destroy_process_group()
        assert seen == args.gpus
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "ranks_seen": seen}), flush=True)
        return
    import torch
    import torch.distributed as dist
    import dfgpu
    from dfgpu import exchange, physical_plan as ops, tpch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    # one ctx on torch's current stream: dfgpu kernels, torch ops and RCCL collectives are stream ordered
    ctx = dfgpu.Context(local_rank, stream=torch.cuda.current_stream().cuda_stream)
    for kv in args.option:
        k_, v_ = kv.split("=", 1); ctx.set_option(k_, int(v_))
    tc = ops.TaskContext(ctx, batch_size=8192)
    tensors = tpch.gen_device_tensors(args.sf, rank=rank, world=world)
    tables = tpch.tables_from_torch(ctx, tensors)
    if world > 1:
        tensors = None
    rows_local = sum(t.num_rows for t in tables.values())
    rows_total = rows_local
    if world > 1:
        t = torch.tensor([rows_local], dtype=torch.int64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t)
        rows_total = int(t.item())
    torch.cuda.synchronize()

    result_rows, last_out = [0], [None]
    ranks_seen = 1
    wd = None
    if world > 1:
        t = torch.ones(1, dtype=torch.int64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t)
        ranks_seen = int(t.item())                      # every rank of the launch takes part in the collectives (reported in the JSON line)
        assert ranks_seen == args.gpus, f"{ranks_seen} ranks answered the all-reduce, --gpus {args.gpus}"
        wd = Watchdog(rank, args.collective_deadline)
        exchange.PROGRESS = wd.beat

    PLANS = {"colocated": tpch.q3_colocated_plan, "broadcast": tpch.q3_broadcast_plan, "shuffle": tpch.q3_distributed_plan}
    Q3_OUTPUT = ["l_orderkey", "revenue", "o_orderdate", "o_shippriority"]
    # The plan description is built once; every step executes a copy with fresh run-once state (no cached build side).  N > 1: the
    # colocated plan is cut at its one exchange into two prebuilt segments (tpch.Q3ColocatedStaged); the other plans are rebuilt per step.
    template = tpch.q3_plan(tables, batch_size=8192) if world == 1 else None
    staged = tpch.Q3ColocatedStaged(tables, batch_size=8192) if world > 1 and args.plan == "colocated" else None
    C_ = ops.Column
    final_keys = [ops.PhysicalSortExpr(C_("revenue", 1), True, True), ops.PhysicalSortExpr(C_("o_orderdate", 2), False, False)]
    final_slot = final_plan = None
    if world > 1 and rank == 0:          # rank 0's merge of the gathered partitions: built once around an input slot, like the staged plan
        import pyarrow as pa
        empty = ops.batch_from_arrow(ctx, pa.table({"l_orderkey": pa.array([], type=pa.int64()), "revenue": pa.array([], type=pa.decimal128(38, 4)),
                                                    "o_orderdate": pa.array([], type=pa.date32()), "o_shippriority": pa.array([], type=pa.int32())}))
        final_slot = ops.MemoryExec([[empty]], empty.schema)
        final_plan = ops.SortExec(final_keys, final_slot)
        final_plan.handle(tc)

    def step():
        if world == 1:
            plan = ops.with_fresh_state(template)
            out = [b for b in plan.execute(0, tc)]
        else:
            wd.arm(True, "Q3 step (%s plan)" % args.plan)
            plan = staged if staged is not None else PLANS[args.plan](tables, batch_size=8192)
            with ctx.deferred_flags():          # one error-flag read-back for the rank's plan + gather instead of one per materialised column
                local = [b for b in plan.execute(0, tc)]
                mine = ops.concat_batches(local[0].schema, local) if local else None
                gathered = exchange.gather_batches(ctx, None, mine, 0, names=Q3_OUTPUT)       # ≙ SortPreservingMergeExec gathering the sorted partitions
            out = []
            if rank == 0 and gathered.num_rows:
                final_slot.replace([[gathered]])
                out = [b for b in ops.with_fresh_state(final_plan).execute(0, tc)]
        ctx.synchronize()
        if wd is not None:
            wd.arm(False, "Q3 step done")
        result_rows[0] = sum(b.num_rows for b in out)
        last_out[0] = out

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Per-kernel breakdown: one untimed step with every kernel bracketed by HIP events (the last warm-up step, or one step
    # after the timed region when --warmup 0).  An event pair costs ~10 us of stream bubble, so the timed region itself
    # brackets only the dominant kernel found here -- that is the measurement the roofline object reports.
    def profiled_step():
        ctx.profile_select(None)
        ctx.profile_enable(True)
        ctx.profile_read()
        step()
        p = ctx.profile_read()
        ctx.profile_enable(False)
        return p

    breakdown = None
    gc.collect(); gc.disable()          # the timed region must not contain a cyclic-GC pass of the Python wrappers (bench_workloads.py: one such pass cost a single step 65 ms)
    for w in range(args.warmup):
        if w == args.warmup - 1:
            breakdown = profiled_step()
        else:
            step()
    def split_syncs(p):
        return ({k: v for k, v in p.items() if not k.startswith("sync:")}, {k[5:]: v[0] for k, v in p.items() if k.startswith("sync:")})

    host_syncs = None
    if breakdown:
        breakdown, host_syncs = split_syncs(breakdown)
    dominant = max(breakdown.items(), key=lambda kv: kv[1][1])[0] if breakdown else "k_probe_match_bitmap"
    ctx.profile_select(dominant)
    ctx.profile_enable(True)
    ctx.profile_read()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    prof, _ = split_syncs(ctx.profile_read())
    ctx.profile_enable(False)
    if breakdown is None:
        breakdown, host_syncs = split_syncs(profiled_step())
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = elapsed / args.steps * 1e3
    value = rows_total * args.steps / elapsed

    # ---- roofline of the dominant kernel (rank 0's view; every rank runs the same kernels on its shard)
    roofline = None
    if prof:
        name, (launches, total_ms) = max(prof.items(), key=lambda kv: kv[1][1])
        per_step = launches / args.steps
        avg_ms = total_ms / launches
        # rows one launch of that kernel processes in this plan (its largest launch dominates the average)
        line_rows = tables["lineitem"].num_rows
        probe_rows = (tables["orders"].num_rows + line_rows) / 2          # two probe launches per step: orders, lineitem
        rows_per_launch = {"k_probe_match_bitmap": probe_rows, "k_probe_match_hash": probe_rows,
                           "k_compare_scalar_fast": (tables["orders"].num_rows + line_rows) / 2,
                           "k_compare": (tables["customer"].num_rows + tables["orders"].num_rows + line_rows) / 3}.get(name, None)
        bpr = KERNEL_BYTES_PER_ROW.get(name)
        tpath = None
        if bpr is not None and rows_per_launch is not None:
            achieved = bpr * rows_per_launch / (avg_ms * 1e-3) / 1e9
            traffic = None
            tpath = next((q for q in (os.path.join(ROOT, "profiles", f) for f in ("traffic_r04.json", "traffic_r03.json", "traffic_r02.json", "traffic_r01.json")) if os.path.exists(q)), None)
            if tpath:
                traffic = json.load(open(tpath)).get(name)
            roofline = {"bound": "hbm", "kernel": name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                        "traffic": traffic, "launches_per_step": per_step, "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": int(bpr * rows_per_launch)}
        else:
            roofline = {"bound": "hbm", "kernel": name, "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                        "launches_per_step": per_step, "avg_launch_ms": round(avg_ms, 4)}
        roofline["kernel_ms_per_step"] = {k: round(v[1], 3) for k, v in sorted(breakdown.items(), key=lambda kv: -kv[1][1])}     # from the one fully bracketed untimed step
        if rank == 0:
            copy_gbs = measure_copy_bandwidth(torch)              # outside the timed region
            roofline["measured_copy_GBps"] = round(copy_gbs, 1)
            if roofline.get("achieved"):
                roofline["frac_of_measured_copy"] = round(roofline["achieved"] / copy_gbs, 4)
        roofline["host_syncs_per_step"] = host_syncs       # stream synchronisations by cause (counts the host must read back)
        roofline["traffic_source"] = "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this kernel, gfx950 x2 FETCH correction; a tracked file, not measured in this run)" % (os.path.relpath(tpath, ROOT) if tpath else None)
        if roofline.get("achieved") and rank == 0:
            assert roofline["achieved"] <= roofline["measured_copy_GBps"] * 1.25, "kernel bandwidth above the box's copy ceiling: wrong byte count"
    # Not a bandwidth: SURVEY 8(d)'s 39.4 B per input row counts every referenced column in full, while the plan (late materialisation, fused
    # selections) touches about 12 GB of the 30 GB at SF100.  Reported as a logical scan rate only.
    logical_scan = {"bytes_per_input_row": Q3_BYTES_PER_ROW, "GBps": round(Q3_BYTES_PER_ROW * rows_total / (ms_per_step * 1e-3) / 1e9, 1),
                    "note": "logical bytes of every referenced column / step time; NOT HBM traffic (filtered rows' payload columns are never read)"}

    # ---- N = 1: result check at full scale, the general (hash) paths of the same query, the other target plan shapes
    result_check = general = workloads = other_plans = dist_workloads = shuffled = None
    if world == 1:
        got = tpch.q3_checksum_result(last_out[0] or [])
        want = tpch.q3_checksum_torch(tensors)
        result_check = {"ok": got == want, "device": got, "torch": want, "what": "result groups, wrapping Int64 sums of l_orderkey and of unscaled revenue: the last timed step against plain torch ops over the same tensors"}
        assert got == want, f"Q3 result differs from the torch recomputation: {got} vs {want}"
        del want
        # the same plan with the clustered-key shortcuts off: joins build hash tables (radix-partitioned LDS tables for the 150 M-row orders build is not
        # taken -- dense domain -- so: membership bitmap + open addressing), the 3-key group-by goes through the hash table
        saved = {k: ctx.get_option(k) for k in ("join_rank_index", "group_run_detection")}
        for k in saved:
            ctx.set_option(k, 0)
        try:
            step()
            ctx.profile_select(None); ctx.profile_enable(True); ctx.profile_read(); step(); gp, _ = split_syncs(ctx.profile_read()); ctx.profile_enable(False)
            barrier(); t1 = time.perf_counter()
            for _ in range(3):
                step()
            barrier(); g_ms = (time.perf_counter() - t1) / 3 * 1e3
            assert tpch.q3_checksum_result(last_out[0] or []) == got, "general-path Q3 result differs"
        finally:
            for k, v in saved.items():
                ctx.set_option(k, v)
        general = {"options": {"join_rank_index": 0, "group_run_detection": 0}, "ms_per_step": round(g_ms, 3), "rows_per_s": round(rows_total / (g_ms * 1e-3), 1), "steps": 3, "result_check": "same checksums as the default paths",
                   "kernel_ms_per_step": {k: round(v[1], 3) for k, v in sorted(gp.items(), key=lambda kv: -kv[1][1])[:12]}}
        last_out[0] = None
        # the same query over the same rows in RANDOM order: every table permuted row-wise, so no key column is sorted and no join or group key arrives clustered -- what the
        # operators see behind a hash repartition, or on tables that were never written in key order.  Default options (nothing switched off): the builds are unique keys over a
        # dense domain in no order (rank index over unsorted keys, join.hip), the probes touch the bitmap at random, the group-by finds no runs (partitioned pre-aggregation).
        shuffled = None
        if not args.no_shuffled:
            gsh = torch.Generator(device="cuda"); gsh.manual_seed(20260311)
            t2 = {}
            for prefix in ("c_", "o_", "l_"):
                cols = [k for k in tensors if k.startswith(prefix)]
                perm = torch.randperm(tensors[cols[0]].shape[0], generator=gsh, device="cuda")
                for k in cols:          # (n, 2) Decimal128 tensors half by half: torch's row gather of a 2-D int64 tensor returned garbage at 240 M rows on this ROCm build
                    t2[k] = tensors[k][perm] if tensors[k].dim() == 1 else torch.stack([tensors[k][:, h][perm] for h in range(tensors[k].shape[1])], dim=1).contiguous()
                del perm
            tables2 = tpch.tables_from_torch(ctx, t2)
            template2 = tpch.q3_plan(tables2, batch_size=8192)

            def step_sh():
                out = [b for b in ops.with_fresh_state(template2).execute(0, tc)]
                ctx.synchronize(); last_out[0] = out
            step_sh()
            ctx.profile_select(None); ctx.profile_enable(True); ctx.profile_read(); step_sh(); sp_, ssync = split_syncs(ctx.profile_read()); ctx.profile_enable(False)
            barrier(); t1 = time.perf_counter()
            for _ in range(3):
                step_sh()
            barrier(); s_ms = (time.perf_counter() - t1) / 3 * 1e3
            same = tpch.q3_checksum_result(last_out[0] or []) == got
            assert same, "Q3 over the shuffled tables differs from Q3 over the clustered ones"
            shuffled = {"ms_per_step": round(s_ms, 3), "rows_per_s": round(rows_total / (s_ms * 1e-3), 1), "steps": 3, "result_check": "same checksums as over the clustered tables",
                        "what": "every table permuted row-wise (torch.randperm); default options",
                        "kernel_ms_per_step": {k: round(v[1], 3) for k, v in sorted(sp_.items(), key=lambda kv: -kv[1][1])[:12]}, "host_syncs_per_step": sum(ssync.values())}
            last_out[0] = None
            del t2, tables2, template2
            torch.cuda.empty_cache()
        if not args.no_workloads:
            import argparse as _ap
            import bench_workloads
            del template, tables, tensors
            torch.cuda.empty_cache()
            res = bench_workloads.run(_ap.Namespace(sf=args.sf, steps=3, warmup=2, only=args.workloads), ctx=ctx, emit=False)
            workloads = {r["workload"]: {k: r[k] for k in ("input_rows", "result_rows", "ms_per_step", "rows_per_s", "algorithmic_GBps", "frac_of_hbm_peak", "algorithmic_bytes_per_row", "roofline", "host_syncs_per_step", "host_syncs_by_cause", "result_check", "step_ms", "memory_GB") if k in r}
                         | {"kernel_ms_per_step": dict(list(r["kernel_ms_per_step"].items())[:8])} | {k: r[k] for k in r if k in ("cardinality", "build_rows", "probe_rows", "rows_per_build_key", "partitions", "exchange", "file_bytes", "decoded_bytes", "decoded_GBps", "file_GBps", "from_host_image_ms", "from_host_image_decoded_GBps", "row_groups",
                                                                                                                                                    "ms_per_step_median", "from_host_image_reads_ms", "host_image", "match_fraction", "global_table_ms_per_step", "hash_table_ms_per_step")} for r in res}
    else:
        # the other two distribution plans, after the timed region (every rank takes part: they hold collectives)
        other_plans = {}
        for name in ("shuffle", "broadcast", "colocated"):
            if name == args.plan:
                continue
            st2 = tpch.Q3ColocatedStaged(tables, batch_size=8192) if name == "colocated" else None
            def step2():
                wd.arm(True, "Q3 step (%s plan)" % name)
                plan = st2 if st2 is not None else PLANS[name](tables, batch_size=8192)
                with ctx.deferred_flags():
                    local = [b for b in plan.execute(0, tc)]
                    mine = ops.concat_batches(local[0].schema, local) if local else None
                    gathered = exchange.gather_batches(ctx, None, mine, 0, names=Q3_OUTPUT)
                if rank == 0 and gathered.num_rows:
                    final_slot.replace([[gathered]])
                    [b for b in ops.with_fresh_state(final_plan).execute(0, tc)]
                ctx.synchronize()
                wd.arm(False)
            step2(); barrier(); t1 = time.perf_counter()
            for _ in range(args.steps):
                step2()
            barrier(); el = time.perf_counter() - t1
            t = torch.tensor([el], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            other_plans[name] = {"ms_per_step": round(float(t.item()) / args.steps * 1e3, 3), "rows_per_s": round(rows_total * args.steps / float(t.item()), 1)}

        # BASELINE configs 4 and 5 at N > 1: TPC-H Q5 as the fully partitioned plan (hash repartition + all-to-all under every join and the aggregate) and the
        # ClickBench Q28 shape (partial states of a high-cardinality dictionary-key group-by shuffled on the key)
        if not args.no_workloads:
            from dfgpu import dist_workloads as dw
            del tables, staged, st2
            torch.cuda.empty_cache()
            dev = "cuda" if args.backend == "nccl" else "cpu"
            exchange.TIMING = True
            dist_workloads = {}

            def run_dist(name, build, outputs, sort_keys, fetch, rows_local, checksum):
                def stepd():
                    wd.arm(True, "workload %s step" % name)
                    plan = build()
                    with ctx.deferred_flags():
                        local = [b for b in plan.execute(0, tc)]
                        mine = ops.concat_batches(local[0].schema, local) if local else None
                        gathered = exchange.gather_batches(ctx, None, mine, 0, names=outputs)
                    res = []
                    if rank == 0 and gathered.num_rows:
                        res = [b for b in ops.SortExec(sort_keys, ops.MemoryExec([[gathered]], gathered.schema), fetch=fetch).execute(0, tc)]
                    ctx.synchronize()
                    wd.arm(False)
                    return plan, res
                for _ in range(max(1, args.warmup)):
                    stepd()
                barrier(); t1 = time.perf_counter()
                for _ in range(args.steps):
                    plan, res = stepd()
                barrier(); el = time.perf_counter() - t1
                tt = torch.tensor([el], dtype=torch.float64, device=dev); dist.all_reduce(tt, op=dist.ReduceOp.MAX); el = float(tt.item())
                nodes = dw.shuffle_nodes(plan)
                sent = sum(nd.bytes_sent for nd in nodes); ex_ms = sum(nd.exchange_ms() for nd in nodes)
                agg = torch.tensor([rows_local, sent], dtype=torch.int64, device=dev); dist.all_reduce(agg)
                mx = torch.tensor([ex_ms], dtype=torch.float64, device=dev); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
                rows_all, sent_all = int(agg[0].item()), int(agg[1].item())
                ms = el / args.steps * 1e3
                dist_workloads[name] = {"ms_per_step": round(ms, 3), "input_rows": rows_all, "rows_per_s": round(rows_all / (ms * 1e-3), 1), "result_rows": sum(b.num_rows for b in res),
                                        "exchanges_per_step": len(nodes), "all_to_all_bytes_per_step": sent_all, "all_to_all_bytes_per_rank": sent_all // world,
                                        "exchange_ms_per_step_max_rank": round(float(mx.item()), 3),
                                        "GBps_per_link": round(sent_all / world / max(1, world - 1) / max(1e-9, float(mx.item()) * 1e-3) / 1e9, 2),
                                        "GBps_per_link_note": "bytes one rank sends / (world - 1) peers / the time its exchanges took, partition gathers and waiting for peers included (last timed step)",
                                        "result_checksum": checksum(res) if rank == 0 else None}

            tt5 = dw.q5_tensors(args.sf, rank, world)
            t5 = dw.q5_tables(ctx, tt5, rank)
            rows5 = sum(t5[k].num_rows for k in ("customer", "orders", "lineitem", "supplier", "nation", "region"))
            q5_sum = lambda res: [[r["n_name"], str(r["revenue"])] for b in res for r in b.to_arrow().to_pylist()]
            run_dist("q5", lambda: dw.q5_plan(t5, batch_size=8192, native=args.native_exchange), dw.Q5_OUTPUT, [ops.PhysicalSortExpr(C_("revenue", 1), True, True)], None, rows5, q5_sum)
            del tt5, t5
            torch.cuda.empty_cache()
            n_hits, card = int(1_000_000 * args.sf), max(10, int(1_000_000 * min(1.0, args.sf / 100.0)))
            ids, length, wcol = dw.clickbench_tensors(n_hits, card, rank, world)
            hits = dw.clickbench_batch(ctx, ids, length, wcol, card)
            cb_sum = lambda res: [[r["k"], r["c"], r["m"]] for b in res for r in b.to_arrow().to_pylist()][:5]
            run_dist(f"clickbench_uniform_{card}", lambda: dw.clickbench_plan(hits, batch_size=8192), dw.CLICKBENCH_OUTPUT,
                     [ops.PhysicalSortExpr(C_("l", 1), True, True), ops.PhysicalSortExpr(C_("k", 0), False, False)], 25, ids.numel(), cb_sum)
            del ids, length, wcol, hits
            exchange.TIMING = False

    # ---- CPU baseline: the oracle's restatement of the same plan on a bounded sample (rank 0, N = 1 only)
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.cpu_sf > 0:
        from oracle import pyoracle as po
        import numpy as np
        cpu_sf = min(args.cpu_sf, args.sf)
        small = tpch.gen_device(ctx, cpu_sf, seed=tpch.SEED + 1)
        host = tpch.tables_to_host(small)
        del small
        torch.cuda.empty_cache()
        # a one-GPU box is given a 16-core share of the host (os.cpu_count() reports the whole machine)
        cores = max(1, min(len(os.sched_getaffinity(0)), 16))
        n_in = tpch.total_input_rows(host)
        seg, times = tpch.SEGMENTS.index(tpch.Q3_SEGMENT), []
        po.tpch_q3(host, seg, tpch.Q3_DATE, cores, 8192)                  # warm-up (page faults)
        for _ in range(5):
            t1 = time.perf_counter()
            res = po.tpch_q3(host, seg, tpch.Q3_DATE, cores, 8192)
            times.append(time.perf_counter() - t1)
        med = statistics.median(times)
        try:
            model = next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "unknown")
        except OSError:
            model = "unknown"
        cpu_baseline = {"value": round(n_in / med, 1), "unit": "rows/s", "cores": cores, "cpu_model": model, "kind": "port",
                        "sample": f"oracle/dfo_tpch.c restatement of DataFusion 36 CPU operators (Q3 plan, target_partitions={cores}, batch_size=8192) on synthetic SF{cpu_sf:g} "
                                  f"({n_in} input rows, {len(res['l_orderkey'])} result rows); 1 warm-up + 5 runs ({sum(times):.2f}s timed), median {med:.3f}s, min {min(times):.3f}s",
                        "min_value": round(n_in / min(times), 1)}
        # a second, INDEPENDENT CPU number (not the reference, not the oracle): the same query through pyarrow's Acero engine
        try:
            cpu_baseline["acero"] = acero_q3(host, tpch, n_in)
        except Exception as e:  # noqa: an optional leg must not fail the bench
            cpu_baseline["acero"] = {"error": repr(e)[:200]}

    if rank == 0:
        detail = {"metric": METRIC, "value": round(value, 1), "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                  "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int64",
                  "data": "synthetic", "config": {"workload": f"TPC-H SF{args.sf:g} Q3 (3-way hash join + group-by SUM + sort), int64 keys, Decimal128(15,2) money, resident in HBM",
                                                  "input_rows": rows_total, "result_rows": result_rows[0], "parallelism": PARALLELISM[args.plan] % world if world > 1 else "1 GPU"},
                  "roofline": roofline, "cpu_baseline": cpu_baseline, "logical_scan_rate": logical_scan, "ranks_seen": ranks_seen}
        detail["config"]["paths"] = "joins: membership bitmap + rank index (dense TPC-H keys); group-by: run numbering (clustered input); hash paths: q3_general_paths"
        if world == 1:
            detail["result_check"] = result_check; detail["q3_general_paths"] = general
            if shuffled is not None:
                detail["q3_shuffled_inputs"] = shuffled
            if workloads is not None:
                detail["workloads"] = workloads
        else:
            detail["plans"] = other_plans
            if dist_workloads:
                detail["workloads"] = dist_workloads
            detail["config"]["plan"] = args.plan
            detail["config"]["exchange_path"] = ("dfgpu_exchange (C ABI, RCCL grouped send / recv inside libdfgpu.so)" if args.native_exchange else "exchange.py over torch.distributed (%s): one metadata all-gather + one all-to-all(v) per exchange" % args.backend) + \
                "; Q3's own plans always use the torch.distributed path, --native-exchange switches the Q5 / ClickBench workloads"
        # everything measured goes to a side file + stderr; the LAST stdout line is the compact object the driver parses
        text = json.dumps(detail)
        try:
            with open(args.detail or os.path.join(ROOT, "bench_detail.json"), "w") as f:
                f.write(json.dumps(detail, indent=1) + "\n")
        except OSError as e:
            sys.stderr.write("bench.py: bench_detail.json not written: %r\n" % (e,))
        sys.stderr.write(text + "\n"); sys.stderr.flush()
        sys.stdout.flush()
        print(compact_line(detail), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
