import os, sys, faulthandler
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
faulthandler.dump_traceback_later(60, exit=True)
import torch, torch.distributed as dist
import pyarrow as pa
import dfgpu
from dfgpu import exchange, physical_plan as ops, tpch
from dfgpu.exchange import ShuffleExec
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29613")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
ctx = dfgpu.Context(0, stream=torch.cuda.current_stream().cuda_stream)
sf = float(os.environ.get("SF", "0.05")); bs = int(os.environ.get("BS", str(1 << 30)))
host = tpch.gen_host(sf)
tables = tpch.upload(ctx, host)
tc = ops.TaskContext(ctx, batch_size=bs)
C, L, B = ops.Column, ops.Literal, ops.BinaryExpr
rows = lambda p: sum(b.num_rows for b in p.execute(0, tc))
cust = ops.MemoryExec([[tables["customer"]]], None); orders = ops.MemoryExec([[tables["orders"]]], None); line = ops.MemoryExec([[tables["lineitem"]]], None)
cb = lambda p: ops.CoalesceBatchesExec(p, 8192)
f_c = cb(ops.FilterExec(B(C("c_mktsegment", 1), "=", L("BUILDING", pa.utf8())), cust))
p_c = ops.ProjectionExec([(C("c_custkey", 0), "c_custkey")], f_c)
s_c = cb(ShuffleExec(p_c, [C("c_custkey", 0)]))
print("s_c", rows(s_c), flush=True)
f_o = cb(ops.FilterExec(B(C("o_orderdate", 2), "<", L(tpch.Q3_DATE, pa.date32())), orders))
s_o = cb(ShuffleExec(f_o, [C("o_custkey", 1)]))
print("s_o", rows(s_o), flush=True)
j1 = cb(ops.HashJoinExec(s_c, s_o, [(C("c_custkey", 0), C("o_custkey", 1))], None, "Inner", "Partitioned"))
print("j1", rows(j1), flush=True)
p_j1 = ops.ProjectionExec([(C("o_orderkey", 1), "o_orderkey"), (C("o_orderdate", 3), "o_orderdate"), (C("o_shippriority", 4), "o_shippriority")], j1)
s_j1 = cb(ShuffleExec(p_j1, [C("o_orderkey", 0)]))
print("s_j1", rows(s_j1), flush=True)
f_l = cb(ops.FilterExec(B(C("l_shipdate", 3), ">", L(tpch.Q3_DATE, pa.date32())), line))
p_l = ops.ProjectionExec([(C("l_orderkey", 0), "l_orderkey"), (C("l_extendedprice", 1), "l_extendedprice"), (C("l_discount", 2), "l_discount")], f_l)
s_l = cb(ShuffleExec(p_l, [C("l_orderkey", 0)]))
print("s_l", rows(s_l), flush=True)
j2 = cb(ops.HashJoinExec(s_j1, s_l, [(C("o_orderkey", 0), C("l_orderkey", 0))], None, "Inner", "Partitioned"))
print("j2", rows(j2), flush=True)
dist.destroy_process_group()
