"""datafusion-upstream_amd -- MI355X (gfx950) execution path for DataFusion 36's data-parallel physical operators.

Layers:
  csrc/        hand-written HIP kernels + the C ABI of include/dfgpu.h  -> libdfgpu.so (built in-tree)
  capi.py      ctypes binding of exactly those symbols
  device.py    Context / Array / JoinTable / GroupValues / GroupsAccumulator wrappers
  csrc/exec/   C++ host layer: ExecutionPlan / PhysicalExpr mirror (FilterExec, ProjectionExec, HashJoinExec, AggregateExec,
               SortExec, RepartitionExec, CoalesceBatchesExec ...) over the kernel ABI; C ABI in include/dfgpu_exec.h
  physical_plan.py  typed Python builders over that C++ layer (same class names as the reference operators)
  operators.py shared plain dataclasses (Field, Schema, TaskContext)
  exchange.py  RCCL all-to-all(v) of partitioned column buffers via torch.distributed (multi-GPU shuffle)
  tpch.py      synthetic TPC-H-shaped columns + the Q3 physical plan of the reference

Import as `import dfgpu` (alias module at the repo root) or importlib.import_module("datafusion-upstream_amd").
The package never falls back to a CPU implementation: without libdfgpu.so or a HIP device it raises.
"""
from . import capi
from .capi import DfgpuError, load_library
from .device import Array, Context, GroupValues, GroupsAccumulator, JoinTable, agg_preaggregate, join_adjust_indices
from . import operators
from . import physical_plan

__all__ = ["capi", "DfgpuError", "load_library", "Array", "Context", "GroupValues", "GroupsAccumulator", "JoinTable",
           "join_adjust_indices", "operators", "physical_plan"]
