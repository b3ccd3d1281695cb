"""-m gpu: the one-pass RepartitionExec kernel path (csrc/radix_partition.h through dfgpu_partition_columns) against the CPU oracle's
BatchPartitioner restatement (repartition/mod.rs:148-221): rows go to hash % n, input order is kept inside a destination, the columns
written in the same pass equal `take(column, indices)`; row counts are conserved (repartition/mod.rs:952-1031)."""
import numpy as np
import pyarrow as pa
import pytest

from oracle import pyoracle as po
from test_gpu_join import keycols

pytestmark = pytest.mark.gpu
RNG = np.random.default_rng(808)


@pytest.mark.parametrize("nparts", [1, 2, 3, 8, 13, 64, 256])
@pytest.mark.parametrize("n", [0, 1, 4095, 4096, 4097, 100003])
def test_partition_columns_equal_take_through_oracle_indices(ctx, nparts, n):
    keys = keycols(["int64", "int32"], n, 0.1, 5000)
    cols = [pa.array(RNG.integers(-2**60, 2**60, n)), pa.array(RNG.integers(0, 100, n).astype(np.int32)), pa.array(RNG.random(n)),
            pa.array([None if i % 7 == 0 else i for i in range(n)], type=pa.int64()), pa.array([f"s{i % 13}" for i in range(n)], type=pa.utf8()),
            pa.array(RNG.integers(0, 2**31, n).astype(np.int16))]
    import decimal
    cols.append(pa.array([decimal.Decimal(int(v)).scaleb(-2) for v in RNG.integers(-10**12, 10**12, n)], type=pa.decimal128(15, 2)))
    dk = [ctx.from_arrow(k) for k in keys]
    dc = [ctx.from_arrow(c) for c in cols]
    dc[2] = None                                         # a column the caller keeps lazy
    outs, idx, counts = ctx.partition_columns(dk, nparts, dc)
    oidx, ocounts = po.hash_partition(keys, nparts)
    assert counts == ocounts.tolist() and sum(counts) == n
    assert np.array_equal(idx.to_numpy(), oidx)
    direct = [c for c in (0, 1, 3, 5, 6) if cols[c].null_count == 0]          # fixed-width, no NULLs: written in the same pass
    for c, o in enumerate(outs):
        if c in direct:
            assert o is not None and o.to_arrow().equals(cols[c].take(pa.array(oidx)))
        else:
            assert o is None                             # nullable / Utf8 / lazy: through the indices


@pytest.mark.parametrize("wide", [False, True], ids=["narrow-columns", "with-decimal128"])
@pytest.mark.parametrize("nparts", [8, 64, 200])
def test_partition_columns_with_fused_selection(ctx, nparts, wide):
    """Selection fused into the partition pass, through both scatters: the LDS-free one (<= 16 partitions, or <= 256 when a 16-byte column moves) and the LDS-staged one."""
    import decimal
    n = 50000
    key = pa.array(RNG.integers(0, 10**9, n)); v = pa.array(RNG.integers(0, 10**6, n)); mask = RNG.random(n) < 0.3
    extra = [pa.array([decimal.Decimal(int(x)).scaleb(-2) for x in RNG.integers(-10**12, 10**12, n)], type=pa.decimal128(15, 2))] if wide else [pa.array(RNG.integers(0, 100, n).astype(np.int8))]
    outs, idx, counts = ctx.partition_columns([ctx.from_arrow(key)], nparts, [ctx.from_arrow(v), ctx.from_arrow(extra[0])], mask=ctx.from_arrow(pa.array(mask)))
    sel = np.flatnonzero(mask)
    oidx, ocounts = po.hash_partition([key.take(pa.array(sel))], nparts)
    assert counts == ocounts.tolist()
    assert np.array_equal(idx.to_numpy(), sel[oidx])      # original row numbers of the selected rows
    assert outs[0].to_arrow().equals(v.take(pa.array(sel[oidx])))
    assert outs[1].to_arrow().equals(extra[0].take(pa.array(sel[oidx])))


def test_repartition_exec_uses_the_one_pass_kernels_and_conserves_rows(ctx, task_ctx):
    from dfgpu import physical_plan as ops
    n = 300000
    t = pa.table({"k": pa.array(RNG.integers(0, 10**7, n)), "a": pa.array(RNG.integers(0, 100, n).astype(np.int32)), "s": pa.array([f"x{i % 17}" for i in range(n)])})
    b = ops.batch_from_arrow(ctx, t)
    plan = ops.RepartitionExec(ops.MemoryExec([[b]], b.schema), ops.Partitioning.Hash([ops.Column("k", 0)], 8))
    ctx.profile_select(None); ctx.profile_enable(True); ctx.profile_read()
    parts = [[bb.materialize() for bb in plan.execute(p, task_ctx)] for p in range(8)]
    ran = set(ctx.profile_read()); ctx.profile_enable(False)
    assert "rp_scatter" in ran and "radix_pass" not in ran
    oidx, ocounts = po.hash_partition([t["k"].combine_chunks()], 8)
    off = 0
    for p in range(8):
        rows = sum(bb.num_rows for bb in parts[p]); assert rows == ocounts[p]
        if rows:
            got = pa.Table.from_arrays([pa.concat_arrays([bb.columns[i].to_arrow() for bb in parts[p]]) for i in range(3)], names=["k", "a", "s"])
            assert got.equals(t.take(pa.array(oidx[off:off + rows])).combine_chunks())
        off += rows
