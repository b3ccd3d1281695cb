"""-m gpu: size-independent properties at BASELINE-scale row counts (hundreds of millions of rows, beyond 2^28 where
a capped launch grid would silently skip rows): filter counts, take round trip, probe of a dense key range, group
interning of a known cardinality, checksum conservation through a hash partition."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
N = 300_000_000            # > 2^28 = 268,435,456


@pytest.fixture(scope="module")
def big(ctx):
    import torch
    import dfgpu
    t = torch.arange(N, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()          # the test ctx runs on its own stream: torch's producer kernel must have finished
    return t, ctx.wrap_tensor(t, dfgpu.capi.INT64)


def test_compare_filter_counts_at_scale(ctx, big):
    import pyarrow as pa
    import dfgpu
    t, col = big
    cut = ctx.from_arrow(pa.array([N - 1000], type=pa.int64()))
    mask = ctx.binary(dfgpu.capi.OP_GTEQ, col, cut, False, True)          # last 1000 rows only: they live beyond 2^28
    sel = ctx.mask_to_indices(mask)
    assert len(sel) == 1000
    got = sel.to_numpy()
    assert got[0] == N - 1000 and got[-1] == N - 1
    back = ctx.take(col, sel).to_numpy()
    assert np.array_equal(back, np.arange(N - 1000, N))


def test_probe_dense_keys_at_scale(ctx, big):
    import dfgpu
    import torch
    t, col = big
    bt = torch.arange(0, N, 1000, dtype=torch.int64, device="cuda")          # 300k build keys, every 1000th probe key matches
    torch.cuda.synchronize()
    table = dfgpu.JoinTable(ctx, [ctx.wrap_tensor(bt, dfgpu.capi.INT64)])
    bidx, pidx = table.probe([col])
    assert len(bidx) == len(bt)
    p = pidx.to_numpy().astype(np.int64)
    b = bidx.to_numpy().astype(np.int64)
    assert np.array_equal(p, np.arange(0, N, 1000)) and np.array_equal(b, np.arange(len(bt)))      # probe order preserved end to end


def test_group_cardinality_and_sum_at_scale(ctx, big):
    import dfgpu
    import torch
    t, col = big
    keys = (t % 1_000_003).contiguous()
    torch.cuda.synchronize()
    kc = ctx.wrap_tensor(keys, dfgpu.capi.INT64)
    gv = dfgpu.GroupValues(ctx, 1)
    gids = gv.intern([kc])
    assert len(gv) == 1_000_003
    acc = dfgpu.GroupsAccumulator(ctx, dfgpu.capi.AGG_SUM, dfgpu.capi.INT64)
    acc.update_batch(col, gids, None, len(gv))
    sums = acc.evaluate().to_numpy()
    assert int(sums.sum()) == N * (N - 1) // 2                               # checksum of checksums
    emitted = gv.emit()[0].to_numpy()
    assert np.array_equal(emitted, np.arange(1_000_003))                    # first-seen order == key order here


def test_hash_partition_conserves_rows_at_scale(ctx, big):
    t, col = big
    idx, counts = ctx.hash_partition([col], 8)
    assert sum(counts) == N and min(counts) > N // 8 * 0.99
    import torch
    # every row id appears exactly once: sum of indices == N(N-1)/2
    d = idx.describe()
    from dfgpu.exchange import _DevicePtr
    it = torch.as_tensor(_DevicePtr(d.values, N * 4, idx), device="cuda").view(torch.int32)
    assert int(it.to(torch.int64).bitwise_and(0xFFFFFFFF).sum().item()) == N * (N - 1) // 2


def test_clustered_group_runs_and_run_combining_sums_at_scale(ctx):
    """Q18's sub-aggregate shape at 300 M rows: sorted keys with 4 rows per key -> run numbering (no hash table), sums through the
    run-combining accumulator.  Group count, first-seen key order and a checksum of checksums must hold; the hash path (option off on a
    10 M-row prefix) gives the same ids."""
    import dfgpu
    import torch
    keys = (torch.arange(N, dtype=torch.int64, device="cuda") // 4) * 3 + 7
    vals = torch.arange(N, dtype=torch.int64, device="cuda") % 1000
    torch.cuda.synchronize()
    kc, vc = ctx.wrap_tensor(keys, dfgpu.capi.INT64), ctx.wrap_tensor(vals, dfgpu.capi.INT64)
    gv = dfgpu.GroupValues(ctx, 1)
    gids = gv.intern([kc])
    assert len(gv) == N // 4
    g = gids.to_numpy()
    assert g[0] == 0 and g[-1] == N // 4 - 1 and np.array_equal(g[::4][:1000], np.arange(1000))
    acc = dfgpu.GroupsAccumulator(ctx, dfgpu.capi.AGG_SUM, dfgpu.capi.INT64)
    acc.update_batch(vc, gids, None, len(gv))
    sums = acc.evaluate().to_numpy()
    assert int(sums.sum()) == int(vals.sum().item())
    want0 = int(vals[:4].sum().item())
    assert sums[0] == want0 and sums[250] == int(vals[1000:1004].sum().item())
    ctx.set_option("group_run_detection", 0)
    try:
        gv2 = dfgpu.GroupValues(ctx, 1)
        g2 = gv2.intern([kc.slice(0, 10_000_000)]).to_numpy()
    finally:
        ctx.set_option("group_run_detection", 1)
    assert np.array_equal(g2, g[:10_000_000])


def test_rank_index_with_runs_at_scale(ctx):
    """A sorted foreign key with repeats as BUILD side (Q18's semi-join build, here 300 M rows, 3 per key): every probe key emits its
    contiguous run in build order; pair count and a sum over the emitted build rows are closed forms."""
    import dfgpu
    import torch
    build = (torch.arange(N, dtype=torch.int64, device="cuda") // 3) * 2
    probe = torch.arange(0, 2_000_000, dtype=torch.int64, device="cuda")              # even keys < 2 M match 3 rows each, odd keys nothing
    torch.cuda.synchronize()
    table = dfgpu.JoinTable(ctx, [ctx.wrap_tensor(build, dfgpu.capi.INT64)])
    bidx, pidx = table.probe([ctx.wrap_tensor(probe, dfgpu.capi.INT64)])
    assert len(bidx) == 3 * 1_000_000
    b, p = bidx.to_numpy().astype(np.int64), pidx.to_numpy().astype(np.int64)
    assert np.array_equal(b, np.arange(3_000_000)) and np.array_equal(p, np.repeat(np.arange(0, 2_000_000, 2), 3))


def test_dense_dictionary_groups_and_shared_pass_accumulators_at_scale(ctx):
    """Q1's shape at 300 M rows: two Int8 dictionary key columns (dense composite map) and four Float64 / Int64 accumulators handed over
    together (shared passes).  Per-group counts must equal torch.bincount, sums the closed forms."""
    import pyarrow as pa
    import dfgpu
    import torch
    from bench_workloads import wrap_dict
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    c1 = torch.randint(0, 3, (N,), generator=g, device="cuda", dtype=torch.int8)
    c2 = torch.randint(0, 2, (N,), generator=g, device="cuda", dtype=torch.int8)
    ones = torch.ones(N, dtype=torch.float64, device="cuda")
    iv = torch.arange(N, dtype=torch.int64, device="cuda") % 7
    torch.cuda.synchronize()
    d1, d2 = ctx.from_arrow(pa.array(["A", "N", "R"])), ctx.from_arrow(pa.array(["F", "O"]))
    k1, k2 = wrap_dict(ctx, dfgpu.capi, c1, d1, dfgpu.capi.INT8), wrap_dict(ctx, dfgpu.capi, c2, d2, dfgpu.capi.INT8)
    gv = dfgpu.GroupValues(ctx, 2)
    gids = gv.intern([k1, k2])
    assert len(gv) == 6
    fo, io = ctx.wrap_tensor(ones, dfgpu.capi.FLOAT64), ctx.wrap_tensor(iv, dfgpu.capi.INT64)
    accs = [dfgpu.GroupsAccumulator(ctx, dfgpu.capi.AGG_SUM, dfgpu.capi.FLOAT64), dfgpu.GroupsAccumulator(ctx, dfgpu.capi.AGG_AVG, dfgpu.capi.FLOAT64),
            dfgpu.GroupsAccumulator(ctx, dfgpu.capi.AGG_SUM, dfgpu.capi.INT64), dfgpu.GroupsAccumulator(ctx, dfgpu.capi.AGG_COUNT, dfgpu.capi.INT64)]
    dfgpu.GroupsAccumulator.update_batch_multi(ctx, accs, [fo, fo, io, None], [None] * 4, gids, 6)
    ids = torch.as_tensor(gids.to_numpy().astype(np.int64))
    counts = torch.bincount(ids, minlength=6).numpy()
    assert np.array_equal(accs[3].evaluate().to_numpy(), counts) and counts.sum() == N
    assert np.array_equal(accs[0].evaluate().to_numpy(), counts.astype(np.float64))          # sums of ones are exact in f64
    assert np.allclose(accs[1].evaluate().to_numpy(), 1.0)
    assert int(accs[2].evaluate().to_numpy().sum()) == int(iv.sum().item())
    comp = (c1.to(torch.int64) * 2 + c2.to(torch.int64)).cpu()
    first_seen = []
    for v in comp[:4096].tolist():
        if v not in first_seen:
            first_seen.append(v)
    em = [c.to_arrow().to_pylist() for c in gv.emit()]
    assert [("ANR".index(a) * 2 + "FO".index(b)) for a, b in zip(*em)] == first_seen          # ids in first-seen order


def test_topk_select_at_scale(ctx):
    """SortExec fetch = 100 over 300 M Int64 keys (radix select, then a sort of the few selected rows) == the 100 smallest, in order."""
    import dfgpu
    import torch
    t = (torch.arange(N, dtype=torch.int64, device="cuda") * 2654435761) % 1_000_000_007
    torch.cuda.synchronize()
    col = ctx.wrap_tensor(t, dfgpu.capi.INT64)
    top = ctx.sort_to_indices([col], [False], [True], fetch=100).to_numpy().astype(np.int64)
    vals = t[torch.as_tensor(top, device="cuda")].cpu().numpy()
    want = torch.topk(t, 100, largest=False, sorted=True).values.cpu().numpy()
    assert np.array_equal(vals, want)
