"""COUNT(DISTINCT) and MIN / MAX over Utf8 -- served by the plan layer's AggregateExec out of other kernels (csrc/exec/exec.cpp StringMinMax, CountDistinct).
CPU: the oracle's plain-Python models pinned on the reference's known answers (group_by.slt dictionary tables :4583-4882 `count(distinct column2)`,
aggregate.slt:3128-3158 distinct_count_string_table, min_max.rs:1298-1318 min_utf8 / max_utf8).  -m gpu: the operator against the same answers and against
the models on random inputs, in Single mode and (string MIN / MAX) through Partial -> FinalPartitioned with several batches, NULLs, a FILTER clause, no GROUP BY."""
import numpy as np
import pyarrow as pa
import pytest

from oracle import pyoracle as po

T = {"c1": [1, 2, 2, 3, 3, 3], "c2": ["a", "b", "b", "c", "c", "c"], "c3": ["longstringtest_a", "longstringtest_b1", "longstringtest_b2", "longstringtest_c1", "longstringtest_c2", "longstringtest_c3"],
     "c4": ["台灣", "日本", "中國", "美國", "歐洲", "韓國"]}
DICT_ROWS = ([1, 2, 2, 4, 1, 1], ["A", "B", "A", "A", "C", "A"])


def test_models_on_reference_answers():
    z = np.zeros(6, dtype=np.int64)
    assert [po.count_distinct(pa.array(T[c]), z, 1)[0].as_py() for c in ("c1", "c2", "c3", "c4")] == [3, 3, 6, 6]                  # aggregate.slt:3144-3147
    g = np.array([0, 1, 1, 2, 2, 2])
    assert [po.count_distinct(pa.array(T[c]), g, 3).to_pylist() for c in ("c1", "c2", "c3", "c4")] == [[1, 1, 1], [1, 1, 1], [1, 2, 3], [1, 2, 3]]    # :3150-3155
    k, v = DICT_ROWS
    gid = np.array([{1: 0, 2: 1, 4: 2}[x] for x in k])
    assert po.count_distinct(pa.array(v), gid, 3).to_pylist() == [2, 2, 1]                                                          # group_by.slt:4608-4614
    a = pa.array(["d", "a", "c", "b"])
    assert po.string_min_max(a, np.zeros(4, dtype=np.int64), 1, False)[0].as_py() == "a" and po.string_min_max(a, np.zeros(4, dtype=np.int64), 1, True)[0].as_py() == "d"   # min_max.rs:1298-1318


def agg_plan(ctx, batches, keys, aggs, mode="Single"):
    from dfgpu import capi, physical_plan as ops
    bs = [ops.batch_from_arrow(ctx, t) for t in batches]
    src = ops.MemoryExec([bs], bs[0].schema)
    names = batches[0].column_names
    C, F = ops.Column, ops.Field
    gb = [(C(k, names.index(k)), k) for k in keys]

    def mk(srcnames):
        out = []
        for fun, col, typ in aggs:
            out.append(ops.AggregateFunctionExpr(fun, C(col, srcnames.index(col)) if col else None, f"{fun}({col})", input_field=F(col or "x", typ)))
        return out
    if mode == "Single":
        plan = ops.AggregateExec("Single", gb, mk(names), src)
    else:
        partial = ops.AggregateExec("Partial", gb, mk(names), src)
        fin_gb = [(C(k, i), k) for i, k in enumerate(keys)]
        plan = ops.AggregateExec("FinalPartitioned", fin_gb, mk(names), partial)
    out = list(plan.execute(0, ops.TaskContext(ctx, 8192)))
    t = pa.concat_tables([b.to_arrow() for b in out])
    return t


@pytest.mark.gpu
def test_device_reference_answers(ctx):
    from dfgpu import capi
    t = pa.table({"c1": pa.array(T["c1"], type=pa.int64()), "c2": pa.array(T["c2"]), "c3": pa.array(T["c3"]), "c4": pa.array(T["c4"])})
    cd = [("COUNT DISTINCT", "c1", capi.INT64), ("COUNT DISTINCT", "c2", capi.UTF8), ("COUNT DISTINCT", "c3", capi.UTF8), ("COUNT DISTINCT", "c4", capi.UTF8)]
    out = agg_plan(ctx, [t], [], cd)
    assert [out.column(i)[0].as_py() for i in range(4)] == [3, 3, 6, 6]
    out = agg_plan(ctx, [t], ["c1"], cd)
    assert sorted(tuple(r.values())[1:] for r in out.to_pylist()) == [(1, 1, 1, 1), (1, 1, 2, 2), (1, 1, 3, 3)]
    k, v = DICT_ROWS
    for kt in (pa.int8(), pa.uint16(), pa.int32(), pa.uint64()):
        enc = pa.array(v).dictionary_encode(); d = pa.DictionaryArray.from_arrays(enc.indices.cast(kt), enc.dictionary)
        out = agg_plan(ctx, [pa.table({"column1": pa.array(k, type=pa.int64()), "column2": d})], ["column1"], [("COUNT DISTINCT", "column2", capi.UTF8)])
        assert sorted(tuple(r.values()) for r in out.to_pylist()) == [(1, 2), (2, 2), (4, 1)]
    a = pa.table({"s": pa.array(["d", "a", "c", "b"])})
    out = agg_plan(ctx, [a], [], [("MIN", "s", capi.UTF8), ("MAX", "s", capi.UTF8)])
    assert (out.column(0)[0].as_py(), out.column(1)[0].as_py()) == ("a", "d")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["Single", "PartialFinal"])
def test_device_equals_models_on_random_batches(ctx, mode):
    from dfgpu import capi
    rng = np.random.default_rng(5)
    batches, n = [], 4000
    for b in range(3):
        words = np.array([f"w{v:03d}" + "é" * (v % 3) for v in range(60)])
        batches.append(pa.table({"k": pa.array(rng.integers(0, 40, n).astype(np.int64), mask=rng.random(n) < 0.05),
                                 "s": pa.array(words[rng.integers(0, 60, n)], mask=rng.random(n) < 0.3),
                                 "v": pa.array(rng.integers(0, 25, n).astype(np.int64), mask=rng.random(n) < 0.2)}))
    aggs = [("MIN", "s", capi.UTF8), ("MAX", "s", capi.UTF8), ("COUNT", "v", capi.INT64), ("COUNT DISTINCT", "v", capi.INT64), ("COUNT DISTINCT", "s", capi.UTF8)]
    out = agg_plan(ctx, batches, ["k"], aggs, mode)
    allt = pa.concat_tables(batches)
    og = po.Groups([pa.int64()]); gids = og.intern([allt["k"].combine_chunks()]); total = len(og)
    assert out.column(0).combine_chunks().equals(og.emit()[0])                       # groups in first-seen order
    s, v = allt["s"].combine_chunks(), allt["v"].combine_chunks()
    assert out.column(1).combine_chunks().equals(po.string_min_max(s, gids, total, False))
    assert out.column(2).combine_chunks().equals(po.string_min_max(s, gids, total, True))
    assert out.column(4).combine_chunks().equals(po.count_distinct(v, gids, total))        # PartialFinal: through the List state of the Partial stage
    assert out.column(5).combine_chunks().equals(po.count_distinct(s, gids, total))        # PartialFinal: a List of strings per group ((length, bytes) in the Utf8 layout)


def string_lists_of(col: pa.Array):
    """a list-of-strings column as it travels on the device (every string as u32 length + bytes inside a Utf8-layout row) -> Python lists of str"""
    col = col.combine_chunks() if isinstance(col, pa.ChunkedArray) else col
    off = np.frombuffer(col.buffers()[1], dtype=np.int32)[col.offset: col.offset + len(col) + 1]
    data = bytes(col.buffers()[2]) if col.buffers()[2] is not None else b""
    out = []
    for i in range(len(col)):
        at, row = int(off[i]), []
        while at < off[i + 1]:
            ln = int.from_bytes(data[at:at + 4], "little"); row.append(data[at + 4:at + 4 + ln].decode()); at += 4 + ln
        assert at == off[i + 1]
        out.append(row)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("parts", [1, 3])
def test_count_distinct_over_strings_partial_state_and_final(ctx, parts):
    """COUNT(DISTINCT s) over a Utf8 argument in Partial -> RepartitionExec -> FinalPartitioned (count_distinct/bytes.rs:47-75: the state is one List of the group's distinct
    strings; merge_batch inserts every string of every incoming list): the Partial stage's lists hold each group's distinct strings in first-seen order -- empty strings,
    multi-byte characters and strings that look like length prefixes included, NULLs excluded --, the counts equal the Single mode's, the model's and pyarrow's."""
    from dfgpu import capi, physical_plan as ops
    rng = np.random.default_rng(40 + parts)
    n = 20_000
    words = np.array(["", "a", "\x05\x00\x00\x00x", "é" * 3, "long-" * 40] + [f"w{v:03d}" for v in range(45)], dtype=object)
    k = rng.integers(0, 200, n).astype(np.int64)
    t = pa.table({"k": pa.array(k, mask=rng.random(n) < 0.02), "s": pa.array(words[rng.integers(0, len(words), n)].tolist(), type=pa.utf8(), mask=rng.random(n) < 0.25)})
    batches = [t.slice(o, 4096) for o in range(0, n, 4096)]
    C, F = ops.Column, ops.Field
    bs = [[ops.batch_from_arrow(ctx, b) for b in batches[p::parts]] for p in range(parts)]
    src = ops.MemoryExec(bs, bs[0][0].schema)
    aggs = lambda: [ops.AggregateFunctionExpr("COUNT DISTINCT", C("s", 1), "cs", input_field=F("s", capi.UTF8)), ops.AggregateFunctionExpr("COUNT", C("s", 1), "n", input_field=F("s", capi.UTF8))]
    tc = ops.TaskContext(ctx, 8192)
    st = pa.concat_tables([b.to_arrow() for b in ops.AggregateExec("Partial", [(C("k", 0), "k")], aggs(), src).execute(0, tc)])
    seen = pa.concat_tables(batches[0::parts])
    og = po.Groups([pa.int64()]); gids = og.intern([seen["k"].combine_chunks()])
    want = [[] for _ in range(len(og))]
    for g, x in zip(np.asarray(gids).tolist(), seen["s"].to_pylist()):
        if x is not None and x not in want[g]:
            want[g].append(x)
    assert st.column(0).combine_chunks().equals(og.emit()[0]) and string_lists_of(st.column(1)) == want
    plan = ops.AggregateExec("FinalPartitioned", [(C("k", 0), "k")], aggs(),
                             ops.CoalesceBatchesExec(ops.RepartitionExec(ops.AggregateExec("Partial", [(C("k", 0), "k")], aggs(), src), ops.Partitioning.Hash([C("k", 0)], 4)), 8192))
    out = pa.concat_tables([b.to_arrow() for p in range(4) for b in plan.execute(p, tc)]).sort_by([("k", "ascending")])
    single = pa.concat_tables([b.to_arrow() for b in ops.AggregateExec("Single", [(C("k", 0), "k")], aggs(), ops.MemoryExec([[x for p in bs for x in p]], bs[0][0].schema)).execute(0, tc)]).sort_by([("k", "ascending")])
    assert out.equals(single)
    ref = t.group_by("k", use_threads=False).aggregate([("s", "count_distinct"), ("s", "count")]).sort_by([("k", "ascending")])
    assert out["cs"].to_pylist() == ref["s_count_distinct"].to_pylist() and out["n"].to_pylist() == ref["s_count"].to_pylist()


def lists_of(col: pa.Array, dtype):
    """a list column as it travels on the device (Utf8 layout: byte offsets into the packed values) -> Python lists"""
    col = col.combine_chunks() if isinstance(col, pa.ChunkedArray) else col
    off = np.frombuffer(col.buffers()[1], dtype=np.int32)[col.offset: col.offset + len(col) + 1]
    data = np.frombuffer(col.buffers()[2], dtype=np.uint8) if col.buffers()[2] is not None else np.zeros(0, np.uint8)
    return [np.frombuffer(data[off[i]: off[i + 1]].tobytes(), dtype=dtype).tolist() for i in range(len(col))]


def partial_and_final(ctx, table, keys, col, typ, batches=1):
    """-> (the Partial stage's state lists, the Final stage's counts) of COUNT(DISTINCT col)"""
    from dfgpu import physical_plan as ops
    n = table.num_rows; step = max(1, (n + batches - 1) // batches)
    bs = [ops.batch_from_arrow(ctx, table.slice(o, step)) for o in range(0, max(n, 1), step)]
    src = ops.MemoryExec([bs], bs[0].schema)
    names = table.column_names; C, F = ops.Column, ops.Field
    agg = lambda: [ops.AggregateFunctionExpr("COUNT DISTINCT", C(col, names.index(col)), "cd", input_field=F(col, typ))]
    gb = [(C(k, names.index(k)), k) for k in keys]
    partial = ops.AggregateExec("Partial", gb, agg(), src)
    tc = ops.TaskContext(ctx, 8192)
    st = pa.concat_tables([b.to_arrow() for b in partial.execute(0, tc)])
    final = ops.AggregateExec("Final", [(C(k, i), k) for i, k in enumerate(keys)], agg(), ops.AggregateExec("Partial", gb, agg(), src))
    out = pa.concat_tables([b.to_arrow() for b in final.execute(0, tc)])
    return st, out


REF = "physical-expr/src/aggregate/count_distinct/mod.rs"
NUMERIC = [1, 1, None, 3, 2, None, 2, 3, 1]                    # test_count_distinct_update_batch_numeric (:305-331): state sorted == [1, 2, 3], result 3


@pytest.mark.gpu
@pytest.mark.parametrize("pat,npt,code", [(pa.int8(), np.int8, "INT8"), (pa.int16(), np.int16, "INT16"), (pa.int32(), np.int32, "INT32"), (pa.int64(), np.int64, "INT64"), (pa.uint8(), np.uint8, "UINT8"),
                                          (pa.uint16(), np.uint16, "UINT16"), (pa.uint32(), np.uint32, "UINT32"), (pa.uint64(), np.uint64, "UINT64")], ids=lambda x: str(x) if isinstance(x, pa.DataType) else "")
def test_count_distinct_state_and_result_reference_numeric(ctx, pat, npt, code):
    """count_distinct_update_batch_i8 .. u64 (mod.rs:482-520): the accumulator's state is ONE list holding the distinct values (compared sorted, as the reference does), the result 3"""
    from dfgpu import capi
    st, out = partial_and_final(ctx, pa.table({"v": pa.array(NUMERIC, type=pat)}), [], "v", getattr(capi, code))
    assert [sorted(x) for x in lists_of(st.column(0), npt)] == [[1, 2, 3]]
    assert out.column(0).to_pylist() == [3]


@pytest.mark.gpu
@pytest.mark.parametrize("pat,npt,code,sub", [(pa.float32(), np.float32, "FLOAT32", 1.0e-40), (pa.float64(), np.float64, "FLOAT64", 1.0e-308)], ids=["f32", "f64"])
def test_count_distinct_state_and_result_reference_floating_point(ctx, pat, npt, code, sub):
    """test_count_distinct_update_batch_floating_point (mod.rs:398-450): infinities, a subnormal, NaN twice (one value: Hashable compares bit patterns) -> 8 distinct"""
    from dfgpu import capi
    inf, nan = float("inf"), float("nan")
    vals = [inf, nan, 1.0, sub, 1.0, inf, None, 3.0, -4.5, 2.0, None, 2.0, 3.0, -inf, 1.0, nan, -inf]
    st, out = partial_and_final(ctx, pa.table({"v": pa.array(vals, type=pat)}), [], "v", getattr(capi, code))
    got = sorted(lists_of(st.column(0), npt)[0], key=lambda x: (np.isnan(x), x))
    want = [-inf, -4.5, float(npt(sub)), 1.0, 2.0, 3.0, inf]
    assert got[:-1] == want and np.isnan(got[-1])
    assert out.column(0).to_pylist() == [8]


@pytest.mark.gpu
def test_count_distinct_state_of_nulls_and_of_nothing(ctx):
    """count_distinct_update_batch_all_nulls / _empty (mod.rs:583-608): an empty list, result 0; count_distinct_update / _with_nulls (:611-680): -1, 5, 2 -> 3; NULLs do not count"""
    from dfgpu import capi
    st, out = partial_and_final(ctx, pa.table({"v": pa.array([None, None, None, None], type=pa.int32())}), [], "v", capi.INT32)
    assert lists_of(st.column(0), np.int32) == [[]] and out.column(0).to_pylist() == [0]
    st, out = partial_and_final(ctx, pa.table({"v": pa.array([], type=pa.int32())}), [], "v", capi.INT32)
    assert lists_of(st.column(0), np.int32) == [[]] and out.column(0).to_pylist() == [0]
    st, out = partial_and_final(ctx, pa.table({"v": pa.array([-1, 5, -1, 5, -1, -1, 2], type=pa.int32())}), [], "v", capi.INT32, batches=7)      # run_update: one row per update_batch
    assert [sorted(x) for x in lists_of(st.column(0), np.int32)] == [[-1, 2, 5]] and out.column(0).to_pylist() == [3]
    st, out = partial_and_final(ctx, pa.table({"v": pa.array([1, 1, 2, 1, None, None], type=pa.uint64())}), [], "v", capi.UINT64, batches=6)
    assert [sorted(x) for x in lists_of(st.column(0), np.uint64)] == [[1, 2]] and out.column(0).to_pylist() == [2]


@pytest.mark.gpu
@pytest.mark.parametrize("parts", [1, 3])
def test_count_distinct_partial_states_through_repartition_into_final(ctx, parts):
    """Partial -> RepartitionExec(Hash on the group key) -> CoalesceBatchesExec -> FinalPartitioned: the list states are taken, partitioned and concatenated as columns like
    any other; per-group lists of the Partial stage hold exactly the group's distinct values in first-seen order; the counts equal the model's and the Single mode's;
    Decimal128 values (16 bytes wide) and a second, plain aggregate next to it."""
    import decimal
    from dfgpu import capi, physical_plan as ops
    rng = np.random.default_rng(parts)
    n = 30_000
    k = rng.integers(0, 300, n).astype(np.int64); v = rng.integers(0, 40, n)
    t = pa.table({"k": pa.array(k, mask=rng.random(n) < 0.02), "v": pa.array(v, mask=rng.random(n) < 0.2), "d": pa.array([decimal.Decimal(int(x) * 7 - 100).scaleb(-2) for x in v], type=pa.decimal128(15, 2), mask=rng.random(n) < 0.1)})
    batches = [t.slice(o, 4096) for o in range(0, n, 4096)]
    C, F = ops.Column, ops.Field
    bs = [[ops.batch_from_arrow(ctx, b) for b in batches[p::parts]] for p in range(parts)]
    src = ops.MemoryExec(bs, bs[0][0].schema)
    aggs = lambda: [ops.AggregateFunctionExpr("COUNT DISTINCT", C("v", 1), "cv", input_field=F("v", capi.INT64)), ops.AggregateFunctionExpr("SUM", C("v", 1), "sv", input_field=F("v", capi.INT64)),
                    ops.AggregateFunctionExpr("COUNT DISTINCT", C("d", 2), "cd", input_field=F("d", capi.DECIMAL128, 15, 2))]
    partial = ops.AggregateExec("Partial", [(C("k", 0), "k")], aggs(), src)
    tc = ops.TaskContext(ctx, 8192)
    # the Partial stage alone (partition 0): lists == the distinct values of each group, first-seen order
    st = pa.concat_tables([b.to_arrow() for b in partial.execute(0, tc)])
    seen = pa.concat_tables(batches[0::parts])
    og = po.Groups([pa.int64()]); gids = og.intern([seen["k"].combine_chunks()])
    want = [[] for _ in range(len(og))]
    for g, x in zip(np.asarray(gids).tolist(), seen["v"].to_pylist()):
        if x is not None and x not in want[g]:
            want[g].append(x)
    assert st.column(0).combine_chunks().equals(og.emit()[0]) and lists_of(st.column(1), np.int64) == want
    plan = ops.AggregateExec("FinalPartitioned", [(C("k", 0), "k")], aggs(),
                             ops.CoalesceBatchesExec(ops.RepartitionExec(ops.AggregateExec("Partial", [(C("k", 0), "k")], aggs(), src), ops.Partitioning.Hash([C("k", 0)], 4)), 8192))
    out = pa.concat_tables([b.to_arrow() for p in range(4) for b in plan.execute(p, tc)]).sort_by([("k", "ascending")])
    single = pa.concat_tables([b.to_arrow() for b in ops.AggregateExec("Single", [(C("k", 0), "k")], aggs(), ops.MemoryExec([[x for p in bs for x in p]], bs[0][0].schema)).execute(0, tc)]).sort_by([("k", "ascending")])
    assert out.equals(single)
    ref = t.group_by("k", use_threads=False).aggregate([("v", "count_distinct"), ("v", "sum"), ("d", "count_distinct")]).sort_by([("k", "ascending")])
    assert out["cv"].to_pylist() == ref["v_count_distinct"].to_pylist() and out["cd"].to_pylist() == ref["d_count_distinct"].to_pylist() and out["sv"].to_pylist() == ref["v_sum"].to_pylist()
