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
    aggs = [("MIN", "s", capi.UTF8), ("MAX", "s", capi.UTF8), ("COUNT", "v", capi.INT64)]
    if mode == "Single":
        aggs += [("COUNT DISTINCT", "v", capi.INT64), ("COUNT DISTINCT", "s", capi.UTF8)]
    out = agg_plan(ctx, batches, ["k"], aggs, mode)
    allt = pa.concat_tables(batches)
    og = po.Groups([pa.int64()]); gids = og.intern([allt["k"].combine_chunks()]); total = len(og)
    assert out.column(0).combine_chunks().equals(og.emit()[0])                       # groups in first-seen order
    s, v = allt["s"].combine_chunks(), allt["v"].combine_chunks()
    assert out.column(1).combine_chunks().equals(po.string_min_max(s, gids, total, False))
    assert out.column(2).combine_chunks().equals(po.string_min_max(s, gids, total, True))
    if mode == "Single":
        assert out.column(4).combine_chunks().equals(po.count_distinct(v, gids, total))
        assert out.column(5).combine_chunks().equals(po.count_distinct(s, gids, total))


@pytest.mark.gpu
def test_device_refuses_count_distinct_in_partial_mode(ctx):
    import dfgpu
    from dfgpu import capi
    t = pa.table({"k": pa.array([1, 2], type=pa.int64()), "v": pa.array([1, 1], type=pa.int64())})
    with pytest.raises(dfgpu.DfgpuError) as e:
        agg_plan(ctx, [t], ["k"], [("COUNT DISTINCT", "v", capi.INT64)], "PartialFinal")
    assert e.value.kind == "NotImplemented"
