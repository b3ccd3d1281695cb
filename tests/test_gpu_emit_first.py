"""-m gpu: EmitTo::First(n) on GroupValues and GroupsAccumulator (dfgpu_groups_emit_first, dfgpu_acc_emit_first) -- the contract of
expr/src/groups_accumulator.rs:25-57: the first n groups leave (keys, final values or states), the remaining groups are renumbered from 0 and keep
accumulating.  Checked against a streaming model over the oracle: emitting in pieces must give exactly what one emit at the end gives, piece by piece in
first-seen order, for every key path of groups.hip (primitive, general multi-column, dictionary, Utf8) and every accumulator kind."""
import numpy as np
import pyarrow as pa
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
KIND = {"SUM": 0, "AVG": 1, "COUNT": 2, "MIN": 3, "MAX": 4}
RNG = np.random.default_rng(77)


def key_sets(n):
    ints = pa.array(RNG.integers(0, 300, n).astype(np.int64) * 7919)
    return {
        "primitive-int64": [ints],
        "two-columns": [pa.array(RNG.integers(0, 20, n).astype(np.int32), mask=RNG.random(n) < 0.1), pa.array(RNG.integers(0, 15, n).astype(np.int64))],
        "utf8": [pa.array([f"key-{v}" for v in RNG.integers(0, 200, n)], mask=RNG.random(n) < 0.05)],
        "dictionary": [pa.array([f"d{v % 50}" for v in RNG.integers(0, 500, n)]).dictionary_encode()],
    }


@pytest.mark.parametrize("shape", ["primitive-int64", "two-columns", "utf8", "dictionary"])
def test_emit_first_in_pieces_equals_one_emit(ctx, shape):
    import dfgpu
    n = 5000
    keys = key_sets(n)[shape]
    vi = pa.array(RNG.integers(-1000, 1000, n).astype(np.int64), mask=RNG.random(n) < 0.1); vf = pa.array(RNG.random(n))
    specs = [("SUM", vi, dfgpu.capi.INT64), ("COUNT", vi, dfgpu.capi.INT64), ("MIN", vi, dfgpu.capi.INT64), ("MAX", vi, dfgpu.capi.INT64), ("AVG", vf, dfgpu.capi.FLOAT64)]
    plain = [k.dictionary_decode() if pa.types.is_dictionary(k.type) else k for k in keys]
    # the reference result: everything interned and accumulated, one emit
    og = po.Groups([k.type for k in plain]); gids = og.intern(plain); want_keys = og.emit()
    want = []
    for kind, v, _ in specs:
        acc = po.Acc(kind, v.type); acc.update_batch(v, gids, None, len(og)); want.append(acc.evaluate())
    total = len(og)
    # device: first half of the rows, emit the first 37 groups, the rest of the rows, emit in two more pieces
    half = n // 2
    gv = dfgpu.GroupValues(ctx, len(keys)); accs = [dfgpu.GroupsAccumulator(ctx, KIND[k], t) for k, _, t in specs]
    dk = [ctx.from_arrow(k) for k in keys]; dv = [ctx.from_arrow(v) for _, v, _ in specs]

    def feed(lo, hi):
        ids = gv.intern([c.slice(lo, hi - lo) for c in dk])
        for a, v in zip(accs, dv):
            a.update_batch(v.slice(lo, hi - lo), ids, None, len(gv))
    feed(0, half)
    seen_first = len(gv)
    k1 = min(37, seen_first)
    pieces_k = [[c.to_arrow() for c in gv.emit_first(k1)]]
    pieces_v = [[a.emit_first(k1)[0].to_arrow() for a in accs]]
    assert len(gv) == seen_first - k1
    # groups emitted early must not receive more rows: feed only rows whose group was not emitted yet (what GroupOrdering guarantees upstream)
    emitted = set(zip(*[c.to_pylist() for c in [x.dictionary_decode() if pa.types.is_dictionary(x.type) else x for x in pieces_k[0]]]))
    rows = [i for i in range(half, n) if tuple(c[i].as_py() for c in plain) not in emitted]
    idx = pa.array(rows, type=pa.int64())
    dk2 = [ctx.from_arrow(k.take(idx)) for k in keys]; dv2 = [ctx.from_arrow(v.take(idx)) for _, v, _ in specs]
    ids = gv.intern(dk2)
    for a, v in zip(accs, dv2):
        a.update_batch(v, ids, None, len(gv))
    left = len(gv); k2 = left // 3
    pieces_k.append([c.to_arrow() for c in gv.emit_first(k2)]); pieces_v.append([a.emit_first(k2)[0].to_arrow() for a in accs])
    pieces_k.append([c.to_arrow() for c in gv.emit_first(10**9)]); pieces_v.append([a.emit_first(10**9)[0].to_arrow() for a in accs])
    assert len(gv) == 0
    dec = lambda x: x.dictionary_decode() if pa.types.is_dictionary(x.type) else x
    got_keys = [pa.concat_arrays([dec(p[c]) for p in pieces_k]) for c in range(len(keys))]
    # model: the same streaming over the oracle's one-shot result restricted to the rows fed
    fed = list(range(half)) + rows
    fk = [k.take(pa.array(fed, type=pa.int64())) for k in plain]
    og2 = po.Groups([k.type for k in plain]); g2 = og2.intern(fk); wk = og2.emit()
    for g, w in zip(got_keys, wk):
        assert g.equals(w)                                  # same groups in the same first-seen order, across the pieces
    for j, (kind, v, _) in enumerate(specs):
        acc = po.Acc(kind, v.type); acc.update_batch(v.take(pa.array(fed, type=pa.int64())), g2, None, len(og2)); w = acc.evaluate()
        g = pa.concat_arrays([p[j] for p in pieces_v])
        assert g.type == w.type and len(g) == len(w)
        if pa.types.is_floating(w.type):
            assert np.allclose(g.to_numpy(zero_copy_only=False), w.to_numpy(zero_copy_only=False), rtol=1e-9, atol=0, equal_nan=True)
        else:
            assert g.equals(w), kind
    assert total >= len(og2)


def test_emit_first_states_merge_like_a_final_stage(ctx):
    """as_state = 1: the state arrays of the first n groups (AVG: counts and sums) merge downstream like any Partial output"""
    import dfgpu
    n = 3000
    k = pa.array(RNG.integers(0, 100, n).astype(np.int64)); v = pa.array(RNG.random(n))
    gv = dfgpu.GroupValues(ctx, 1); acc = dfgpu.GroupsAccumulator(ctx, KIND["AVG"], dfgpu.capi.FLOAT64)
    ids = gv.intern([ctx.from_arrow(k)]); acc.update_batch(ctx.from_arrow(v), ids, None, len(gv))
    total = len(gv)
    st_a = acc.emit_first(40, as_state=True); keys_a = gv.emit_first(40)
    st_b = acc.emit_first(total, as_state=True); keys_b = gv.emit_first(total)
    assert len(st_a) == 2 and len(st_a[0]) == 40 and len(st_b[0]) == total - 40
    fin_g = dfgpu.GroupValues(ctx, 1); fin = dfgpu.GroupsAccumulator(ctx, KIND["AVG"], dfgpu.capi.FLOAT64)
    for keys, st in ((keys_a, st_a), (keys_b, st_b)):
        g = fin_g.intern(keys); fin.merge_batch(st, g, None, len(fin_g))
    og = po.Groups([k.type]); gids = og.intern([k]); oa = po.Acc("AVG", v.type); oa.update_batch(v, gids, None, len(og))
    assert fin_g.emit()[0].to_arrow().equals(og.emit()[0])
    assert np.allclose(fin.evaluate().to_arrow().to_numpy(), oa.evaluate().to_numpy(), rtol=1e-9, atol=0)
