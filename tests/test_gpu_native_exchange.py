"""-m gpu: dfgpu_exchange (the C entry point, csrc/exchange.hip) over the caller-provided transport with 2 and 3 ranks on ONE GPU (gloo, staged
through the host; RCCL refuses two ranks on one device): Utf8, nullable Utf8, Boolean, dictionary and nullable Decimal128 lanes travel natively;
a rank without rows takes part; a rank that fails alone before the collective makes EVERY rank fail (status word in the metadata all-gather) instead
of leaving its peers waiting; a transport error on one lane leaves the communicator usable for the next call."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tables(rank, world):
    import decimal
    import pyarrow as pa
    rng = np.random.default_rng(100 + rank)
    n = 0 if (world == 3 and rank == 2) else 4000 + 500 * rank           # world 3: the last rank has no rows at all
    keys = rng.integers(0, 700, n).astype(np.int64)
    words = [f"w{int(k) % 97}-{'x' * int(k % 7)}" for k in keys]
    s_null = pa.array(words, mask=(rng.random(n) < 0.2) if rank == 0 else None)      # nullable on rank 0 only: every rank must receive a bitmap
    s_plain = pa.array([w[::-1] for w in words])
    flag = pa.array((keys % 3 == 0), mask=rng.random(n) < 0.1)
    dic = pa.DictionaryArray.from_arrays(pa.array((keys % 5).astype(np.int32)), pa.array(["alpha", "beta", "", "delta", "epsilon-long-value"]))
    dec = pa.array([decimal.Decimal(int(k) * 37 - 5000).scaleb(-2) for k in keys], type=pa.decimal128(15, 2), mask=rng.random(n) < 0.15)
    return pa.table({"k": pa.array(keys), "s_null": s_null, "s_plain": s_plain, "flag": flag, "dic": dic, "dec": dec})


def _rows(table):
    cols = [c.combine_chunks() for c in table.columns]
    cols = [c.cast(c.type.value_type) if hasattr(c.type, "value_type") else c for c in cols]
    return list(zip(*[c.to_pylist() for c in cols]))


def _worker(rank, world, port, q, scenario):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    try:
        import faulthandler
        faulthandler.dump_traceback_later(120, exit=True)      # a rank stuck in a collective must not hold the GPU box
        import pyarrow as pa
        import torch
        import torch.distributed as dist
        import dfgpu
        from dfgpu import capi, exchange
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        ctx = dfgpu.Context(0, stream=torch.cuda.current_stream().cuda_stream)
        comm = exchange.Comm(ctx, force_callbacks=True)
        t = _tables(rank, world)
        arrays = [ctx.from_arrow(c.combine_chunks()) for c in t.columns] if t.num_rows else None

        def run():
            got, sent, recv = comm.exchange([arrays[0]] if arrays else None, arrays, t.num_columns)
            if got is None:
                return []
            return _rows(pa.table({name: a.to_arrow() for name, a in zip(t.column_names, got)}))

        out = {"mine": _rows(t)}
        if scenario == "lanes":
            out["got"] = run()
        elif scenario == "transport_error":
            comm.fail_lane = 3                           # every rank's callback refuses the fourth lane of the next call (no rank enters that collective alone)
            try:
                run(); out["first"] = "no error"
            except capi.DfgpuError as e:
                out["first"] = str(e)
            comm.fail_lane = None
            out["got"] = run()                           # the communicator is still usable
        elif scenario == "rank_fails_alone":
            if rank == 1:                                # a column shorter than the keys: a local argument error BEFORE the collective
                bad = list(arrays); bad[2] = bad[2].slice(0, len(bad[2]) - 1)
                try:
                    comm.exchange([arrays[0]], bad, t.num_columns); out["first"] = "no error"
                except capi.DfgpuError as e:
                    out["first"] = str(e)
            else:
                try:
                    run(); out["first"] = "no error"
                except capi.DfgpuError as e:
                    out["first"] = str(e)
            out["got"] = run()
        elif scenario in ("column_count_mismatch", "memory_limit_on_one_rank"):
            try:
                if rank == 1 and scenario == "column_count_mismatch":          # one rank hands over one column fewer: the metadata round still has one length everywhere
                    comm.exchange([arrays[0]], arrays[:-1], t.num_columns - 1)
                elif scenario == "memory_limit_on_one_rank":
                    # a per-ctx option that differs between ranks: only rank 1 runs under a limit, raised round by round, so that its allocation failure moves from
                    # the partition pass (status word of the first round) through the receive buffers (second status round) to none at all
                    base, rounds = ctx.get_option("live_bytes"), []
                    for k in range(48):
                        if rank == 1:
                            ctx.set_option("memory_limit", base + 4096 + k * 16384)
                        try:
                            run(); rounds.append("ok")
                        except capi.DfgpuError as e:
                            rounds.append(str(e))
                        finally:
                            ctx.set_option("memory_limit", 0)
                    out["rounds"] = rounds
                else:
                    run()
                out["first"] = "no error"
            except capi.DfgpuError as e:
                out["first"] = str(e)
            out["got"] = run()
        q.put((rank, out))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # noqa
        import traceback
        q.put((rank, traceback.format_exc()))


def _launch(world, scenario):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 28100 + (os.getpid() % 800) + world + {"lanes": 0, "transport_error": 10, "rank_fails_alone": 20, "column_count_mismatch": 30, "memory_limit_on_one_rank": 40}[scenario]
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, scenario)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=150) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for r in range(world):
        assert isinstance(results[r], dict), results[r]
    return results


def _check_exchange(results, world):
    everything = [row for r in range(world) for row in results[r]["mine"]]
    arrived = [row for r in range(world) for row in results[r]["got"]]
    assert sorted(arrived, key=repr) == sorted(everything, key=repr), "every row arrives exactly once, every cell intact (NULLs, empty strings, Booleans, decimals)"
    owner = {}
    for r in range(world):
        for row in results[r]["got"]:
            assert owner.setdefault(row[0], r) == r, "equal keys must land on one rank"
    # rows of one source keep their order inside a destination (rows arrive ordered by source rank, RepartitionExec's per-input order)
    for r in range(world):
        pos = 0
        for src in range(world):
            mine = [row for row in results[src]["mine"] if owner.get(row[0]) == r]
            assert results[r]["got"][pos:pos + len(mine)] == mine
            pos += len(mine)


@pytest.mark.parametrize("world", [2, 3])
def test_native_exchange_moves_utf8_boolean_dictionary_and_nullable_lanes(world):
    _check_exchange(_launch(world, "lanes"), world)


def test_transport_error_leaves_the_communicator_usable():
    res = _launch(2, "transport_error")
    for r in range(2):
        assert "all_to_all_v failed on lane 3" in res[r]["first"], res[r]["first"]
    _check_exchange(res, 2)


def test_a_rank_that_fails_alone_fails_every_rank_and_nothing_hangs():
    res = _launch(3, "rank_fails_alone")
    assert "differs in length" in res[1]["first"], res[1]["first"]
    for r in (0, 2):
        assert "rank 1 failed before the collective" in res[r]["first"], res[r]["first"]
    _check_exchange(res, 3)


def test_a_rank_with_another_column_count_fails_every_rank():
    """The metadata all-gather is sized by the column limit, not by the local column count, so ranks that were handed different counts still run one
    well-formed collective, read the same matrix and fail together."""
    res = _launch(3, "column_count_mismatch")
    for r in range(3):
        assert "columns" in res[r]["first"] and "no error" not in res[r]["first"], res[r]["first"]
    _check_exchange(res, 3)


def test_a_memory_limit_on_one_rank_fails_every_rank():
    """One rank runs under a memory limit its peers do not have; the limit is raised round by round so the rank fails first in its partition pass, then while
    allocating its receive buffers, then not at all.  In every round either every rank gets an error or none does (the status rounds do not depend on a
    per-ctx option), nobody waits in the data collective, and the communicator works afterwards."""
    res = _launch(3, "memory_limit_on_one_rank")
    rounds = [res[r]["rounds"] for r in range(3)]
    assert len(rounds[0]) == len(rounds[1]) == len(rounds[2]) == 48
    for k in range(48):
        oks = [rounds[r][k] == "ok" for r in range(3)]
        # all ranks fail together, or none does -- or the limited rank alone runs out AFTER the data collective (a scan's scratch while the received lanes become arrays):
        # that is a local error like any other operator's, its peers hold their rows and nobody waits
        assert all(oks) or not any(oks) or (oks == [True, False, True] and "Resources exhausted" in rounds[1][k]), (k, [rounds[r][k] for r in range(3)])
        if not oks[0]:
            assert "rank 1" in rounds[0][k] and "rank 1" in rounds[2][k], rounds[0][k]
    assert rounds[0][0] != "ok" and rounds[0][-1] == "ok"
    assert any("could not allocate its receive buffers" in x for x in rounds[0]), "the sweep never failed inside the receive-buffer allocation: " + repr(sorted(set(rounds[0])))
    _check_exchange(res, 3)
