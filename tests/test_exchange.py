"""N>1 path on CPU: world_size-2 `gloo` run of the all-to-all exchange used by ShuffleExec.  The hash partitioning
here is done by the CPU oracle (test infrastructure) so the test needs no GPU; what is under test is the exchange
logic of datafusion-upstream_amd/exchange.py: count matrix, split sizes, row alignment across columns, source order,
row conservation (≙ repartition/mod.rs:952-1031), and that a partitioned hash join over exchanged shards equals the
single-process join (hash_join.rs:1505-1549 partitioned_join_collect)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import pyarrow as pa
    import torch
    import torch.distributed as dist
    import dfgpu
    from dfgpu import exchange
    from oracle import pyoracle as po
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(1234)              # same global tables on every rank; each rank owns a slice
        nb, npr = 4000, 15000
        bkeys, bvals = rng.integers(0, 3000, nb).astype(np.int64), rng.integers(0, 10**6, nb).astype(np.int32)
        pkeys, pvals = rng.integers(0, 3500, npr).astype(np.int64), rng.integers(0, 10**6, npr).astype(np.int32)

        def shuffle(keys, vals):
            lo, hi = len(keys) * rank // world, len(keys) * (rank + 1) // world
            k, v = keys[lo:hi], vals[lo:hi]
            idx, counts = po.hash_partition([pa.array(k)], world)
            parts, off = [], 0
            for d in range(world):
                sel = idx[off:off + counts[d]]
                off += counts[d]
                parts.append([torch.from_numpy(k[sel].copy()).view(torch.uint8), torch.from_numpy(v[sel].copy()).view(torch.uint8)] if counts[d] else None)
            rc, cols = exchange.exchange_byte_columns(parts, counts.tolist(), [8, 4])
            rk, rv = cols[0].view(torch.int64).numpy(), cols[1].view(torch.int32).numpy()
            assert len(rk) == len(rv) == sum(rc)
            # every received row belongs to this rank's partition
            assert (po.create_hashes([pa.array(rk)]) % world == rank).all()
            # rows stay aligned across columns: (key, val) pairs received are pairs that were sent
            sent = set(zip(keys.tolist(), vals.tolist()))
            assert all(p in sent for p in zip(rk.tolist(), rv.tolist()))
            total = torch.tensor([len(rk)]); dist.all_reduce(total)
            assert int(total) == len(keys)              # row conservation
            return rk, rv

        bk, bv = shuffle(bkeys, bvals)
        pk, pv = shuffle(pkeys, pvals)
        res = po.hash_join([[pa.array(bk)]], [[pa.array(pk)]], "Inner")
        local = sorted(zip(bk[res.build_idx].tolist(), bv[res.build_idx].tolist(), pv[res.probe_idx].tolist()))
        gathered = [None] * world
        dist.all_gather_object(gathered, local)
        if rank == 0:
            full = po.hash_join([[pa.array(bkeys)]], [[pa.array(pkeys)]], "Inner")
            want = sorted(zip(bkeys[full.build_idx].tolist(), bvals[full.build_idx].tolist(), pvals[full.probe_idx].tolist()))
            assert sorted(r for part in gathered for r in part) == want
        # counts matrix helper on its own
        rc = exchange.all_to_all_counts([rank * 10 + d for d in range(world)])
        assert rc == [s * 10 + rank for s in range(world)]
        # the single metadata all-gather of exchange_batches: count matrix + column types / validity flags; rank 1 holds no rows
        fields = [(5, 0, 0), (13, 15, 2)] if rank == 0 else None
        ub = [[30, 7], [41, 0]] if rank == 0 else None                  # value bytes of two Utf8 columns per destination
        recv, allc, flds, nullable, ru = exchange._exchange_meta([3, 4] if rank == 0 else [0, 0], fields, [0, 1] if rank == 0 else None, None, ub)
        assert allc == [[3, 4], [0, 0]] and recv == ([3, 0] if rank == 0 else [4, 0])
        assert flds == [(5, 0, 0), (13, 15, 2)] and nullable == [False, True]
        assert ru[0][:2] == ([30, 7] if rank == 0 else [41, 0]) and ru[1][:2] == [0, 0]      # what each source sends to ME
        recv, allc, flds, nullable, ru = exchange._exchange_meta([0, 0], None, None, None)       # nobody holds rows: no columns, no data collectives
        assert flds == [] and recv == [0, 0]
        q.put((rank, "ok"))
    except Exception as e:  # noqa
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_all_to_all_exchange_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}: {msg}"
