"""Multi-GPU shuffle: the RepartitionExec exchange as one RCCL all-to-all(v) per column buffer.

The reference moves per-destination RecordBatch slices through in-process channels
(physical-plan/src/repartition/mod.rs:442-580, :684-760).  With one process per GPU the same slices
(produced by dfgpu_hash_partition + take) become the send segments of `torch.distributed.all_to_all_single`
(backend "nccl" == RCCL over xGMI; "gloo" on CPU for the tests).  xGMI is point-to-point, so an all-to-all
keeps all 7 links of a GPU busy at once; one collective per column buffer keeps messages large.

`exchange_byte_columns` is device agnostic (CPU tensors under gloo -- tests/test_exchange.py runs it with
world_size 2); `exchange_batches` / `ShuffleExec` adapt device RecordBatches to it.
"""
from __future__ import annotations

from typing import List, Optional, Sequence


def all_to_all_counts(send_counts: Sequence[int], group=None) -> List[int]:
    """Exchange the row-count matrix: returns recv_counts[src] (rows this rank receives from every rank)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    send = torch.tensor(list(send_counts), dtype=torch.int64, device=dev)
    recv = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_to_all_single(recv, send, group=group)
    return [int(x) for x in recv.cpu().tolist()]


def all_to_all_buffers(send, send_elems: Sequence[int], recv_elems: Sequence[int], group=None):
    """all-to-all(v) of one flat buffer: `send` holds the segments for rank 0..N-1 back to back
    (send_elems[i] elements each); returns the received buffer (segments ordered by source rank)."""
    import torch
    import torch.distributed as dist
    out = torch.empty(int(sum(recv_elems)), dtype=send.dtype, device=send.device)
    dist.all_to_all_single(out, send.contiguous(), output_split_sizes=[int(x) for x in recv_elems],
                           input_split_sizes=[int(x) for x in send_elems], group=group)
    return out


def _staged(t, group):
    """RCCL moves device tensors directly; under gloo (CPU tests, or 2 ranks rehearsing on ONE GPU) stage through host memory."""
    return t if _is_nccl(group) or not t.is_cuda else t.cpu()


def exchange_byte_columns(parts: List[Optional[List]], counts: Sequence[int], widths: Sequence[int], group=None):
    """parts[dest] = list (one per column) of 1-D uint8 tensors holding counts[dest] rows of widths[c] bytes,
    or None when nothing goes to `dest`.  Returns (recv_counts, [uint8 tensor per column]) with the received
    rows ordered by source rank -- row i of every column still belongs to the same logical row."""
    import torch
    recv_counts = all_to_all_counts(counts, group)
    ref = next((p[0] for p in parts if p), None)
    device = ref.device if ref is not None else ("cuda" if _is_nccl(group) else "cpu")
    out = []
    for c, w in enumerate(widths):
        segs = [p[c] for p, n in zip(parts, counts) if p and n]
        send = torch.cat(segs) if segs else torch.empty(0, dtype=torch.uint8, device=device)
        assert send.numel() == sum(counts) * w, "segment sizes do not match the row counts"
        got = all_to_all_buffers(_staged(send, group), [n * w for n in counts], [n * w for n in recv_counts], group)
        out.append(got.to(device) if got.device != send.device else got)
    return recv_counts, out


def _is_nccl(group) -> bool:
    import torch.distributed as dist
    return dist.get_backend(group) == "nccl"


class _DevicePtr:
    """__cuda_array_interface__ view of raw device memory so torch can alias a dfgpu buffer without a copy."""

    def __init__(self, ptr: int, nbytes: int, owner):
        self.owner = owner
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


# byte width per dfgpu_type for the fixed-width types
_WIDTH = {2: 1, 6: 1, 3: 2, 7: 2, 4: 4, 8: 4, 10: 4, 12: 4, 5: 8, 9: 8, 11: 8, 13: 16}


_MAX_COLS = 64


def _same_stream(ctx) -> bool:
    """dfgpu kernels and torch / RCCL work are stream ordered when the context runs on torch's current stream (bench.py);
    a context with a private stream (the unit tests) needs explicit synchronisation around the collectives."""
    import torch
    return ctx.stream == torch.cuda.current_stream().cuda_stream


_MAX_UTF8 = 8


def _exchange_meta(counts: Sequence[int], fields, has_valid: Sequence[int], group, utf8_bytes: Optional[Sequence[Sequence[int]]] = None):
    """ONE small all-gather carries everything the ranks must agree on before the data moves: the row-count matrix, the value
    bytes of every Utf8 column per destination, and -- for ranks that hold no rows and therefore no arrays -- the column types
    and which columns carry a validity bitmap.
    Returns (recv_counts[src], all_counts[src][dst], fields [(dtype, precision, scale)], nullable [bool], recv_utf8[src][k])."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    ncols = len(fields) if fields is not None else 0
    if ncols > _MAX_COLS:
        raise ValueError(f"exchange of {ncols} columns (max {_MAX_COLS})")
    row = list(counts) + [ncols]
    for c in range(ncols):
        row += [fields[c][0], fields[c][1], fields[c][2], int(has_valid[c])]
    row += [0] * (world + 1 + 4 * _MAX_COLS - len(row))
    ub = [[0] * _MAX_UTF8 for _ in range(world)]
    if utf8_bytes is not None:
        for d in range(world):
            for k, b in enumerate(utf8_bytes[d]):
                ub[d][k] = int(b)
    for d in range(world):
        row += ub[d]
    dev = "cuda" if _is_nccl(group) else "cpu"
    mine = torch.tensor(row, dtype=torch.int64, device=dev)
    allm = torch.empty((world, mine.numel()), dtype=torch.int64, device=dev)      # also at world 1: the RCCL path is the one a 1-rank test covers
    _beat("exchange metadata all-gather (%d columns)" % ncols)
    dist.all_gather_into_tensor(allm, mine, group=group) if _is_nccl(group) else dist.all_gather(list(allm.unbind(0)), mine, group=group)
    m = allm.cpu().tolist()                                     # the one host round trip of an exchange
    all_counts = [r[:world] for r in m]
    src = next((r for r in m if r[world] > 0), None)
    if src is None:
        return [0] * world, all_counts, [], [], [[0] * _MAX_UTF8 for _ in range(world)]
    nc = src[world]
    flds = [tuple(src[world + 1 + 4 * c: world + 4 + 4 * c]) for c in range(nc)]
    nullable = [any(r[world] > 0 and r[world + 4 + 4 * c] for r in m) for c in range(nc)]
    ub0 = world + 1 + 4 * _MAX_COLS
    recv_utf8 = [m[s][ub0 + rank * _MAX_UTF8: ub0 + (rank + 1) * _MAX_UTF8] for s in range(world)]
    return [all_counts[s][rank] for s in range(world)], all_counts, flds, nullable, recv_utf8


def exchange_batches(ctx, schema, parts: List[Optional["RecordBatch"]], group=None, names: Optional[Sequence[str]] = None, broadcast: bool = False):
    with ctx.deferred_flags():          # materialising pending gathers of the outgoing columns: one flag check for all of them
        return _exchange_batches(ctx, schema, parts, group, names, broadcast)


def _exchange_batches(ctx, schema, parts, group, names, broadcast):
    """parts[dest] = the rows of this rank bound for rank `dest` (None/empty allowed; fixed-width and Utf8 columns, nullable or not).
    Returns one RecordBatch holding everything this rank received, source ranks in order (≙ the batches a RepartitionExec
    output partition yields).  One metadata all-gather (_exchange_meta), then one collective per column buffer: all-to-all(v)
    for a shuffle, all-gather for `broadcast` (every part is the same batch).  A column that carries a validity bitmap on ANY
    rank also moves its bitmap words, and the receiver splices the per-source bitmaps at bit granularity.  `schema` may be
    None on a rank without rows; `names` (static, from the plan) then names the columns."""
    import torch
    import torch.distributed as dist
    from . import capi, operators as ops, physical_plan as pp
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    assert len(parts) == world
    counts = [0 if p is None else p.num_rows for p in parts]
    first = next((p for p, n in zip(parts, counts) if p is not None and n), None)
    fields = has_valid = None
    if first is not None:
        fs = (schema or first.schema).fields
        fields = [(f.dtype, f.precision, f.scale) for f in fs]
        for f in fs:
            if f.dtype not in _WIDTH and f.dtype != capi.UTF8:
                raise ops.DfgpuError(4, f"exchange of column type {f.dtype} is not supported yet")
        ucols = [c for c, f in enumerate(fs) if f.dtype == capi.UTF8]
        if len(ucols) > _MAX_UTF8:
            raise ops.DfgpuError(4, f"exchange of more than {_MAX_UTF8} Utf8 columns")
        has_valid = [0] * len(fs)
        for p, n in zip(parts[:1] if broadcast else parts, counts):
            if p is not None and n:
                for c, col in enumerate(p.columns):
                    d = col.describe()
                    if d.type == capi.DICTIONARY:
                        raise ops.DfgpuError(4, "exchange of dictionary-encoded columns is not supported yet (cast to the value type first)")
                    if d.validity:
                        has_valid[c] = 1
        utf8_bytes = [[(p.columns[c].describe().values_bytes if p is not None and n else 0) for c in ucols] for p, n in zip(parts, counts)]
    else:
        utf8_bytes = None
    recv_counts, all_counts, flds, nullable, recv_utf8 = _exchange_meta(counts, fields, has_valid, group, utf8_bytes)
    if not flds:
        return pp.RecordBatch.from_arrays(ctx, [], [])
    if names is None:
        names = (schema or first.schema).names() if (schema is not None or first is not None) else [f"c{c}" for c in range(len(flds))]
    widths = [_WIDTH.get(t, 0) for t, _, _ in flds]            # 0 = Utf8: (n + 1) int32 offsets + value bytes
    ucols = [c for c, (t, _, _) in enumerate(flds) if t == capi.UTF8]
    sync = not _same_stream(ctx)
    nccl = _is_nccl(group)
    total = int(sum(recv_counts))
    pad8 = lambda x: (x + 7) // 8 * 8
    vwords = lambda n: ((n + 63) // 64) * 8

    # One message per (source, destination): every column's values (padded to 8 bytes; a Utf8 column = its offsets, then its value
    # bytes) followed by its validity words when the column is nullable anywhere -- so an exchange is ONE data collective, whatever
    # the number of columns (a collective costs ~100 us of launch latency on RCCL; TPC-H Q3's gather has 4 columns, its shuffles 3-5).
    def layout(n, ubytes):
        offs, o = [], 0
        if n == 0:
            return [(0, None, None)] * len(widths), 0          # nothing is sent for an empty part (not even a Utf8 column's single offset)
        for c, w in enumerate(widths):
            vo = o
            if w:
                o += pad8(n * w); bo2 = None
            else:
                o += pad8((n + 1) * 4); bo2 = o; o += pad8(ubytes[ucols.index(c)])
            bo = None
            if nullable[c]:
                bo = o; o += vwords(n)
            offs.append((vo, bo, bo2))
        return offs, o

    def device_bytes(arr, nbytes):
        d = arr.describe()
        return torch.as_tensor(_DevicePtr(d.values, nbytes, arr), device="cuda")

    zeros8 = torch.zeros(8, dtype=torch.uint8, device="cuda")
    keep = []

    def message(p, n):
        """the packed bytes of batch p (n rows) as a list of tensors to concatenate"""
        segs = []
        for c, col in enumerate(p.columns):
            d = col.describe()
            if widths[c]:
                nb = n * widths[c]
                segs.append(device_bytes(col, nb))
            else:
                nb = (n + 1) * 4
                segs.append(torch.as_tensor(_DevicePtr(d.offsets, nb, col), device="cuda"))
            if pad8(nb) != nb:
                segs.append(zeros8[:pad8(nb) - nb])
            if not widths[c]:
                vb = d.values_bytes
                if vb:
                    segs.append(torch.as_tensor(_DevicePtr(d.values, vb, col), device="cuda"))
                if pad8(vb) != vb:
                    segs.append(zeros8[:pad8(vb) - vb])
            if nullable[c]:
                bm = ctx.is_null(col, negate=True)           # validity as a Boolean column (all ones when there is no bitmap)
                keep.append(bm)
                segs.append(device_bytes(bm, vwords(n)))
        return segs

    recv_sizes = [layout(n, recv_utf8[s_])[1] for s_, n in enumerate(recv_counts)]
    if broadcast:
        segs = message(parts[rank], counts[rank]) if parts[rank] is not None and counts[rank] else []
        if sync:
            ctx.synchronize()                  # producers (and is_null above) ran on the ctx stream
        stride = (max(recv_sizes) + 63) // 64 * 64           # row stride of the gathered buffer: keeps every source's message aligned
        pad = torch.empty(stride, dtype=torch.uint8, device="cuda")
        if segs:
            torch.cat(segs, out=pad[:recv_sizes[rank]])
        out = torch.empty((world, stride), dtype=torch.uint8, device="cuda" if nccl else "cpu")
        _beat("broadcast data all-gather (%d bytes per rank)" % stride)
        if nccl:
            dist.all_gather_into_tensor(out, pad, group=group)
        else:
            dist.all_gather(list(out.unbind(0)), pad.cpu(), group=group)
            out = out.to("cuda")
        base = [s * stride for s in range(world)]
        buf = out.view(-1)
    else:
        segs, send_sizes = [], []
        for p, n in zip(parts, counts):
            send_sizes.append(layout(n, utf8_bytes[len(send_sizes)])[1] if p is not None and n else 0)
            if p is not None and n:
                segs += message(p, n)
        if sync:
            ctx.synchronize()
        send = torch.cat(segs) if len(segs) > 1 else (segs[0] if segs else torch.empty(0, dtype=torch.uint8, device="cuda"))
        buf = all_to_all_buffers(_staged(send, group), send_sizes, recv_sizes, group)
        buf = buf.to("cuda") if buf.device.type != "cuda" else buf
        base, o = [], 0
        for sz in recv_sizes:
            base.append(o); o += sz
    if sync:
        torch.cuda.current_stream().synchronize()
    out_cols = []
    lay = [layout(n, recv_utf8[s_])[0] for s_, n in enumerate(recv_counts)]
    for c, (t, prec, scale) in enumerate(flds):
        w = widths[c]
        if w:
            vals = [ctx.wrap_tensor(buf[base[s] + lay[s][c][0]: base[s] + lay[s][c][0] + n * w], t, prec, scale) for s, n in enumerate(recv_counts) if n]
        else:
            vals = []
            for s_, n in enumerate(recv_counts):
                if not n:
                    continue
                d = capi.ArrayDesc()
                d.type, d.length, d.null_count = capi.UTF8, n, 0
                d.offsets = buf.data_ptr() + base[s_] + lay[s_][c][0]
                d.values_bytes = recv_utf8[s_][ucols.index(c)]
                d.values = buf.data_ptr() + base[s_] + lay[s_][c][2]
                vals.append(ctx.wrap_device(d, keepalive=buf))
        # concat = the owned copy: the torch receive buffer dies with this function, the column may outlive it inside a C++ plan
        values = ctx.concat(vals) if vals else ctx.new_null(t, 0, prec, scale)
        if not nullable[c] or not total:
            out_cols.append(values)
            continue
        bools = [ctx.wrap_tensor_bool(buf[base[s] + lay[s][c][1]: base[s] + lay[s][c][1] + vwords(n)], n) for s, n in enumerate(recv_counts) if n]
        validity = ctx.concat(bools) if len(bools) > 1 else bools[0]            # spliced at bit granularity
        vd = values.describe()
        d = capi.ArrayDesc()
        d.type, d.precision, d.scale, d.length, d.null_count = t, prec, scale, total, -1
        d.values, d.validity, d.offsets, d.values_bytes = vd.values, validity.describe().values, vd.offsets, vd.values_bytes
        out_cols.append(ctx.concat([ctx.wrap_device(d, keepalive=(values, validity, buf))]))          # values + spliced validity in one owned array
    if sync:
        ctx.synchronize()                              # the owned copies are complete before the torch buffers are released
    return pp.RecordBatch.from_arrays(ctx, list(names), out_cols)


# ----------------------------------------------------------------------------- the exchange behind the C ABI (dfgpu_exchange)
class Comm:
    """dfgpu_comm (include/dfgpu.h): the communicator the C entry point dfgpu_exchange runs over.  Under torch.distributed's nccl backend it is
    RCCL itself (ncclCommInitRank inside libdfgpu.so; the 128-byte id travels from rank 0 by a torch broadcast); under gloo -- the CPU-launched
    tests and several ranks rehearsing on ONE GPU -- it is a caller-provided transport: two callbacks that move the device buffers through the
    host with torch.distributed."""

    def __init__(self, ctx, group=None, force_callbacks: bool = False):
        import ctypes as C
        import torch
        import torch.distributed as dist
        from . import capi
        self.ctx, self.group, self.lib = ctx, group, ctx.lib
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.h = C.c_void_p()
        if _is_nccl(group) and not force_callbacks:
            ident = torch.zeros(128, dtype=torch.uint8, device="cuda")
            if self.rank == 0:
                buf = C.create_string_buffer(128)
                ctx.check(self.lib.dfgpu_comm_unique_id(buf))
                ident = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).cuda()
            dist.broadcast(ident, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            ctx.check(self.lib.dfgpu_comm_create_rccl(ctx.h, bytes(ident.cpu().numpy().tobytes()), self.rank, self.world, C.byref(self.h)))
            self.kind = "rccl"
            return
        self.kind = "callbacks"
        world, rank = self.world, self.rank
        self.fail_lane, self._lane = None, 0            # tests: make the transport refuse one lane of the next exchange

        def all_gather_host(user, send, nbytes, recv):
            try:
                mine = torch.frombuffer(bytearray(C.string_at(send, nbytes)), dtype=torch.uint8)
                dev = "cuda" if _is_nccl(group) else "cpu"
                outs = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(world)]
                dist.all_gather(outs, mine.to(dev), group=group)
                C.memmove(recv, torch.cat([o.cpu() for o in outs]).numpy().tobytes(), nbytes * world)
                return 0
            except Exception:       # noqa: a callback must not raise through C
                import traceback; traceback.print_exc()
                return 1

        def all_to_all_v(user, send, so, sb, recv, ro, rb):
            try:
                lane = self._lane; self._lane += 1
                if self.fail_lane is not None and lane == self.fail_lane:
                    return 1
                so, sb, ro, rb = ([int(a[i]) for i in range(world)] for a in (so, sb, ro, rb))
                tot_s, tot_r = so[-1] + sb[-1], ro[-1] + rb[-1]
                sd = torch.as_tensor(_DevicePtr(send, tot_s, None), device="cuda") if tot_s else torch.empty(0, dtype=torch.uint8, device="cuda")
                staged = sd if _is_nccl(group) else sd.cpu()
                out = torch.empty(tot_r, dtype=torch.uint8, device=staged.device)
                dist.all_to_all_single(out, staged.contiguous(), output_split_sizes=rb, input_split_sizes=sb, group=group)
                if tot_r:
                    torch.as_tensor(_DevicePtr(recv, tot_r, None), device="cuda").copy_(out)
                torch.cuda.synchronize()
                return 0
            except Exception:       # noqa
                import traceback; traceback.print_exc()
                return 1
        I64P = C.POINTER(C.c_int64)
        self._ag = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)(all_gather_host)
        self._aa = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, I64P, I64P, C.c_void_p, I64P, I64P)(all_to_all_v)

        class VT(C.Structure):
            _fields_ = [("user", C.c_void_p), ("rank", C.c_int32), ("world", C.c_int32), ("all_gather_host", C.c_void_p), ("all_to_all_v", C.c_void_p)]
        self._vt = VT(None, rank, world, C.cast(self._ag, C.c_void_p), C.cast(self._aa, C.c_void_p))
        st = self.lib.dfgpu_comm_create_custom(C.byref(self._vt), C.byref(self.h))
        if st != 0:
            raise capi.DfgpuError(st, "dfgpu_comm_create_custom failed")

    def __del__(self):
        try:
            if self.h is not None and self.h.value:
                self.lib.dfgpu_comm_free(self.h); self.h = None
        except Exception:
            pass

    def exchange(self, keys, cols, ncols: int, mask=None):
        """dfgpu_exchange: hash-partition this rank's rows on `keys`, all-to-all, -> (columns this rank owns afterwards or None when no rank had
        rows, rows sent per rank, rows received per rank).  keys = cols = None on a rank without rows."""
        import ctypes as C
        from .device import Array
        kh = (C.c_void_p * max(1, len(keys or [])))(*[k.h for k in (keys or [])])
        ch = (C.c_void_p * max(1, ncols))(*[c.h for c in (cols or [])])
        out = (C.c_void_p * ncols)()
        counts = (C.c_int64 * (2 * self.world))()
        self._lane = 0
        _beat("dfgpu_exchange (%s transport, %d columns)" % (self.kind, ncols))
        self.ctx.check(self.lib.dfgpu_exchange(self.ctx.h, self.h, kh if keys else None, len(keys or []), ch if cols else None, ncols, mask.h if mask is not None else None, out, counts))
        arrays = [Array(self.ctx, C.c_void_p(out[i])) for i in range(ncols)] if out[0] else None
        return arrays, list(counts[:self.world]), list(counts[self.world:])


def gather_batches(ctx, schema, batch, dst: int = 0, group=None, names: Optional[Sequence[str]] = None):
    """≙ CoalescePartitionsExec / SortPreservingMergeExec input gathering: every rank's batch to rank `dst`."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    parts = [None] * world
    parts[dst] = batch
    return exchange_batches(ctx, schema, parts, group, names=names)


PROGRESS = None         # bench.py's watchdog: called with a label before every collective, so that a stalled job can say where it stopped


def _beat(what):
    if PROGRESS is not None:
        PROGRESS(what)


TIMING = False          # bench.py: bracket every ShuffleExec's collectives with events on torch's current stream (the ctx stream in the bench)


class ShuffleExec:
    """RepartitionExec(Partitioning::Hash(exprs, world_size)) across GPUs: this rank's input rows are
    hash-partitioned on device by the C++ RepartitionExec (dfgpu_hash_partition, same create_hashes on every rank),
    exchanged with one all-to-all per column, and the rows received form this rank's output partition
    (≙ physical-plan/src/repartition/mod.rs:232-294 with the in-process channels replaced by RCCL over xGMI).
    A Python-only node: the C++ plan above it sees a MemoryExec of the received rows (physical_plan._child_handle)."""

    def __init__(self, input, exprs, group=None, native: bool = False):
        """native: run the exchange through the C entry point dfgpu_exchange (one-pass partition of all columns + one grouped collective; RCCL
        under the nccl backend).  For inputs whose columns are all fixed width and whose keys are plain columns -- every rank must make the
        same choice, so it is the plan builder's, not a run-time test."""
        import torch.distributed as dist
        from . import physical_plan as pp
        self.input, self.exprs, self.group = input, list(exprs), group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._pp = pp
        self.bytes_sent = 0
        self.native = native and all(isinstance(e, pp.Column) for e in self.exprs)
        self._comm = None
        self._events = []           # (start, end) torch events around the collectives while TIMING is on (bench: all-to-all time and GB/s per link)

    def schema(self):
        return self.input.schema()

    def children(self):
        return [self.input]

    def output_partitioning(self):
        return self._pp.Partitioning.Hash(self.exprs, 1)      # one local output partition per rank

    def _execute_native(self, context):
        pp = self._pp
        if self._comm is None or self._comm.ctx is not context.ctx:
            self._comm = Comm(context.ctx, self.group)
        local = []
        with context.ctx.deferred_flags():
            for p in range(self.input.output_partitioning().partition_count()):
                local += [b for b in self.input.execute(p, context) if b.num_rows]
            names = self.input.schema().names()        # after execute: the C++ node exists, its schema is the exact one (a rank without rows needs the column count)
            mine = pp.concat_batches(None, local) if local else None
            cols = mine.columns if mine is not None else None
            keys = [cols[e.index] for e in self.exprs] if cols is not None else None
            ev = self._mark()
            got, sent, _ = self._comm.exchange(keys, cols, len(names), None)
            self._mark(ev)
        if cols is not None:
            row_bytes = sum(_WIDTH.get(c.describe().type, 0) for c in cols)
            self.bytes_sent += row_bytes * sum(n for d, n in enumerate(sent) if d != self.rank)
        if got is not None and len(got[0]):
            yield pp.RecordBatch.from_arrays(context.ctx, names, got)

    def execute(self, partition, context):
        if self.native:
            yield from self._execute_native(context)
            return
        pp = self._pp
        rep = pp.RepartitionExec(self.input, pp.Partitioning.Hash(self.exprs, self.world))
        merged, schema = [], None
        with context.ctx.deferred_flags():
            for d in range(self.world):
                bs = [b for b in rep.execute(d, context) if b.num_rows]
                m = pp.concat_batches(None, bs) if bs else None
                if m is not None:
                    schema = m.schema
                    if d != self.rank:
                        self.bytes_sent += sum(_WIDTH.get(f.dtype, 0) for f in schema.fields) * m.num_rows
                merged.append(m)
            ev = self._mark()
            out = exchange_batches(context.ctx, schema, merged, self.group, names=self.input.schema().names())
            self._mark(ev)
        if out.num_rows:
            yield out

    def _mark(self, start=None):
        if not TIMING:
            return None
        import torch
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        if start is not None:
            self._events.append((start, e))
        return e

    def exchange_ms(self) -> float:
        """device time between entering and leaving the exchange (partition gathers excluded in the non-native path), after a synchronize"""
        return sum(a.elapsed_time(b) for a, b in self._events)

    def _agree_schema(self, ctx, schema):
        """A rank with no local rows still has to take part in the collectives with the right column list."""
        import torch.distributed as dist
        objs = [None] * self.world
        mine = None if schema is None else [(f.name, f.dtype, f.precision, f.scale) for f in schema.fields]
        if self.world == 1:
            objs = [mine]
        else:
            dist.all_gather_object(objs, mine, group=self.group)
        ref = next((o for o in objs if o is not None), None)
        if ref is None:
            return self._pp.Schema([])
        return self._pp.Schema([self._pp.Field(n, t, p, s) for n, t, p, s in ref])


class BroadcastExec:
    """PartitionMode::CollectLeft across GPUs: the (small) build side of a HashJoinExec is replicated on every rank
    with one all-gather per column instead of shuffling the (large) probe side -- the reference shares one build table
    between all probe partitions through OnceAsync (joins/hash_join.rs:594-606, joins/utils.rs:736-776); here every rank
    receives every rank's rows (source ranks in order) and builds the same table.  On xGMI an all-gather of B bytes costs
    each GPU (N-1)/N * B over 7 concurrent links, independent of the probe side's size."""

    def __init__(self, input, group=None):
        import torch.distributed as dist
        from . import physical_plan as pp
        self.input, self.group = input, group
        self.world = dist.get_world_size(group)
        self._pp = pp
        self.bytes_sent = 0

    def schema(self):
        return self.input.schema()

    def children(self):
        return [self.input]

    def output_partitioning(self):
        return self._pp.Partitioning.UnknownPartitioning(1)

    def execute(self, partition, context):
        pp = self._pp
        with context.ctx.deferred_flags():
            local = []
            for p in range(self.input.output_partitioning().partition_count()):
                local += [b for b in self.input.execute(p, context) if b.num_rows]
            mine = pp.concat_batches(None, local) if local else None
            if mine is not None:
                self.bytes_sent += sum(_WIDTH.get(f.dtype, 0) for f in mine.schema.fields) * mine.num_rows * (self.world - 1)
            out = exchange_batches(context.ctx, None, [mine] * self.world, self.group, names=self.input.schema().names(), broadcast=True)
        if out.num_rows:
            yield out


def agree_schema(schema, group=None):
    """All ranks must call a collective with the same column list even when some rank holds no rows."""
    import torch.distributed as dist
    from . import physical_plan as pp
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    mine = None if schema is None else [(f.name, f.dtype, f.precision, f.scale) for f in schema.fields]
    objs = [mine]
    if world > 1:
        objs = [None] * world
        dist.all_gather_object(objs, mine, group=group)
    ref = next((o for o in objs if o is not None), None)
    return pp.Schema([pp.Field(n, t, p, s) for n, t, p, s in ref]) if ref else pp.Schema([])
