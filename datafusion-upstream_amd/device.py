"""Thin object wrappers over the C ABI: Context, Array (HBM-resident Arrow column), JoinTable,
GroupValues, GroupsAccumulator.  Names follow the reference types they stand for."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

from . import capi
from .capi import DfgpuError


# memory lent to the library (Context.wrap_device): owner objects by cookie, dropped by the library's release callback
_OWNERS = {}
_OWNER_SEQ = 0


def _release_owner(cookie):
    try:
        _OWNERS.pop(int(cookie), None)
    except Exception:       # interpreter shutdown
        pass


_RELEASE_CB = C.CFUNCTYPE(None, C.c_void_p)(_release_owner)


def lent_memory_owners() -> int:
    """number of owner objects the library still holds a reference to (tests)"""
    return len(_OWNERS)


class Context:
    """One device + one HIP stream (what ExecutionPlan::execute(partition, ..) runs on)."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self.lib = capi.load_library()
        h = C.c_void_p()
        st = self.lib.dfgpu_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(h))
        if st != 0 or not h.value:
            raise DfgpuError(st or 1, f"dfgpu_ctx_create(device={device}) failed: no usable HIP device (there is no CPU fallback)")
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.lib.dfgpu_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, status: int):
        if status != 0:
            msg = self.lib.dfgpu_last_error(self.h)
            raise DfgpuError(status, msg.decode() if msg else "")

    def synchronize(self):
        self.check(self.lib.dfgpu_ctx_synchronize(self.h))

    def set_option(self, key: str, value: int):
        self.check(self.lib.dfgpu_ctx_set_option(self.h, key.encode(), int(value)))

    def get_option(self, key: str) -> int:
        import ctypes
        v = ctypes.c_int64(0)
        self.check(self.lib.dfgpu_ctx_get_option(self.h, key.encode(), ctypes.byref(v)))
        return int(v.value)

    def deferred_flags(self):
        """Context manager: kernel error flags raised inside are checked once, on exit (option "defer_flag_checks"), instead of after
        every C-ABI call -- one stream sync for a sequence of calls (materialising the columns of a batch before an exchange)."""
        import contextlib

        @contextlib.contextmanager
        def region():
            self.set_option("defer_flag_checks", 1)
            try:
                yield
            finally:
                self.set_option("defer_flag_checks", 0)
        return region()

    def profile_enable(self, on: bool = True):
        self.check(self.lib.dfgpu_profile_enable(self.h, int(on)))

    def profile_select(self, kernel_name: Optional[str] = None):
        self.check(self.lib.dfgpu_profile_select(self.h, kernel_name.encode() if kernel_name else None))

    def profile_read(self) -> dict:
        """kernel name -> (launches, total_ms) since the last read; HIP events on this ctx's stream."""
        buf = C.create_string_buffer(1 << 16)
        self.check(self.lib.dfgpu_profile_read(self.h, buf, len(buf)))
        out = {}
        for line in buf.value.decode().splitlines():
            name, cnt, ms = line.rsplit(" ", 2)
            out[name] = (int(cnt), float(ms))
        return out

    @property
    def stream(self) -> int:
        return self.lib.dfgpu_ctx_stream(self.h) or 0

    # ---- arrays
    def _wrap(self, handle: C.c_void_p) -> "Array":
        return Array(self, handle)

    def from_arrow(self, arr) -> "Array":
        """pyarrow.Array -> HBM (Arrow C Data Interface, PCIe copy)."""
        import pyarrow as pa
        if isinstance(arr, pa.ChunkedArray):
            arr = arr.combine_chunks()
        if arr.offset != 0:
            arr = pa.concat_arrays([arr])       # materialise the slice so that offset == 0
        ca, cs = capi.ArrowArray(), capi.ArrowSchema()
        arr._export_to_c(C.addressof(ca), C.addressof(cs))
        out = C.c_void_p()
        try:
            self.check(self.lib.dfgpu_array_import_arrow(self.h, C.byref(ca), C.byref(cs), C.byref(out)))
        finally:
            for s in (ca, cs):
                if s.release:
                    C.CFUNCTYPE(None, C.c_void_p)(s.release)(C.addressof(s))
        return self._wrap(out)

    def wrap_device(self, desc: capi.ArrayDesc, keepalive=None) -> "Array":
        """Zero-copy view of caller-owned device memory.  `keepalive` (the tensors behind the pointers) is handed to the library through
        dfgpu_array_wrap_device_owned: it stays referenced until the last array, slice or C++ plan node over the memory is gone, whatever
        happens to the Python objects meanwhile."""
        out = C.c_void_p()
        if keepalive is None:
            self.check(self.lib.dfgpu_array_wrap_device(self.h, C.byref(desc), C.byref(out)))
            return self._wrap(out)
        global _OWNER_SEQ
        _OWNER_SEQ += 1
        key = _OWNER_SEQ
        _OWNERS[key] = keepalive
        self.check(self.lib.dfgpu_array_wrap_device_owned(self.h, C.byref(desc), C.cast(_RELEASE_CB, C.c_void_p), C.c_void_p(key), C.byref(out)))
        return self._wrap(out)

    def wrap_tensor(self, tensor, dtype: int, precision: int = 0, scale: int = 0) -> "Array":
        """Zero-copy view of a contiguous torch CUDA tensor as a non-null fixed-width column."""
        d = capi.ArrayDesc()
        width = {capi.INT8: 1, capi.UINT8: 1, capi.INT16: 2, capi.UINT16: 2, capi.INT32: 4, capi.UINT32: 4, capi.FLOAT32: 4, capi.DATE32: 4,
                 capi.INT64: 8, capi.UINT64: 8, capi.FLOAT64: 8, capi.DECIMAL128: 16}[dtype]
        nbytes = tensor.numel() * tensor.element_size()
        assert tensor.is_contiguous() and nbytes % width == 0
        d.type, d.precision, d.scale, d.length, d.null_count = dtype, precision, scale, nbytes // width, 0
        d.values = tensor.data_ptr()
        return self.wrap_device(d, keepalive=tensor)

    def wrap_tensor_bool(self, tensor, length: int) -> "Array":
        """Zero-copy view of a uint8 CUDA tensor holding `length` bits (LSB first, padded to 8 bytes) as a Boolean column."""
        d = capi.ArrayDesc()
        assert tensor.is_contiguous() and tensor.numel() * tensor.element_size() >= ((length + 63) // 64) * 8
        d.type, d.length, d.null_count = capi.BOOL, length, 0
        d.values = tensor.data_ptr()
        return self.wrap_device(d, keepalive=tensor)

    def concat(self, arrays: Sequence["Array"]) -> "Array":
        hs, n = capi.handle_array([a.h.value for a in arrays])
        out = C.c_void_p()
        self.check(self.lib.dfgpu_concat(self.h, hs, n, C.byref(out)))
        return self._wrap(out)

    def new_null(self, dtype: int, length: int, precision: int = 0, scale: int = 0) -> "Array":
        out = C.c_void_p()
        self.check(self.lib.dfgpu_array_new_null(self.h, dtype, precision, scale, length, C.byref(out)))
        return self._wrap(out)

    # ---- a1
    def hash_columns(self, cols: Sequence["Array"], seed: int = 0) -> "Array":
        hs, n = capi.handle_array([a.h.value for a in cols])
        out = C.c_void_p()
        self.check(self.lib.dfgpu_hash_columns(self.h, hs, n, seed, C.byref(out)))
        return self._wrap(out)

    # ---- arrow-select
    def take(self, values: "Array", indices: "Array") -> "Array":
        out = C.c_void_p()
        self.check(self.lib.dfgpu_take(self.h, values.h, indices.h, C.byref(out)))
        return self._wrap(out)

    def filter(self, values: "Array", mask: "Array") -> "Array":
        out = C.c_void_p()
        self.check(self.lib.dfgpu_filter(self.h, values.h, mask.h, C.byref(out)))
        return self._wrap(out)

    def mask_to_indices(self, mask: "Array") -> "Array":
        out = C.c_void_p()
        self.check(self.lib.dfgpu_mask_to_indices(self.h, mask.h, C.byref(out)))
        return self._wrap(out)

    # ---- a12
    def binary(self, op: int, lhs: "Array", rhs: "Array", lhs_scalar: bool = False, rhs_scalar: bool = False) -> "Array":
        out = C.c_void_p()
        self.check(self.lib.dfgpu_binary(self.h, op, lhs.h, int(lhs_scalar), rhs.h, int(rhs_scalar), C.byref(out)))
        return self._wrap(out)

    def not_(self, a: "Array") -> "Array":
        out = C.c_void_p()
        self.check(self.lib.dfgpu_not(self.h, a.h, C.byref(out)))
        return self._wrap(out)

    def is_null(self, a: "Array", negate: bool = False) -> "Array":
        out = C.c_void_p()
        self.check(self.lib.dfgpu_is_null(self.h, a.h, int(negate), C.byref(out)))
        return self._wrap(out)

    def negative(self, a: "Array") -> "Array":
        out = C.c_void_p()
        self.check(self.lib.dfgpu_negative(self.h, a.h, C.byref(out)))
        return self._wrap(out)

    def cast(self, a: "Array", to_type: int, precision: int = 0, scale: int = 0) -> "Array":
        out = C.c_void_p()
        self.check(self.lib.dfgpu_cast(self.h, a.h, to_type, precision, scale, C.byref(out)))
        return self._wrap(out)

    def in_list(self, a: "Array", lst: "Array", negated: bool = False) -> "Array":
        out = C.c_void_p()
        self.check(self.lib.dfgpu_in_list(self.h, a.h, lst.h, int(negated), C.byref(out)))
        return self._wrap(out)

    # ---- a13 / a14
    def sort_to_indices(self, cols: Sequence["Array"], descending: Sequence[bool], nulls_first: Sequence[bool], fetch: Optional[int] = None) -> "Array":
        hs, n = capi.handle_array([a.h.value for a in cols])
        out = C.c_void_p()
        self.check(self.lib.dfgpu_sort_to_indices(self.h, hs, bytes(int(bool(x)) for x in descending), bytes(int(bool(x)) for x in nulls_first), n,
                                                  -1 if fetch is None else int(fetch), C.byref(out)))
        return self._wrap(out)

    def sort_to_indices_keys(self, cols: Sequence["Array"], descending: Sequence[bool], nulls_first: Sequence[bool], fetch: Optional[int] = None):
        """(indices, [cols[c] in sorted order, or None when the sort did not produce it as a by-product])."""
        hs, n = capi.handle_array([a.h.value for a in cols])
        out = C.c_void_p(); sk = (C.c_void_p * n)()
        self.check(self.lib.dfgpu_sort_to_indices_keys(self.h, hs, bytes(int(bool(x)) for x in descending), bytes(int(bool(x)) for x in nulls_first), n,
                                                       -1 if fetch is None else int(fetch), C.byref(out), sk))
        return self._wrap(out), [self._wrap(C.c_void_p(sk[i])) if sk[i] else None for i in range(n)]

    def sort_take(self, cols: Sequence["Array"], descending: Sequence[bool], nulls_first: Sequence[bool], payload: Sequence["Array"], fetch: Optional[int] = None):
        """dfgpu_sort_take: (indices, [sorted key columns or None], [payload columns in sorted order or None])."""
        hs, n = capi.handle_array([a.h.value for a in cols])
        ps, m = capi.handle_array([a.h.value for a in payload])
        out = C.c_void_p(); sk = (C.c_void_p * n)(); op = (C.c_void_p * max(1, m))()
        self.check(self.lib.dfgpu_sort_take(self.h, hs, bytes(int(bool(x)) for x in descending), bytes(int(bool(x)) for x in nulls_first), n,
                                            -1 if fetch is None else int(fetch), ps, m, C.byref(out), sk, op))
        return self._wrap(out), [self._wrap(C.c_void_p(sk[i])) if sk[i] else None for i in range(n)], [self._wrap(C.c_void_p(op[i])) if op[i] else None for i in range(m)]

    def hash_partition(self, keys: Sequence["Array"], num_partitions: int):
        hs, n = capi.handle_array([a.h.value for a in keys])
        out = C.c_void_p()
        counts = (C.c_int64 * num_partitions)()
        self.check(self.lib.dfgpu_hash_partition(self.h, hs, n, num_partitions, C.byref(out), counts))
        return self._wrap(out), list(counts)


    def partition_columns(self, keys: Sequence["Array"], num_partitions: int, cols: Sequence[Optional["Array"]], mask: Optional["Array"] = None):
        """dfgpu_partition_columns -> ([column grouped by destination or None], row numbers grouped by destination, counts)"""
        hs, n = capi.handle_array([a.h.value for a in keys])
        ch = (C.c_void_p * max(1, len(cols)))(*[(c.h if c is not None else None) for c in cols])
        oc = (C.c_void_p * max(1, len(cols)))()
        out = C.c_void_p()
        counts = (C.c_int64 * num_partitions)()
        self.check(self.lib.dfgpu_partition_columns(self.h, hs, n, num_partitions, ch, len(cols), mask.h if mask is not None else None, oc, C.byref(out), counts))
        return [self._wrap(C.c_void_p(oc[i])) if oc[i] else None for i in range(len(cols))], self._wrap(out), list(counts)


class Array:
    """Immutable Arrow column in HBM (dfgpu_array)."""

    def __init__(self, ctx: Context, handle: C.c_void_p):
        self.ctx = ctx
        self.h = handle
        self._keepalive = None

    def __del__(self):
        try:
            if self.h is not None and self.h.value:
                self.ctx.lib.dfgpu_array_release(self.h)
                self.h = None
        except Exception:
            pass

    def __len__(self) -> int:
        return self.ctx.lib.dfgpu_array_length(self.h)

    def describe(self) -> capi.ArrayDesc:
        d = capi.ArrayDesc()
        self.ctx.check(self.ctx.lib.dfgpu_array_describe(self.h, C.byref(d)))
        return d

    @property
    def type(self) -> int:
        return self.describe().type

    @property
    def null_count(self) -> int:
        return self.ctx.lib.dfgpu_array_null_count(self.ctx.h, self.h)

    def slice(self, offset: int, length: int) -> "Array":
        out = C.c_void_p()
        self.ctx.check(self.ctx.lib.dfgpu_array_slice(self.ctx.h, self.h, offset, length, C.byref(out)))
        return Array(self.ctx, out)

    def to_arrow(self):
        """HBM -> pyarrow.Array (Arrow C Data Interface)."""
        import pyarrow as pa
        ca, cs = capi.ArrowArray(), capi.ArrowSchema()
        self.ctx.check(self.ctx.lib.dfgpu_array_export_arrow(self.ctx.h, self.h, C.byref(ca), C.byref(cs)))
        return pa.Array._import_from_c(C.addressof(ca), C.addressof(cs))

    def to_numpy(self):
        return self.to_arrow().to_numpy(zero_copy_only=False)


class JoinTable:
    """≙ JoinLeftData { JoinHashMap, batch, visited bitmap } (joins/hash_join.rs:77-118)."""

    def __init__(self, ctx: Context, keys: Sequence[Array], mask: Optional[Array] = None, null_equals_null: bool = False):
        self.ctx = ctx
        hs, n = capi.handle_array([a.h.value for a in keys])
        h = C.c_void_p()
        ctx.check(ctx.lib.dfgpu_join_build(ctx.h, hs, n, mask.h if mask is not None else None, int(null_equals_null), C.byref(h)))
        self.h = h

    def __del__(self):
        try:
            if self.h is not None and self.h.value:
                self.ctx.lib.dfgpu_join_table_free(self.h)
                self.h = None
        except Exception:
            pass

    @property
    def num_rows(self) -> int:
        return self.ctx.lib.dfgpu_join_table_num_rows(self.h)

    @property
    def memory(self) -> int:
        return self.ctx.lib.dfgpu_join_table_memory(self.h)

    def probe(self, keys: Sequence[Array], mask: Optional[Array] = None):
        hs, n = capi.handle_array([a.h.value for a in keys])
        ob, op = C.c_void_p(), C.c_void_p()
        self.ctx.check(self.ctx.lib.dfgpu_join_probe(self.ctx.h, self.h, hs, n, mask.h if mask is not None else None, C.byref(ob), C.byref(op)))
        return Array(self.ctx, ob), Array(self.ctx, op)

    def probe_deferred(self, keys: Sequence[Array], mask: Optional[Array] = None):
        """dfgpu_join_probe_deferred -> (build indices or None, probe indices): None when the table can give the build rows later (`lookup`)"""
        hs, n = capi.handle_array([a.h.value for a in keys])
        ob, op = C.c_void_p(), C.c_void_p()
        self.ctx.check(self.ctx.lib.dfgpu_join_probe_deferred(self.ctx.h, self.h, hs, n, mask.h if mask is not None else None, C.byref(ob), C.byref(op)))
        return (Array(self.ctx, ob) if ob.value else None), Array(self.ctx, op)

    def probe_selection(self, keys: Sequence[Array], mask: Optional[Array] = None) -> Optional[Array]:
        """dfgpu_join_probe_selection -> Boolean column over the probe rows (selected and matching), or None when the table does not answer that way"""
        hs, n = capi.handle_array([a.h.value for a in keys])
        out = C.c_void_p()
        st = self.ctx.lib.dfgpu_join_probe_selection(self.ctx.h, self.h, hs, n, mask.h if mask is not None else None, C.byref(out))
        if st == 4:
            return None
        self.ctx.check(st)
        return Array(self.ctx, out)

    def lookup(self, keys: Sequence[Array], rows: Optional[Array] = None) -> Array:
        """dfgpu_join_lookup: build rows of the probe rows `rows` (UInt32, known to match; None = every row of `keys`)"""
        hs, n = capi.handle_array([a.h.value for a in keys])
        out = C.c_void_p()
        self.ctx.check(self.ctx.lib.dfgpu_join_lookup(self.ctx.h, self.h, hs, n, rows.h if rows is not None else None, C.byref(out)))
        return Array(self.ctx, out)

    def mark_visited(self, build_idx: Array):
        self.ctx.check(self.ctx.lib.dfgpu_join_mark_visited(self.ctx.h, self.h, build_idx.h))

    def final_indices(self, join_type: int) -> Array:
        out = C.c_void_p()
        self.ctx.check(self.ctx.lib.dfgpu_join_final_indices(self.ctx.h, self.h, join_type, C.byref(out)))
        return Array(self.ctx, out)


def join_adjust_indices(ctx: Context, build_idx: Array, probe_idx: Array, range_start: int, range_end: int, join_type: int):
    ob, op = C.c_void_p(), C.c_void_p()
    ctx.check(ctx.lib.dfgpu_join_adjust_indices(ctx.h, build_idx.h, probe_idx.h, range_start, range_end, join_type, C.byref(ob), C.byref(op)))
    return Array(ctx, ob), Array(ctx, op)


def agg_preaggregate(ctx: Context, key, kinds: Sequence[int], values: Sequence[Optional[Array]], mask: Optional[Array] = None, any_order: bool = False, casts: Optional[Sequence[int]] = None):
    """dfgpu_agg_preaggregate(_flags): one batch -> (group keys in first-seen order, [state arrays per aggregate]).  `key`: one Array (-> one key Array back) or a
    sequence of 1..4 Arrays (-> a list of key Arrays back).  any_order (DFGPU_PREAGG_ANY_ORDER): the rows may come back in any order.  casts[i] = capi.FLOAT64: aggregate i's argument is CAST(values[i] AS DOUBLE).  Raises
    DfgpuError(NOT_IMPLEMENTED) for shapes it does not take."""
    n = len(kinds)
    keys = [key] if isinstance(key, Array) else list(key)
    kh = (C.c_void_p * len(keys))(*[k.h for k in keys])
    kk = (C.c_int32 * max(1, n))(*kinds)
    vh = (C.c_void_p * max(1, n))(*[(v.h if v is not None else None) for v in values])
    ok = (C.c_void_p * 4)()
    os_ = (C.c_void_p * (2 * max(1, n)))()
    cc = (C.c_int32 * max(1, n))(*[int(c) for c in casts]) if casts is not None else None
    ctx.check(ctx.lib.dfgpu_agg_preaggregate_flags(ctx.h, kh, len(keys), kk, vh, cc, n, mask.h if mask is not None else None, 1 if any_order else 0, ok, os_))
    states = [[Array(ctx, C.c_void_p(os_[2 * i + j])) for j in range(2) if os_[2 * i + j]] for i in range(n)]
    out_keys = [Array(ctx, C.c_void_p(ok[c])) for c in range(len(keys))]
    return (out_keys[0] if isinstance(key, Array) else out_keys), states


class GroupValues:
    """≙ trait GroupValues (aggregates/group_values/mod.rs:35-53)."""

    def __init__(self, ctx: Context, nkeys: int):
        self.ctx, self.nkeys = ctx, nkeys
        h = C.c_void_p()
        ctx.check(ctx.lib.dfgpu_groups_new(ctx.h, nkeys, C.byref(h)))
        self.h = h

    def __del__(self):
        try:
            if self.h is not None and self.h.value:
                self.ctx.lib.dfgpu_groups_free(self.h)
                self.h = None
        except Exception:
            pass

    def intern(self, cols: Sequence[Array], mask: Optional[Array] = None, deferred: bool = False) -> Array:
        """`deferred`: dfgpu_groups_intern_deferred -- the ids may only be handed to accumulators (or exported), see include/dfgpu.h."""
        hs, n = capi.handle_array([a.h.value for a in cols])
        out = C.c_void_p()
        fn = self.ctx.lib.dfgpu_groups_intern_deferred if deferred else self.ctx.lib.dfgpu_groups_intern
        self.ctx.check(fn(self.ctx.h, self.h, hs, n, mask.h if mask is not None else None, C.byref(out)))
        a = Array(self.ctx, out)
        a._keepalive = list(cols)
        return a

    def __len__(self) -> int:
        return self.ctx.lib.dfgpu_groups_len(self.h)

    def size(self) -> int:
        return self.ctx.lib.dfgpu_groups_size(self.h)

    def emit_first(self, n: int) -> List[Array]:
        """EmitTo::First(n): the first n groups leave, the others are renumbered from 0"""
        out = (C.c_void_p * self.nkeys)()
        self.ctx.check(self.ctx.lib.dfgpu_groups_emit_first(self.ctx.h, self.h, n, out))
        return [Array(self.ctx, C.c_void_p(out[i])) for i in range(self.nkeys)]

    def emit(self) -> List[Array]:
        outs = (C.c_void_p * self.nkeys)()
        self.ctx.check(self.ctx.lib.dfgpu_groups_emit(self.ctx.h, self.h, outs))
        return [Array(self.ctx, C.c_void_p(outs[i])) for i in range(self.nkeys)]

    def emit_deferred(self):
        """dfgpu_groups_emit_deferred -> (key columns, first rows) while the keys of a first clustered batch are not gathered yet, else None"""
        out = (C.c_void_p * self.nkeys)(); rows = C.c_void_p()
        st = self.ctx.lib.dfgpu_groups_emit_deferred(self.ctx.h, self.h, out, C.byref(rows))
        if st == 4:
            return None
        self.ctx.check(st)
        return [Array(self.ctx, C.c_void_p(out[i])) for i in range(self.nkeys)], Array(self.ctx, rows)


class GroupsAccumulator:
    """≙ trait GroupsAccumulator (expr/src/groups_accumulator.rs:78-164)."""

    def __init__(self, ctx: Context, kind: int, in_type: int, precision: int = 0, scale: int = 0):
        self.ctx, self.kind = ctx, kind
        h = C.c_void_p()
        ctx.check(ctx.lib.dfgpu_acc_new(ctx.h, kind, in_type, precision, scale, C.byref(h)))
        self.h = h

    def __del__(self):
        try:
            if self.h is not None and self.h.value:
                self.ctx.lib.dfgpu_acc_free(self.h)
                self.h = None
        except Exception:
            pass

    def update_batch(self, values: Optional[Array], group_ids: Array, opt_filter: Optional[Array], total_num_groups: int):
        self.ctx.check(self.ctx.lib.dfgpu_acc_update_batch(self.ctx.h, self.h, values.h if values is not None else None, group_ids.h,
                                                           opt_filter.h if opt_filter is not None else None, total_num_groups))

    @staticmethod
    def update_batch_multi(ctx: "Context", accs: Sequence["GroupsAccumulator"], values: Sequence[Optional[Array]], filters: Sequence[Optional[Array]], group_ids: Array, total_num_groups: int):
        """All update_batch calls of one input batch at once (dfgpu_acc_update_batch_multi): same results, shared passes where possible."""
        n = len(accs)
        ah = (C.c_void_p * n)(*[a.h for a in accs])
        vh = (C.c_void_p * n)(*[(v.h if v is not None else None) for v in values])
        fh = (C.c_void_p * n)(*[(f.h if f is not None else None) for f in filters])
        ctx.check(ctx.lib.dfgpu_acc_update_batch_multi(ctx.h, ah, vh, fh, n, group_ids.h, total_num_groups))

    @staticmethod
    def update_batch_fused(ctx: "Context", accs: Sequence["GroupsAccumulator"], acc_nodes: Sequence[int], nodes: Sequence[tuple], cols: Sequence[Array],
                           group_ids: Array, opt_filter: Optional[Array], total_num_groups: int):
        """dfgpu_acc_update_batch_fused: nodes[k] = (op, lhs, rhs) with op "column" / "scalar" (lhs = index into cols) or "+", "-", "*" over
        earlier nodes; acc_nodes[i] = argument node of accumulator i (-1 for COUNT(*)).  Raises DfgpuError(NOT_IMPLEMENTED) for shapes it
        does not take; nothing has been accumulated then."""
        ops = {"column": capi.NODE_COLUMN, "scalar": capi.NODE_SCALAR, "+": 0, "-": 1, "*": 2}
        n = len(accs)
        ah = (C.c_void_p * n)(*[a.h for a in accs])
        an = (C.c_int32 * n)(*acc_nodes)
        nd = (C.c_int32 * (3 * max(1, len(nodes))))(*[x for op, l, r in nodes for x in (ops[op], l, r)])
        ch = (C.c_void_p * max(1, len(cols)))(*[c.h for c in cols])
        ctx.check(ctx.lib.dfgpu_acc_update_batch_fused(ctx.h, ah, an, n, nd, len(nodes), ch, len(cols), group_ids.h, opt_filter.h if opt_filter is not None else None, total_num_groups))

    def merge_batch(self, states: Sequence[Array], group_ids: Array, opt_filter: Optional[Array], total_num_groups: int):
        hs, n = capi.handle_array([a.h.value for a in states])
        self.ctx.check(self.ctx.lib.dfgpu_acc_merge_batch(self.ctx.h, self.h, hs, n, group_ids.h, opt_filter.h if opt_filter is not None else None, total_num_groups))

    def emit_first(self, n: int, as_state: bool = False) -> List[Array]:
        """evaluate / state with EmitTo::First(n)"""
        out = (C.c_void_p * 2)(); k = C.c_int32()
        self.ctx.check(self.ctx.lib.dfgpu_acc_emit_first(self.ctx.h, self.h, n, 1 if as_state else 0, out, C.byref(k)))
        return [Array(self.ctx, C.c_void_p(out[i])) for i in range(k.value)]

    def evaluate(self) -> Array:
        out = C.c_void_p()
        self.ctx.check(self.ctx.lib.dfgpu_acc_evaluate(self.ctx.h, self.h, C.byref(out)))
        return Array(self.ctx, out)

    def state(self) -> List[Array]:
        outs = (C.c_void_p * 2)()
        n = C.c_int32()
        self.ctx.check(self.ctx.lib.dfgpu_acc_state(self.ctx.h, self.h, outs, C.byref(n)))
        return [Array(self.ctx, C.c_void_p(outs[i])) for i in range(n.value)]

    def size(self) -> int:
        return self.ctx.lib.dfgpu_acc_size(self.h)
