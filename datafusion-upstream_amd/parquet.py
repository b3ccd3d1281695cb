"""Parquet scan -> device columns: Python binding of dfgpu_parquet_* (include/dfgpu.h; kernels in csrc/parquet.hip).

≙ the per-file part of ParquetExec (core/src/datasource/physical_plan/parquet/mod.rs): footer + page headers on the host, page decompression
and decoding on the device.  `ParquetFile.read` returns dfgpu Arrays in HBM; `physical_plan.ParquetExec` is the plan node over the same file."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

from . import capi
from .device import Array, Context
from .operators import Field, Schema


class ParquetFile:
    def __init__(self, ctx: Context, path: Optional[str] = None, data: Optional[bytes] = None, stage_on_device: bool = False, utf8_dictionary: bool = True):
        """path: the file is mapped (stage_on_device also copies the image to HBM once, pages are then decoded in place);
        data: a bytes object holding the whole file (kept alive by this object)."""
        self.ctx, self.lib, self.h = ctx, (ctx.lib if ctx is not None else capi.load_library()), C.c_void_p()
        self._data = self._dev = None
        ch = ctx.h if ctx is not None else None           # ctx None: metadata only (footer parsing needs no device)
        if path is not None:
            self._check(self.lib.dfgpu_parquet_open_file(ch, path.encode(), 1 if stage_on_device else 0, C.byref(self.h)))
        else:
            self._data = data
            buf = C.cast(C.c_char_p(data), C.c_void_p)
            dev = None
            if stage_on_device:
                import torch
                self._dev = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
                dev = C.c_void_p(self._dev.data_ptr())
            self._check(self.lib.dfgpu_parquet_open(ch, buf, len(data), dev, C.byref(self.h)))
        if not utf8_dictionary:
            self.lib.dfgpu_parquet_set_option(self.h, b"utf8_dictionary", 0)

    def _check(self, st: int):
        if st == 0:
            return
        if self.ctx is not None:
            self.ctx.check(st)
        raise capi.DfgpuError(st, "Parquet file could not be opened (no ctx to carry the message)")

    def close(self):
        if self.h is not None and self.h.value:
            self.lib.dfgpu_parquet_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def num_rows(self) -> int:
        return self.lib.dfgpu_parquet_num_rows(self.h)

    @property
    def num_row_groups(self) -> int:
        return self.lib.dfgpu_parquet_num_row_groups(self.h)

    @property
    def num_columns(self) -> int:
        return self.lib.dfgpu_parquet_num_columns(self.h)

    def row_group_rows(self, rg: int) -> int:
        return self.lib.dfgpu_parquet_row_group_rows(self.h, rg)

    def column_names(self) -> List[str]:
        return [self.lib.dfgpu_parquet_column_name(self.h, i).decode() for i in range(self.num_columns)]

    def column_type(self, c: int):
        """-> (type as read, logical value type, precision, scale, nullable); type 0 = outside the device scan"""
        v = [C.c_int32() for _ in range(5)]
        st = self.lib.dfgpu_parquet_column_type(self.h, c, *[C.byref(x) for x in v])
        if st != 0:
            raise capi.DfgpuError(st, f"no column {c}")
        return tuple(x.value for x in v[:4]) + (bool(v[4].value),)

    def schema(self, columns: Optional[Sequence[int]] = None) -> Schema:
        names = self.column_names()
        cols = range(self.num_columns) if columns is None else columns
        out = []
        for c in cols:
            _, vt, p, s, nl = self.column_type(c)
            out.append(Field(names[c], vt, p, s, nl))
        return Schema(out)

    def column_stats(self, rg: int, c: int):
        """-> (min, max, null_count) of an integer / date column's row group; min = max = None without usable statistics"""
        mn, mx, nc, has = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32()
        st = self.lib.dfgpu_parquet_column_stats(self.h, rg, c, C.byref(mn), C.byref(mx), C.byref(nc), C.byref(has))
        if st != 0:
            raise capi.DfgpuError(st, f"no row group {rg} / column {c}")
        return (mn.value, mx.value, nc.value) if has.value else (None, None, nc.value)

    def chunk_bytes(self, rg: int, c: int, uncompressed: bool = False) -> int:
        return self.lib.dfgpu_parquet_column_chunk_bytes(self.h, rg, c, 1 if uncompressed else 0)

    def read(self, first_row_group: int = 0, num_row_groups: Optional[int] = None, columns: Optional[Sequence] = None) -> List[Array]:
        """Decode the columns (leaf indices or names; default all) of a run of row groups into HBM."""
        if num_row_groups is None:
            num_row_groups = self.num_row_groups - first_row_group
        names = self.column_names()
        cols = list(range(self.num_columns)) if columns is None else [names.index(c) if isinstance(c, str) else int(c) for c in columns]
        idx = (C.c_int32 * max(1, len(cols)))(*cols)
        out = (C.c_void_p * max(1, len(cols)))()
        self.ctx.check(self.lib.dfgpu_parquet_read(self.ctx.h, self.h, first_row_group, num_row_groups, idx, len(cols), out))
        return [Array(self.ctx, C.c_void_p(out[i])) for i in range(len(cols))]
