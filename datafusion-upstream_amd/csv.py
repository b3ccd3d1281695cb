"""Delimited text -> device columns: Python binding of dfgpu_csv_read (include/dfgpu.h; kernels in csrc/csv.hip).  ≙ the per-file part of CsvExec
(core/src/datasource/physical_plan/csv.rs): the schema is the caller's, every byte of the file image is parsed on the device."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

from .device import Array, Context


def read_csv(ctx: Context, data, schema: Sequence[Tuple[str, int, int, int]], projection: Optional[Sequence] = None, delimiter: str = ",", quote: str = '"', has_header: bool = True,
             on_device: bool = False, escape: Optional[str] = None) -> List[Array]:
    """data: bytes (host image) or a uint8 CUDA tensor (on_device=True).  schema: (name, DFGPU type, precision, scale) of EVERY file column, in file order;
    projection: names or indices of the wanted columns (default all).  escape: CsvExec::escape (inside quotes, escape + byte = that byte).  Returns one Array per wanted column, in file-column order."""
    names = [s[0] for s in schema]
    cols = sorted(range(len(schema)) if projection is None else [names.index(c) if isinstance(c, str) else int(c) for c in projection])
    idx = (C.c_int32 * len(cols))(*cols)
    types = (C.c_int32 * (3 * len(cols)))(*[v for c in cols for v in schema[c][1:4]])
    out = (C.c_void_p * len(cols))(); rows = C.c_int64()
    if on_device:
        ptr, n = C.c_void_p(data.data_ptr()), data.numel()
    else:
        ptr, n = C.cast(C.c_char_p(data), C.c_void_p), len(data)
    ctx.check(ctx.lib.dfgpu_csv_read(ctx.h, ptr, n, 1 if on_device else 0, ord(delimiter), ord(quote), ord(escape) if escape else 0, 1 if has_header else 0, len(schema), idx, types, len(cols), out, C.byref(rows)))
    return [Array(ctx, C.c_void_p(out[i])) for i in range(len(cols))]
