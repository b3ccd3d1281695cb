"""Writes the small Parquet fixtures under tests/golden/parquet/ with pyarrow (run once in the build container: python tests/golden/make_parquet_fixtures.py).
Every file holds the same seeded table under a different writer configuration; the expected columns are whatever pyarrow reads back from the
file, so the fixtures pin the device decoder on bytes an independent writer produced.  clickbench_hits_10.parquet in the same directory is not
made here: it is the data file of the reference's own clickbench.slt (core/tests/data/clickbench_hits_10.parquet, written by DuckDB: Snappy,
PLAIN + RLE_DICTIONARY)."""
import decimal
import os

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "parquet")


def table(n, seed=7, null_frac=0.15):
    rng = np.random.default_rng(seed)
    m = lambda: (rng.random(n) < null_frac) if null_frac else None
    words = ["", "a", "BUILDING", "MACHINERY", "déjà vu", "x" * 40, "furniture", "HOUSEHOLD", "0123456789"]
    cols = {
        "i8": pa.array(rng.integers(-128, 128, n).astype(np.int8), mask=m()),
        "u8": pa.array(rng.integers(0, 256, n).astype(np.uint8), mask=m()),
        "i16": pa.array(rng.integers(-2**15, 2**15, n).astype(np.int16), mask=m()),
        "u16": pa.array(rng.integers(0, 2**16, n).astype(np.uint16), mask=m()),
        "i32": pa.array(rng.integers(-2**31, 2**31, n).astype(np.int32), mask=m()),
        "u32": pa.array(rng.integers(0, 2**32, n).astype(np.uint32), mask=m()),
        "i64": pa.array(rng.integers(-2**62, 2**62, n).astype(np.int64), mask=m()),
        "u64": pa.array(rng.integers(0, 2**63, n).astype(np.uint64) * 2 + 1, mask=m()),
        "f32": pa.array(rng.random(n).astype(np.float32), mask=m()),
        "f64": pa.array(rng.standard_normal(n), mask=m()),
        "b": pa.array(rng.random(n) < 0.5, mask=m()),
        "d32": pa.array(rng.integers(8000, 11000, n).astype(np.int32), type=pa.date32(), mask=m()),
        "dec9": pa.array([None if (null_frac and rng.random() < null_frac) else decimal.Decimal(int(v)).scaleb(-2) for v in rng.integers(-10**8, 10**8, n)], type=pa.decimal128(9, 2)),
        "dec15": pa.array([None if (null_frac and rng.random() < null_frac) else decimal.Decimal(int(v)).scaleb(-2) for v in rng.integers(-10**14, 10**14, n)], type=pa.decimal128(15, 2)),
        "dec38": pa.array([None if (null_frac and rng.random() < null_frac) else decimal.Decimal(int(v) * 10**19 + 12345).scaleb(-4) for v in rng.integers(-10**17, 10**17, n)], type=pa.decimal128(38, 4)),
        "low_card": pa.array([words[i] for i in rng.integers(0, len(words), n)], mask=m()),
        "high_card": pa.array([f"row-{i}-{'y' * int(k)}" for i, k in enumerate(rng.integers(0, 30, n))], mask=m()),
        "few_i64": pa.array(rng.integers(0, 5, n).astype(np.int64) * 1000003, mask=m()),
        "required_i64": pa.array(np.arange(n, dtype=np.int64) * 3 - 1000),
    }
    t = pa.table(cols)
    return t.cast(pa.schema([f.with_nullable(False) if f.name == "required_i64" else f for f in t.schema]))


CONFIGS = {
    "plain_uncompressed_v1": dict(use_dictionary=False, compression="none", data_page_version="1.0"),
    "dict_snappy_v1": dict(use_dictionary=True, compression="snappy", data_page_version="1.0"),
    "dict_uncompressed_v2": dict(use_dictionary=True, compression="none", data_page_version="2.0"),
    "plain_snappy_v2": dict(use_dictionary=False, compression="snappy", data_page_version="2.0"),
    "dict_snappy_small_pages": dict(use_dictionary=True, compression="snappy", data_page_size=512, row_group_size=700),
    "dict_fallback_snappy": dict(use_dictionary=True, compression="snappy", dictionary_pagesize_limit=2048, data_page_size=4096),
    "decimal_as_integer": dict(use_dictionary=False, compression="snappy", store_decimal_as_integer=True),
    "dict_zstd_v1": dict(use_dictionary=True, compression="zstd", data_page_version="1.0"),                 # the codec DataFusion's own writer defaults to
    "plain_zstd_v2": dict(use_dictionary=False, compression="zstd", compression_level=9, data_page_version="2.0"),
}

if __name__ == "__main__":
    os.makedirs(HERE, exist_ok=True)
    for name, kw in CONFIGS.items():
        if not os.path.exists(os.path.join(HERE, name + ".parquet")):          # fixtures already committed stay byte for byte what they were
            pq.write_table(table(2500), os.path.join(HERE, name + ".parquet"), **kw)
    pq.write_table(table(1500, seed=11, null_frac=0), os.path.join(HERE, "no_nulls_dict_snappy.parquet"), use_dictionary=True, compression="snappy")
    pq.write_table(table(10).slice(0, 0), os.path.join(HERE, "empty.parquet"))
    t = table(64, seed=3)
    pq.write_table(pa.table({"ts": pa.array(np.arange(64), type=pa.timestamp("us")), "bin": pa.array([b"x"] * 64), "lst": pa.array([[1, 2]] * 64), "ok": t["i64"]}), os.path.join(HERE, "unsupported_columns.parquet"))
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
