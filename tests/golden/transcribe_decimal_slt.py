"""Writes decimal_slt.json: the decimal_simple table (datafusion/core/tests/data/decimal_data.csv, 15 rows) and the known answers
that datafusion/sqllogictest/test_files/decimal.slt asserts over it, transcribed BY HAND (data only: inputs, expected Arrow
type strings, expected value lists), each case citing the .slt lines it comes from.
Literals: the SQL planner's coercion of `1` / `20` to Decimal128(20,0) (expr/src/type_coercion/binary.rs:524-538) is recorded in
the case; float literals compared with a decimal column are given as exact decimals of the column's scale (same truth value,
the coercion itself belongs to the planner, outside the path).   Run: python transcribe_decimal_slt.py"""
import json

REF = "datafusion/sqllogictest/test_files/decimal.slt:"
table = {
    "ref": "datafusion/core/tests/data/decimal_data.csv + decimal.slt:39-49",
    "columns": {
        "c1": {"type": "decimal128", "precision": 10, "scale": 6, "unscaled": [10, 20, 20, 30, 30, 30, 40, 40, 40, 40, 50, 50, 50, 50, 50]},
        "c2": {"type": "float64", "values": [1e-12, 2e-12, 2e-12, 3e-12, 3e-12, 3e-12, 4e-12, 4e-12, 4e-12, 4e-12, 5e-12, 5e-12, 5e-12, 5e-12, 5e-12]},
        "c3": {"type": "int64", "values": [1, 2, 3, 4, 5, 5, 5, 12, 14, 8, 9, 4, 8, 100, 1]},
        "c4": {"type": "bool", "values": [True, True, False, True, False, True, True, False, True, False, True, True, False, True, False]},
        "c5": {"type": "decimal128", "precision": 12, "scale": 7, "unscaled": [140, 250, 190, 320, 350, 110, 440, 400, 400, 440, 520, 780, 330, 680, 1000]},
    },
}
col = lambda n: {"column": n}
dec = lambda unscaled, p, s: {"literal": {"type": "decimal128", "precision": p, "scale": s, "unscaled": unscaled}}
L = lambda text: text.split()
cases = []

def binary(name, ref, op, l, r, expected_type, expected):
    cases.append({"kind": "binary", "name": name, "ref": REF + ref, "op": op, "left": l, "right": r, "expected_type": expected_type, "expected_rowsort": L(expected)})

binary("c1_plus_1", "208-231", "+", col("c1"), dec(1, 20, 0), "Decimal128(27, 6)",
       "1.00001 1.00002 1.00002 1.00003 1.00003 1.00003 1.00004 1.00004 1.00004 1.00004 1.00005 1.00005 1.00005 1.00005 1.00005")
binary("c1_plus_c5", "234-258", "+", col("c1"), col("c5"), "Decimal128(13, 7)",
       "0.000024 0.000039 0.000041 0.000045 0.000062 0.000065 0.00008 0.00008 0.000083 0.000084 0.000084 0.000102 0.000118 0.000128 0.00015")
binary("c1_minus_1", "261-284", "-", col("c1"), dec(1, 20, 0), "Decimal128(27, 6)",
       "-0.99995 -0.99995 -0.99995 -0.99995 -0.99995 -0.99996 -0.99996 -0.99996 -0.99996 -0.99997 -0.99997 -0.99997 -0.99998 -0.99998 -0.99999")
binary("c1_minus_c5", "287-310", "-", col("c1"), col("c5"), "Decimal128(13, 7)",
       "-0.000002 -0.000002 -0.000004 -0.000004 -0.000004 -0.000005 -0.000005 -0.000018 -0.000028 -0.00005 0 0 0.000001 0.000017 0.000019")
binary("c1_times_20", "313-336", "*", col("c1"), dec(20, 20, 0), "Decimal128(31, 6)",
       "0.0002 0.0004 0.0004 0.0006 0.0006 0.0006 0.0008 0.0008 0.0008 0.0008 0.001 0.001 0.001 0.001 0.001")
binary("c1_times_c5", "339-362", "*", col("c1"), col("c5"), "Decimal128(23, 13)",
       "0.00000000014 0.00000000033 0.00000000038 0.0000000005 0.00000000096 0.00000000105 0.0000000016 0.0000000016 0.00000000165 0.00000000176 "
       "0.00000000176 0.0000000026 0.0000000034 0.0000000039 0.000000005")
binary("c1_div_dec5_5", "365-388", "/", col("c1"), dec(1, 5, 5), "Decimal128(19, 10)", "1 2 2 3 3 3 4 4 4 4 5 5 5 5 5")
binary("c1_div_c5", "391-414", "/", col("c1"), col("c5"), "Decimal128(21, 10)",
       "0.5 0.641025641 0.7142857142 0.7352941176 0.8 0.8571428571 0.909090909 0.909090909 0.9375 0.9615384615 1 1 1.0526315789 1.5151515151 2.7272727272")
binary("c5_mod_dec5_5", "417-440", "%", col("c5"), dec(1, 5, 5), "Decimal128(7, 7)",
       "0 0 0 0.000001 0.000002 0.000002 0.000003 0.000004 0.000004 0.000004 0.000005 0.000005 0.000008 0.000008 0.000009")
binary("c1_mod_c5", "443-466", "%", col("c1"), col("c5"), "Decimal128(11, 7)",
       "0 0 0.000001 0.000008 0.00001 0.000017 0.00002 0.00003 0.00003 0.00004 0.00004 0.00005 0.00005 0.00005 0.00005")

T, F = True, False
ALL = ["c1", "c2", "c3", "c4", "c5"]
ROWS = lambda text: [line.split() for line in text.strip().splitlines()]

def filt(name, ref, op, l, r, select, expected, cast_left=None):
    """WHERE <l op r>; expected rows exactly as the .slt prints them (`rowsort`: compared as a multiset of parsed values)"""
    cases.append({"kind": "filter", "name": name, "ref": REF + ref, "op": op, "left": l, "right": r, "cast_left": cast_left, "select": select, "expected_rowsort": ROWS(expected)})

filt("c1_gt_0.00003", "77-88", ">", col("c1"), dec(30, 10, 6), ["c1"], """
0.00004
0.00004
0.00004
0.00004
0.00005
0.00005
0.00005
0.00005
0.00005""")
# comparison coercion of Decimal128(10,6) with Decimal128(12,7) -> Decimal128(12,7) (get_comparison_common_decimal_type)
filt("c1_gt_c5", "91-96", ">", col("c1"), col("c5"), ALL, """
0.00002 0.000000000002 3 false 0.000019
0.00003 0.000000000003 5 true 0.000011
0.00005 0.000000000005 8 false 0.000033""", cast_left={"precision": 12, "scale": 7})
# c1 = CAST(0.00002 AS Decimal(10,8)): common type Decimal128(12,8)
filt("c1_eq_cast_dec10_8", "133-137", "=", col("c1"), dec(2000, 12, 8), ALL, """
0.00002 0.000000000002 2 true 0.000025
0.00002 0.000000000002 3 false 0.000019""", cast_left={"precision": 12, "scale": 8})
filt("c1_ne_0.00002", "140-155", "!=", col("c1"), dec(20, 10, 6), ["c2", "c3"], """
0.000000000001 1
0.000000000003 4
0.000000000003 5
0.000000000003 5
0.000000000004 12
0.000000000004 14
0.000000000004 5
0.000000000004 8
0.000000000005 1
0.000000000005 100
0.000000000005 4
0.000000000005 8
0.000000000005 9""")
filt("0.00002_gt_c1", "158-161", ">", dec(20, 10, 6), col("c1"), ALL, """
0.00001 0.000000000001 1 true 0.000014""")
filt("c1_le_0.00002", "164-169", "<=", col("c1"), dec(20, 10, 6), ALL, """
0.00001 0.000000000001 1 true 0.000014
0.00002 0.000000000002 2 true 0.000025
0.00002 0.000000000002 3 false 0.000019""")
filt("c1_gt_0.00002", "172-186", ">", col("c1"), dec(20, 10, 6), ALL, """
0.00003 0.000000000003 4 true 0.000032
0.00003 0.000000000003 5 false 0.000035
0.00003 0.000000000003 5 true 0.000011
0.00004 0.000000000004 12 false 0.00004
0.00004 0.000000000004 14 true 0.00004
0.00004 0.000000000004 5 true 0.000044
0.00004 0.000000000004 8 false 0.000044
0.00005 0.000000000005 1 false 0.0001
0.00005 0.000000000005 100 true 0.000068
0.00005 0.000000000005 4 true 0.000078
0.00005 0.000000000005 8 false 0.000033
0.00005 0.000000000005 9 true 0.000052""")
filt("c1_ge_0.00002", "189-205", ">=", col("c1"), dec(20, 10, 6), ALL, """
0.00002 0.000000000002 2 true 0.000025
0.00002 0.000000000002 3 false 0.000019
0.00003 0.000000000003 4 true 0.000032
0.00003 0.000000000003 5 false 0.000035
0.00003 0.000000000003 5 true 0.000011
0.00004 0.000000000004 12 false 0.00004
0.00004 0.000000000004 14 true 0.00004
0.00004 0.000000000004 5 true 0.000044
0.00004 0.000000000004 8 false 0.000044
0.00005 0.000000000005 1 false 0.0001
0.00005 0.000000000005 100 true 0.000068
0.00005 0.000000000005 4 true 0.000078
0.00005 0.000000000005 8 false 0.000033
0.00005 0.000000000005 9 true 0.000052""")

def agg(name, ref, func, column, expected_type, expected, where_c4=None):
    cases.append({"kind": "aggregate", "name": name, "ref": REF + ref, "func": func, "column": column, "where_c4_equals": where_c4, "expected_type": expected_type, "expected": expected})

agg("min_c1_where_c4_false", "99-102", "MIN", "c1", "Decimal128(10, 6)", "0.00002", where_c4=False)
agg("max_c1_where_c4_false", "105-108", "MAX", "c1", "Decimal128(10, 6)", "0.00005", where_c4=False)
agg("sum_c1", "111-116", "SUM", "c1", "Decimal128(20, 6)", "0.00055")
agg("avg_c1", "119-124", "AVG", "c1", "Decimal128(14, 10)", "0.0000366666")

def sort(name, ref, where, keys, fetch, expected):
    """SELECT * WHERE c1 <where> ORDER BY keys [LIMIT fetch]; expected rows in the order the .slt lists them (no rowsort: the key
    order is asserted; rows that tie on every key may come in any order, arrow lexsort is unstable)"""
    cases.append({"kind": "sort", "name": name, "ref": REF + ref, "where": where, "keys": keys, "fetch": fetch, "select": ALL, "expected_ordered": ROWS(expected)})

GE4 = {"op": ">=", "left": col("c1"), "right": dec(40, 10, 6)}
sort("order_by_c1", "495-506", GE4, [["c1", False]], None, """
0.00004 0.000000000004 5 true 0.000044
0.00004 0.000000000004 12 false 0.00004
0.00004 0.000000000004 14 true 0.00004
0.00004 0.000000000004 8 false 0.000044
0.00005 0.000000000005 9 true 0.000052
0.00005 0.000000000005 4 true 0.000078
0.00005 0.000000000005 8 false 0.000033
0.00005 0.000000000005 100 true 0.000068
0.00005 0.000000000005 1 false 0.0001""")
sort("order_by_c1_c3_limit_10", "509-520", GE4, [["c1", False], ["c3", False]], 10, """
0.00004 0.000000000004 5 true 0.000044
0.00004 0.000000000004 8 false 0.000044
0.00004 0.000000000004 12 false 0.00004
0.00004 0.000000000004 14 true 0.00004
0.00005 0.000000000005 1 false 0.0001
0.00005 0.000000000005 4 true 0.000078
0.00005 0.000000000005 8 false 0.000033
0.00005 0.000000000005 9 true 0.000052
0.00005 0.000000000005 100 true 0.000068""")
sort("order_by_c1_c3_limit_5", "522-529", GE4, [["c1", False], ["c3", False]], 5, """
0.00004 0.000000000004 5 true 0.000044
0.00004 0.000000000004 8 false 0.000044
0.00004 0.000000000004 12 false 0.00004
0.00004 0.000000000004 14 true 0.00004
0.00005 0.000000000005 1 false 0.0001""")
sort("order_by_c1_desc", "532-543", GE4, [["c1", True]], None, """
0.00005 0.000000000005 9 true 0.000052
0.00005 0.000000000005 4 true 0.000078
0.00005 0.000000000005 8 false 0.000033
0.00005 0.000000000005 100 true 0.000068
0.00005 0.000000000005 1 false 0.0001
0.00004 0.000000000004 5 true 0.000044
0.00004 0.000000000004 12 false 0.00004
0.00004 0.000000000004 14 true 0.00004
0.00004 0.000000000004 8 false 0.000044""")
sort("order_by_c1_desc_c4", "546-551", {"op": "<", "left": col("c1"), "right": dec(30, 10, 6)}, [["c1", True], ["c4", False]], None, """
0.00002 0.000000000002 3 false 0.000019
0.00002 0.000000000002 2 true 0.000025
0.00001 0.000000000001 1 true 0.000014""")

cases.append({"kind": "groupby", "name": "count_group_by_c1", "ref": REF + "554-561", "keys": ["c1"],
              "expected": [[1, "0.00001"], [2, "0.00002"], [3, "0.00003"], [4, "0.00004"], [5, "0.00005"]]})
cases.append({"kind": "groupby", "name": "count_group_by_c1_c4", "ref": REF + "564-575", "keys": ["c1", "c4"],
              "expected": [[1, "0.00001", T], [1, "0.00002", F], [1, "0.00002", T], [1, "0.00003", F], [2, "0.00003", T], [2, "0.00004", F], [2, "0.00004", T], [2, "0.00005", F], [3, "0.00005", T]]})

# foo(a DECIMAL(38,20), b DECIMAL(38,0)) VALUES (1, 5): a / b = 0.2 (decimal.slt:606-615)
cases.append({"kind": "binary_values", "name": "dec38_20_div_dec38_0", "ref": REF + "606-615", "op": "/",
              "left": {"type": "decimal128", "precision": 38, "scale": 20, "unscaled": [10**20]}, "right": {"type": "decimal128", "precision": 38, "scale": 0, "unscaled": [5]},
              "expected_rowsort": ["0.2"]})

json.dump({"table": table, "cases": cases}, open(__file__.replace("transcribe_decimal_slt.py", "decimal_slt.json"), "w"), indent=1)
print(len(cases), "cases")
