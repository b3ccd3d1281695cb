"""Writes aggregates.json: known answers of the reference's own aggregate unit tests, transcribed BY HAND (data only).
  scalar   one-group results of Sum / Avg / Count / Min / Max over small arrays (physical-expr/src/aggregate/{sum,average,count,
           min_max}.rs `mod tests`).  `coerced` is the argument type after the reference's type coercion (sum: ints -> Int64 /
           UInt64, floats -> Float64; avg: numerics -> Float64; decimals unchanged -- expr/src/type_coercion/aggregates.rs), which
           assert_aggregate applies with try_cast before aggregating (physical-expr/src/expressions/mod.rs:174-202).
  grouped  AggregateExec Partial -> Final over two batches (physical-plan/src/aggregates/mod.rs:1256-1286 data, :1509-1613 AVG,
           :1353-1507 COUNT(1) -- the (a, b) grouping set of that test, which is a plain GROUP BY a, b).
Run: python transcribe_aggregates.py"""
import json

N = None
AG = "datafusion/physical-expr/src/aggregate/"
scalar = []

def s(name, ref, func, typ, values, coerced, expected_type, expected, **kw):
    scalar.append({"name": name, "ref": AG + ref, "func": func, "type": typ, "values": values, "coerced": coerced, "expected_type": expected_type, "expected": expected, **kw})

dec = lambda p, sc: {"decimal128": [p, sc]}
# sum.rs:300-402
s("sum_decimal", "sum.rs:300-317", "SUM", dec(10, 0), [1, 2, 3, 4, 5], dec(10, 0), dec(20, 0), 15)
s("sum_decimal_with_nulls", "sum.rs:319-337", "SUM", dec(35, 0), [1, N, 3, 4, 5], dec(35, 0), dec(38, 0), 13)
s("sum_decimal_all_nulls", "sum.rs:339-357", "SUM", dec(10, 0), [N] * 6, dec(10, 0), dec(20, 0), N)
s("sum_i32", "sum.rs:359-363", "SUM", "int32", [1, 2, 3, 4, 5], "int64", "int64", 15)
s("sum_i32_with_nulls", "sum.rs:365-375", "SUM", "int32", [1, N, 3, 4, 5], "int64", "int64", 13)
s("sum_i32_all_nulls", "sum.rs:377-381", "SUM", "int32", [N, N], "int64", "int64", N)
s("sum_u32", "sum.rs:383-388", "SUM", "uint32", [1, 2, 3, 4, 5], "uint64", "uint64", 15)
s("sum_f32", "sum.rs:390-395", "SUM", "float32", [1.0, 2.0, 3.0, 4.0, 5.0], "float64", "float64", 15.0)
s("sum_f64", "sum.rs:397-402", "SUM", "float64", [1.0, 2.0, 3.0, 4.0, 5.0], "float64", "float64", 15.0)
# average.rs:577-675
s("avg_decimal", "average.rs:577-595", "AVG", dec(10, 0), [1, 2, 3, 4, 5, 6], dec(10, 0), dec(14, 4), 35000)
s("avg_decimal_with_nulls", "average.rs:597-612", "AVG", dec(10, 0), [1, N, 3, 4, 5], dec(10, 0), dec(14, 4), 32500)
s("avg_decimal_all_nulls", "average.rs:614-630", "AVG", dec(10, 0), [N] * 6, dec(10, 0), dec(14, 4), N)
s("avg_i32", "average.rs:632-636", "AVG", "int32", [1, 2, 3, 4, 5], "float64", "float64", 3.0)
s("avg_i32_with_nulls", "average.rs:638-648", "AVG", "int32", [1, N, 3, 4, 5], "float64", "float64", 3.25)
s("avg_i32_all_nulls", "average.rs:650-654", "AVG", "int32", [N, N], "float64", "float64", N)
s("avg_u32", "average.rs:656-661", "AVG", "uint32", [1, 2, 3, 4, 5], "float64", "float64", 3.0)
s("avg_f32", "average.rs:663-668", "AVG", "float32", [1.0, 2.0, 3.0, 4.0, 5.0], "float64", "float64", 3.0)
s("avg_f64", "average.rs:670-675", "AVG", "float64", [1.0, 2.0, 3.0, 4.0, 5.0], "float64", "float64", 3.0)
# count.rs:345-392
s("count_elements", "count.rs:345-349", "COUNT", "int32", [1, 2, 3, 4, 5], "int32", "int64", 5)
s("count_with_nulls", "count.rs:351-362", "COUNT", "int32", [1, 2, N, N, 3, N], "int32", "int64", 3)
s("count_all_nulls", "count.rs:364-370", "COUNT", "bool", [N] * 8, "bool", "int64", 0)
s("count_empty", "count.rs:372-377", "COUNT", "bool", [], "bool", "int64", 0)
s("count_utf8", "count.rs:379-384", "COUNT", "utf8", ["a", "bb", "ccc", "dddd", "ad"], "utf8", "int64", 5)
# min_max.rs:1120-1430
s("min_decimal", "min_max.rs:1120-1162", "MIN", dec(10, 0), [1, 2, 3, 4, 5], dec(10, 0), dec(10, 0), 1)
s("min_decimal_all_nulls", "min_max.rs:1164-1179", "MIN", dec(10, 0), [N] * 6, dec(10, 0), dec(10, 0), N)
s("min_decimal_with_nulls", "min_max.rs:1181-1197", "MIN", dec(10, 0), [1, N, 3, 4, 5], dec(10, 0), dec(10, 0), 1)
s("max_decimal", "min_max.rs:1199-1251", "MAX", dec(10, 0), [1, 2, 3, 4, 5], dec(10, 0), dec(10, 0), 5)
s("max_decimal_scale5", "min_max.rs:1221-1229", "MAX", dec(10, 5), [1, 2, 3, 4, 5], dec(10, 5), dec(10, 5), 5)
s("max_decimal_with_nulls", "min_max.rs:1253-1267", "MAX", dec(10, 0), [1, N, 3, 4, 5], dec(10, 0), dec(10, 0), 5)
s("max_decimal_all_nulls", "min_max.rs:1269-1283", "MIN", dec(10, 0), [N] * 6, dec(10, 0), dec(10, 0), N)
s("max_i32", "min_max.rs:1285-1289", "MAX", "int32", [1, 2, 3, 4, 5], "int32", "int32", 5)
s("min_i32", "min_max.rs:1291-1295", "MIN", "int32", [1, 2, 3, 4, 5], "int32", "int32", 1)
s("max_i32_with_nulls", "min_max.rs:1331-1341", "MAX", "int32", [1, N, 3, 4, 5], "int32", "int32", 5)
s("min_i32_with_nulls", "min_max.rs:1343-1353", "MIN", "int32", [1, N, 3, 4, 5], "int32", "int32", 1)
s("max_i32_all_nulls", "min_max.rs:1355-1359", "MAX", "int32", [N, N], "int32", "int32", N)
s("min_i32_all_nulls", "min_max.rs:1361-1365", "MIN", "int32", [N, N], "int32", "int32", N)
s("max_u32", "min_max.rs:1367-1372", "MAX", "uint32", [1, 2, 3, 4, 5], "uint32", "uint32", 5)
s("min_u32", "min_max.rs:1374-1379", "MIN", "uint32", [1, 2, 3, 4, 5], "uint32", "uint32", 1)
s("max_f32", "min_max.rs:1381-1386", "MAX", "float32", [1.0, 2.0, 3.0, 4.0, 5.0], "float32", "float32", 5.0)
s("min_f32", "min_max.rs:1388-1393", "MIN", "float32", [1.0, 2.0, 3.0, 4.0, 5.0], "float32", "float32", 1.0)
s("max_f64", "min_max.rs:1395-1400", "MAX", "float64", [1.0, 2.0, 3.0, 4.0, 5.0], "float64", "float64", 5.0)
s("min_f64", "min_max.rs:1402-1407", "MIN", "float64", [1.0, 2.0, 3.0, 4.0, 5.0], "float64", "float64", 1.0)
s("min_date32", "min_max.rs:1409-1413", "MIN", "date32", [1, 2, 3, 4, 5], "date32", "date32", 1)
s("max_date32", "min_max.rs:1421-1425", "MAX", "date32", [1, 2, 3, 4, 5], "date32", "date32", 5)

PM = "datafusion/physical-plan/src/aggregates/mod.rs:"
some_data = {"ref": PM + "1256-1286", "a": {"type": "uint32"}, "b": {"type": "float64"},
             "batches": [{"a": [2, 3, 4, 4], "b": [1.0, 2.0, 3.0, 4.0]}, {"a": [2, 3, 3, 4], "b": [1.0, 2.0, 3.0, 4.0]}]}
grouped = [
    {"name": "avg_partial_then_final", "ref": PM + "1509-1613", "group_by": ["a"], "agg": {"func": "AVG", "column": "b", "type": "float64"},
     "partial_columns": ["a", "AVG(b)[count]", "AVG(b)[sum]"], "partial_sorted": [[2, 2, 2.0], [3, 3, 7.0], [4, 3, 11.0]],
     "final_columns": ["a", "AVG(b)"], "final_sorted": [[2, 1.0], [3, 2.3333333333333335], [4, 3.6666666666666665]]},
    {"name": "count1_group_by_a_b", "ref": PM + "1353-1507 (rows of the (a, b) grouping set)", "group_by": ["a", "b"], "agg": {"func": "COUNT", "column": None, "type": "int64"},
     "partial_columns": ["a", "b", "COUNT(1)[count]"], "partial_sorted": [[2, 1.0, 2], [3, 2.0, 2], [3, 3.0, 1], [4, 3.0, 1], [4, 4.0, 2]],
     "final_columns": ["a", "b", "COUNT(1)"], "final_sorted": [[2, 1.0, 2], [3, 2.0, 2], [3, 3.0, 1], [4, 3.0, 1], [4, 4.0, 2]]},
]
# SortExec known answer (physical-plan/src/sorts/sort.rs:1290-1392): Float32 DESC NULLS FIRST, Float64 ASC NULLS LAST; NaN sorts above every number
NAN = "NaN"
sort = [{"name": "lex_sort_by_float", "ref": "datafusion/physical-plan/src/sorts/sort.rs:1290-1392",
         "columns": [{"type": "float32", "values": [NAN, N, N, NAN, 1.0, 1.0, 2.0, 3.0], "descending": True, "nulls_first": True},
                     {"type": "float64", "values": [200.0, 20.0, 10.0, 100.0, NAN, N, N, NAN], "descending": False, "nulls_first": False}],
         "expected": [[N, 10.0], [N, 20.0], [NAN, 100.0], [NAN, 200.0], [3.0, NAN], [2.0, N], [1.0, NAN], [1.0, N]]}]

# ClickBench shapes over the in-tree 10-row sample (datafusion/core/tests/data/clickbench_hits_10.parquet; 5 of its 105 columns, read
# with pyarrow), answers from datafusion/sqllogictest/test_files/clickbench.slt.  COUNT(DISTINCT ..) columns of those queries are
# outside the path and left out; SUM / AVG arguments are coerced as the reference does (Int16 -> Int64 / Float64).
CB = "datafusion/sqllogictest/test_files/clickbench.slt:"
clickbench = {
    "ref": "datafusion/core/tests/data/clickbench_hits_10.parquet",
    "columns": {"RegionID": {"type": "int32", "values": [839, 839, 839, 839, 39, 839, 197, 197, 229, 839]},
                "AdvEngineID": {"type": "int16", "values": [0] * 10},
                "ResolutionWidth": {"type": "int16", "values": [0] * 10},
                "UserID": {"type": "int64", "values": [-2461439046089301801, -2461439046089301801, -2461439046089301801, -2461439046089301801, 376160620089546609,
                                                        427738049800818189, 519640690937130534, 519640690937130534, 7418527520126366595, -2461439046089301801]},
                "SearchPhrase": {"type": "utf8", "values": [""] * 10}},
    "cases": [
        {"name": "q2_sum_count_avg", "ref": CB + "47-50", "group_by": [], "aggs": [["SUM", "AdvEngineID", "int64"], ["COUNT", None, "int64"], ["AVG", "ResolutionWidth", "float64"]],
         "expected_rowsort": [[0, 10, 0.0]]},
        {"name": "q3_avg_userid", "ref": CB + "52-55", "group_by": [], "aggs": [["AVG", "UserID", "float64"]], "expected_rowsort": [[-304548765855551740.0]]},
        {"name": "q10_group_by_region", "ref": CB + "84-90", "group_by": ["RegionID"], "aggs": [["SUM", "AdvEngineID", "int64"], ["COUNT", None, "int64"], ["AVG", "ResolutionWidth", "float64"]],
         "expected_rowsort": [[197, 0, 2, 0.0], [229, 0, 1, 0.0], [39, 0, 1, 0.0], [839, 0, 6, 0.0]]},
        {"name": "q16_group_by_userid", "ref": CB + "112-119", "group_by": ["UserID"], "aggs": [["COUNT", None, "int64"]],
         "expected_rowsort": [[-2461439046089301801, 5], [376160620089546609, 1], [427738049800818189, 1], [519640690937130534, 2], [7418527520126366595, 1]]},
        {"name": "q17_group_by_userid_searchphrase", "ref": CB + "121-128", "group_by": ["UserID", "SearchPhrase"], "aggs": [["COUNT", None, "int64"]],
         "expected_rowsort": [[-2461439046089301801, "", 5], [376160620089546609, "", 1], [427738049800818189, "", 1], [519640690937130534, "", 2], [7418527520126366595, "", 1]]},
    ],
}
json.dump({"clickbench": clickbench, "scalar": scalar, "some_data": some_data, "grouped": grouped, "sort": sort}, open(__file__.replace("transcribe_aggregates.py", "aggregates.json"), "w"), indent=1)
print(len(scalar), "scalar,", len(grouped), "grouped,", len(sort), "sort cases")
