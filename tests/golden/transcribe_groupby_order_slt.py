"""Writes groupby_order_slt.json: self-contained (CREATE TABLE ... AS VALUES / INSERT) cases of the reference's sqllogictest files, transcribed BY HAND
(data only: table rows, GROUP BY keys, aggregate calls with the planner's coerced argument type, expected rows):
  group_by.slt   GROUP BY ALL over tab3 (a NULL key forms its own group), the Dictionary(IntN / UIntN, Utf8) tables: group by the dictionary column,
                 COUNT of a dictionary column
  aggregate.slt  test_decimal_table: COUNT / AVG / SUM / MIN / MAX of Decimal128(5, 2) by an Int32 key and by a Decimal128(5, 1) key with a NULL,
                 an all-NULL Decimal128 argument
  order.slt      ORDER BY num [DESC] [NULLS FIRST | LAST] over (num Int64 with a NULL, letter Utf8)
AVG(Int32) is planned as AVG(CAST(x AS Float64)) (expr/src/type_coercion/aggregates.rs), SUM(Int32) as SUM(CAST(x AS Int64)): the coerced type is part of
each case.   Run: python transcribe_groupby_order_slt.py"""
import json

G = "datafusion/sqllogictest/test_files/group_by.slt:"
A = "datafusion/sqllogictest/test_files/aggregate.slt:"
O = "datafusion/sqllogictest/test_files/order.slt:"
N = None
tables = {
    "tab3": {"ref": G + "1926-1942", "columns": {"col0": {"type": "int32", "values": [0, 0, 0, 0, 1]}, "col1": {"type": "int32", "values": [1, 2, 1, 2, N]},
                                                  "col2": {"type": "int32", "values": [12, 13, 10, 15, 10]}, "col3": {"type": "int32", "values": [-1, -1, -2, -2, -2]}}},
    "test_decimal_table": {"ref": A + "2395-2401", "columns": {
        "c1": {"type": "int32", "values": [1, 1, 2, 2, 3, 3]},
        "c2": {"type": {"decimal128": [5, 2]}, "values": [1010, 2020, 1010, 2020, 1010, 1010]},
        "c3": {"type": {"decimal128": [5, 1]}, "values": [1001, 2002, 7001, 7001, 1001, N]},
        "c4": {"type": {"decimal128": [5, 1]}, "values": [N, N, N, N, N, N]}}},
}
D = lambda p, s: {"decimal128": [p, s]}
cases = [
    {"name": "group_by_all_avg", "ref": G + "1944-1949", "table": "tab3", "group_by": ["col1", "col0"], "aggs": [["AVG", "col2", "float64"]],
     "expected_rowsort": [[1, 0, 11.0], [2, 0, 14.0], [N, 1, 10.0]]},
    {"name": "group_by_all_count_sum", "ref": G + "1976-1981", "table": "tab3", "group_by": ["col0", "col1"], "aggs": [["COUNT", "col2", "int32"], ["SUM", "col3", "int64"]],
     "expected_rowsort": [[0, 1, 2, -3], [0, 2, 2, -3], [1, N, 1, -2]]},
    {"name": "aggregate_decimal_with_group_by", "ref": A + "2403-2409", "table": "test_decimal_table", "group_by": ["c1"],
     "aggs": [["COUNT", "c2", D(5, 2)], ["AVG", "c2", D(5, 2)], ["SUM", "c2", D(5, 2)], ["MIN", "c2", D(5, 2)], ["MAX", "c2", D(5, 2)], ["COUNT", "c3", D(5, 1)], ["COUNT", "c4", D(5, 1)], ["SUM", "c4", D(5, 1)]],
     "expected_rowsort": [[1, 2, "15.15", "30.3", "10.1", "20.2", 2, 0, N], [2, 2, "15.15", "30.3", "10.1", "20.2", 2, 0, N], [3, 2, "10.1", "20.2", "10.1", "10.1", 1, 0, N]]},
    {"name": "aggregate_decimal_with_group_by_decimal", "ref": A + "2411-2418", "table": "test_decimal_table", "group_by": ["c3"],
     "aggs": [["COUNT", "c2", D(5, 2)], ["AVG", "c2", D(5, 2)], ["SUM", "c2", D(5, 2)], ["MIN", "c2", D(5, 2)], ["MAX", "c2", D(5, 2)], ["COUNT", "c4", D(5, 1)], ["SUM", "c4", D(5, 1)]],
     "expected_rowsort": [["100.1", 2, "10.1", "20.2", "10.1", "10.1", 0, N], ["200.2", 1, "20.2", "20.2", "20.2", "20.2", 0, N], ["700.1", 2, "15.15", "30.3", "10.1", "20.2", 0, N], [N, 1, "10.1", "10.1", "10.1", "10.1", 0, N]]},
]
# the eight dictionary tables (group_by.slt:4583-4887): same six rows, key type differs
lines = {"int8": "4583-4616", "int16": "4621-4654", "int32": "4659-4692", "int64": "4697-4730", "uint8": "4735-4768", "uint16": "4773-4806", "uint32": "4811-4844", "uint64": "4849-4882"}
for kt, ref in lines.items():
    tables[f"{kt}_dict"] = {"ref": G + ref, "columns": {"column1": {"type": "int64", "values": [1, 2, 2, 4, 1, 1]},
                                                         "column2": {"dict": {"keys_type": kt, "values": ["A", "B", "A", "A", "C", "A"]}}}}
    cases.append({"name": f"group_by_{kt}_dictionary_column", "ref": G + ref, "table": f"{kt}_dict", "group_by": ["column2"], "aggs": [["COUNT", "column1", "int64"]],
                  "expected_rowsort": [["A", 4], ["B", 1], ["C", 1]]})
    cases.append({"name": f"count_of_{kt}_dictionary_column", "ref": G + ref, "table": f"{kt}_dict", "group_by": ["column1"], "aggs": [["COUNT", "column2", "utf8"]],
                  "expected_rowsort": [[1, 3], [2, 2], [4, 1]]})

order_table = {"ref": O + "65", "columns": {"num": {"type": "int64", "values": [1, 2, N]}, "letter": {"type": "utf8", "values": ["one", "two", "three"]}}}
order = [
    {"name": "test_nulls_first_asc", "ref": O + "62-69", "descending": False, "nulls_first": False, "expected": [[1, "one"], [2, "two"], [N, "three"]]},
    {"name": "test_nulls_first_desc", "ref": O + "71-78", "descending": True, "nulls_first": True, "expected": [[N, "three"], [2, "two"], [1, "one"]]},
    {"name": "test_specific_nulls_last_desc", "ref": O + "80-87", "descending": True, "nulls_first": False, "expected": [[2, "two"], [1, "one"], [N, "three"]]},
    {"name": "test_specific_nulls_first_asc", "ref": O + "89-95", "descending": False, "nulls_first": True, "expected": [[N, "three"], [1, "one"], [2, "two"]]},
]
json.dump({"tables": tables, "cases": cases, "order": {"table": order_table, "cases": order}}, open(__file__.rsplit("/", 1)[0] + "/groupby_order_slt.json", "w"), indent=1)
print(len(cases), "group-by cases,", len(order), "order cases")
