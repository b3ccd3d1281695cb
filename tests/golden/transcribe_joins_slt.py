"""Hand transcription of equi-join cases of datafusion/sqllogictest/test_files/joins.slt (reference @ DataFusion 36) as DATA: the
VALUES tables and the expected rows of each query, with the physical operator shape the query plans to (HashJoinExec join type, the
join filter if any, the WHERE predicate on top, the projected columns, the ORDER BY key).  Run once to (re)write joins_slt.json."""
import json
import os

T = "sqllogictest/test_files/joins.slt"
# tables: name -> {"columns": [(name, type)], "rows": [...]}
tables = {
    "t1_a": {"ref": T + ":282-288", "columns": [["t1_id", "int32"], ["t1_name", "utf8"]], "rows": [[11, "a"], [22, "b"], [33, "c"], [44, "d"], [77, "e"]]},
    "t2_a": {"ref": T + ":292-296", "columns": [["t2_id", "int32"], ["t2_name", "utf8"]], "rows": [[11, "z"], [22, "y"], [44, "x"], [55, "w"]]},
    "t1_b": {"ref": T + ":350-358", "columns": [["t1_id", "int32"], ["t1_name", "utf8"]], "rows": [[11, "a"], [22, "b"], [33, "c"], [44, "d"], [77, "e"], [88, None], [99, None]]},
    "t2_b": {"ref": T + ":361-367", "columns": [["t2_id", "int32"], ["t2_name", "utf8"]], "rows": [[11, "z"], [22, None], [44, "x"], [55, "w"], [99, "u"]]},
    "lsaj_t1": {"ref": T + ":59-66", "columns": [["t1_id", "uint32"], ["t1_name", "utf8"], ["t1_int", "uint32"]],
                "rows": [[11, "a", 1], [11, "a", 1], [22, "b", 2], [33, "c", 3], [44, "d", 4], [None, "e", 0]]},
    "lsaj_t2": {"ref": T + ":69-76", "columns": [["t2_id", "uint32"], ["t2_name", "utf8"], ["t2_int", "uint32"]],
                "rows": [[11, "z", 3], [11, "z", 3], [22, "y", 1], [44, "x", 3], [55, "w", 3], [None, "v", 0]]},
    "rsaj_t1": {"ref": T + ":100-106", "columns": [["t1_id", "uint32"], ["t1_name", "utf8"], ["t1_int", "uint32"]],
                "rows": [[11, "a", 1], [22, "b", 2], [33, "c", 3], [44, "d", 4], [None, "e", 0]]},
    "rsaj_t2": {"ref": T + ":109-113", "columns": [["t2_id", "uint32"], ["t2_name", "utf8"]], "rows": [[11, "a"], [11, "x"], [None, None]]},
}
# join output columns are left columns then right columns (semi / anti: one side); "where" = [column index in the join output, "is_null" | "is_not_null"];
# "filter" = JoinFilter over [side, column] pairs: the first compared with the second by "op"
cases = [
    {"name": "left_join_unbalanced", "ref": T + ":300-306", "left": "t1_a", "right": "t2_a", "on": [[0, 0]], "join_type": "Left", "project": [0, 1, 3], "order_by": 0,
     "expected": [[11, "a", "z"], [22, "b", "y"], [33, "c", None], [44, "d", "x"], [77, "e", None]]},
    {"name": "left_join_null_filter", "ref": T + ":373-379", "left": "t1_b", "right": "t2_b", "on": [[0, 0]], "join_type": "Left", "where": [3, "is_null"], "project": [0, 2, 3], "order_by": 0,
     "expected": [[22, 22, None], [33, None, None], [77, None, None], [88, None, None]]},
    {"name": "left_join_null_filter_on_join_column", "ref": T + ":383-388", "left": "t1_b", "right": "t2_b", "on": [[0, 0]], "join_type": "Left", "where": [2, "is_null"], "project": [0, 2, 3], "order_by": 0,
     "expected": [[33, None, None], [77, None, None], [88, None, None]]},
    {"name": "left_join_not_null_filter", "ref": T + ":391-396", "left": "t1_b", "right": "t2_b", "on": [[0, 0]], "join_type": "Left", "where": [3, "is_not_null"], "project": [0, 2, 3], "order_by": 0,
     "expected": [[11, 11, "z"], [44, 44, "x"], [99, 99, "u"]]},
    {"name": "left_join_not_null_filter_on_join_column", "ref": T + ":399-405", "left": "t1_b", "right": "t2_b", "on": [[0, 0]], "join_type": "Left", "where": [2, "is_not_null"], "project": [0, 2, 3], "order_by": 0,
     "expected": [[11, 11, "z"], [22, 22, None], [44, 44, "x"], [99, 99, "u"]]},
    {"name": "right_join_null_filter", "ref": T + ":414-418", "left": "t1_b", "right": "t2_b", "on": [[0, 0]], "join_type": "Right", "where": [1, "is_null"], "project": [0, 1, 2], "order_by": 2,
     "expected": [[None, None, 55], [99, None, 99]]},
    {"name": "right_join_null_filter_on_join_column", "ref": T + ":421-424", "left": "t1_b", "right": "t2_b", "on": [[0, 0]], "join_type": "Right", "where": [0, "is_null"], "project": [0, 1, 2], "order_by": 2,
     "expected": [[None, None, 55]]},
    {"name": "right_join_not_null_filter", "ref": T + ":427-432", "left": "t1_b", "right": "t2_b", "on": [[0, 0]], "join_type": "Right", "where": [1, "is_not_null"], "project": [0, 1, 2], "order_by": 2,
     "expected": [[11, "a", 11], [22, "b", 22], [44, "d", 44]]},
    {"name": "right_join_not_null_filter_on_join_column", "ref": T + ":435-441", "left": "t1_b", "right": "t2_b", "on": [[0, 0]], "join_type": "Right", "where": [0, "is_not_null"], "project": [0, 1, 2], "order_by": 2,
     "expected": [[11, "a", 11], [22, "b", 22], [44, "d", 44], [99, None, 99]]},
    {"name": "full_join_null_filter", "ref": T + ":444-449", "left": "t1_b", "right": "t2_b", "on": [[0, 0]], "join_type": "Full", "where": [1, "is_null"], "project": [0, 1, 2], "order_by": 0,
     "expected": [[88, None, None], [99, None, 99], [None, None, 55]]},
    {"name": "full_join_not_null_filter", "ref": T + ":452-459", "left": "t1_b", "right": "t2_b", "on": [[0, 0]], "join_type": "Full", "where": [1, "is_not_null"], "project": [0, 1, 2], "order_by": 0,
     "expected": [[11, "a", 11], [22, "b", 22], [33, "c", None], [44, "d", 44], [77, "e", None]]},
    {"name": "left_anti_join_null_key_survives", "ref": T + ":1236-1244", "left": "lsaj_t1", "right": "lsaj_t2", "on": [[0, 0]], "join_type": "LeftAnti", "project": [0, 1], "order_by": 0,
     "expected": [[33, "c"], [None, "e"]]},
    {"name": "left_anti_join_with_filter", "ref": T + ":1280-1290", "left": "lsaj_t1", "right": "lsaj_t2", "on": [[0, 0]], "join_type": "LeftAnti",
     "filter": {"columns": [["left", 0]], "op": ">", "literal": 11}, "project": [0, 1], "order_by": 0,
     "expected": [[11, "a"], [11, "a"], [33, "c"], [None, "e"]]},
    {"name": "left_semi_join_duplicates_and_nulls", "ref": T + ":2873-2879", "left": "lsaj_t1", "right": "lsaj_t2", "on": [[0, 0]], "join_type": "LeftSemi", "project": [0, 1], "order_by": 0,
     "expected": [[11, "a"], [11, "a"], [22, "b"], [44, "d"]]},
    {"name": "right_semi_join_with_filter", "ref": T + ":3051-3055", "left": "rsaj_t2", "right": "rsaj_t1", "on": [[0, 0]], "join_type": "RightSemi",
     "filter": {"columns": [["left", 1], ["right", 1]], "op": "!="}, "project": [0, 1, 2], "order_by": 0,
     "expected": [[11, "a", 1]]},
]
out = {"_source": "hand-transcribed from " + T + " (DataFusion 36); ORDER BY is ASC NULLS LAST (the SQL default)", "tables": tables, "cases": cases}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "joins_slt.json"), "w"), indent=1)
print(len(cases), "cases")
