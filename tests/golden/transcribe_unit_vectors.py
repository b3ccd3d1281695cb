"""Writes unit_vectors.json: input / expected vectors of the reference's operator-level unit tests, transcribed BY HAND (data only):
  binary    physical-expr/src/expressions/binary.rs    plus / minus / multiply / divide / modulus over Int32, dictionary Int32 and dictionary
                                                       Decimal128 columns (array op array, array op scalar), Kleene AND / OR, the six comparisons and
                                                       IS [NOT] DISTINCT FROM over Boolean columns with NULLs, Decimal128 comparisons
  in_list   physical-expr/src/expressions/in_list.rs   IN / NOT IN over Utf8, Int64, Float64 (NaN, -NaN), Boolean, Date32, Decimal128 with and without a NULL in the list
  nested_loop_join physical-plan/src/joins/nested_loop_join.rs   the eight join types with a JoinFilter (rows compared sorted, as assert_batches_sorted_eq does)
  sort_merge_join physical-plan/src/joins/sort_merge_join.rs    Inner / Left / Right / LeftSemi / LeftAnti: rows and row ORDER (assert_batches_eq), NULL keys, descending keys with null_equals_null, inputs in several batches
  sort      physical-plan/src/sorts/sort.rs            test_in_mem_sort (4 x make_partition(100)), test_sort_metadata
  repartition physical-plan/src/repartition/mod.rs     RoundRobinBatch batch counts, Hash row conservation
Mixed-type comparisons of the same tests (Int64 / Float64 column against a Decimal128 literal) go through the planner's coercion and are left out.
Dictionary operands: arrow-arith unpacks dictionaries before computing, the expected arrays are plain.   Run: python transcribe_unit_vectors.py"""
import json

B = "datafusion/physical-expr/src/expressions/binary.rs:"
i32 = lambda v: {"type": "int32", "values": v}
boo = lambda v: {"type": "bool", "values": v}
dec = lambda v, p, s: {"type": {"decimal128": [p, s]}, "values": v}
dct = lambda keys, values: {"dict": {"keys_type": "int8", "keys": keys, "values": values}}
N = None
cases = []


def case(name, ref, op, l, r, expected, ls=False, rs=False):
    cases.append({"name": name, "ref": B + ref, "op": op, "left": l, "right": r, "left_scalar": ls, "right_scalar": rs, "expected": expected})


# ---- Int32 arithmetic
case("plus_op", "1247-1263", "+", i32([1, 2, 3, 4, 5]), i32([1, 2, 4, 8, 16]), i32([2, 4, 7, 12, 21]))
case("minus_op", "1455-1469", "-", i32([1, 2, 4, 8, 16]), i32([1, 2, 3, 4, 5]), i32([0, 0, 1, 4, 11]))
case("minus_op_negative", "1470-1479", "-", i32([1, 2, 3, 4, 5]), i32([1, 2, 4, 8, 16]), i32([0, 0, -1, -4, -11]))
case("multiply_op", "1671-1687", "*", i32([4, 8, 16, 32, 64]), i32([2, 4, 8, 16, 32]), i32([8, 32, 128, 512, 2048]))
case("divide_op", "1877-1893", "/", i32([8, 32, 128, 512, 2048]), i32([2, 4, 8, 16, 32]), i32([4, 8, 16, 32, 64]))
case("modulus_op", "2095-2111", "%", i32([8, 32, 128, 512, 2048]), i32([2, 4, 7, 14, 32]), i32([0, 0, 2, 8, 0]))
for name, ref, op, k, exp in (("plus_op_scalar", "1358-1371", "+", 1, [2, 3, 4, 5, 6]), ("minus_op_scalar", "1574-1587", "-", 1, [0, 1, 2, 3, 4]),
                              ("multiply_op_scalar", "1786-1799", "*", 2, [2, 4, 6, 8, 10]), ("divide_op_scalar", "2004-2017", "/", 2, [0, 1, 1, 2, 2]),
                              ("modulus_op_scalar", "2212-2225", "%", 2, [1, 0, 1, 0, 1])):
    case(name, ref, op, i32([1, 2, 3, 4, 5]), i32([k]), i32(exp), rs=True)

# ---- dictionary(Int8, Int32) operands
DA = dct([0, N, 1, 3, N], i32([1, 2, 3, 4, 5]))
DB = dct([0, 1, 1, 2, 1], i32([1, 2, 4, 8, 16]))
case("plus_op_dict", "1266-1296", "+", DA, DB, i32([2, N, 4, 8, N]))
case("minus_op_dict", "1482-1512", "-", DA, DB, i32([0, N, 0, 0, N]))
case("multiply_op_dict", "1690-1720", "*", DA, DB, i32([1, N, 4, 16, N]))
BUILT5 = dct([0, N, 1, 2, 3], i32([1, 2, 5, 0]))          # PrimitiveDictionaryBuilder: append 1, null, 2, 5, 0
case("divide_op_dict", "1896-1932", "/", BUILT5, DB, i32([1, N, 1, 1, 0]))
case("modulus_op_dict", "2114-2150", "%", BUILT5, DB, i32([0, N, 0, 1, 0]))
BUILT4 = dct([0, N, 1, 2], i32([1, 2, 5]))                # append 1, null, 2, 5
for name, ref, op, k, exp in (("plus_op_dict_scalar", "1374-1405", "+", 1, [2, N, 3, 6]), ("minus_op_dict_scalar", "1590-1621", "-", 1, [0, N, 1, 4]),
                              ("multiply_op_dict_scalar", "1802-1833", "*", 2, [2, N, 4, 10]), ("divide_op_dict_scalar", "2020-2051", "/", 2, [0, N, 1, 2]),
                              ("modules_op_dict_scalar", "2228-2259", "%", 2, [1, N, 0, 1])):
    case(name, ref, op, BUILT4, i32([k]), i32(exp), rs=True)

# ---- dictionary(Int8, Decimal128(10, 0)) operands; value = 123
V = 123
DDA = dct([0, 2, N, 3, 0], dec([V, V + 2, V - 1, V + 1], 10, 0))
DDB = dct([0, N, 3, 2, 2], dec([V + 1, V + 3, V, V + 2], 10, 0))
case("plus_op_dict_decimal", "1299-1355", "+", DDA, DDB, dec([247, N, N, 247, 246], 11, 0))
case("minus_op_dict_decimal", "1515-1571", "-", DDA, DDB, dec([-1, N, N, 1, 0], 11, 0))
case("multiply_op_dict_decimal", "1723-1783", "*", DDA, DDB, dec([15252, N, N, 15252, 15129], 21, 0))
case("divide_op_dict_decimal", "1935-2001", "/", DDA, DDB, dec([9919, N, N, 10081, 10000], 14, 4))
case("modulus_op_dict_decimal", "2153-2209", "%", DDA, DDB, dec([123, N, N, 1, 0], 10, 0))
DDS = dct([0, 2, 1, 3, 0], dec([V, N, V - 1, V + 1], 10, 0))
case("plus_op_dict_scalar_decimal", "1408-1452", "+", DDS, dec([1], 10, 0), dec([V + 1, V, N, V + 2, V + 1], 11, 0), rs=True)
case("minus_op_dict_scalar_decimal", "1624-1668", "-", DDS, dec([1], 10, 0), dec([V - 1, V - 2, N, V, V - 1], 11, 0), rs=True)
case("multiply_op_dict_scalar_decimal", "1836-1874", "*", DDS, dec([2], 10, 0), dec([246, 244, N, 248, 246], 21, 0), rs=True)
case("divide_op_dict_scalar_decimal", "2054-2092", "/", DDS, dec([2], 10, 0), dec([615000, 610000, N, 620000, 615000], 14, 4), rs=True)
case("modulus_op_dict_scalar_decimal", "2262-2300", "%", DDS, dec([2], 10, 0), dec([1, 0, N, 0, 1], 10, 0), rs=True)

# ---- Kleene logic
T, F = True, False
KA = boo([T, F, N, T, F, N, T, F, N]); KB = boo([T, T, T, F, F, F, N, N, N])
case("and_with_nulls_op", "2399-2441", "AND", KA, KB, boo([T, F, N, F, F, F, N, F, N]))
case("or_with_nulls_op", "2444-2486", "OR", KA, KB, boo([T, T, T, T, F, N, T, N, N]))

# ---- Boolean comparisons: a = [T, T, T, N, N, N, F, F, F], b = [T, N, F, T, N, F, T, N, F]
BA = boo([T, T, T, N, N, N, F, F, F]); BB = boo([T, N, F, T, N, F, T, N, F])
for name, ref, op, exp in (("eq_op_bool", "2534-2550", "=", [T, N, F, N, N, N, F, N, T]), ("neq_op_bool", "2593-2609", "!=", [F, N, T, N, N, N, T, N, F]),
                           ("lt_op_bool", "2652-2668", "<", [F, N, F, N, N, N, T, N, F]), ("lt_eq_op_bool", "2715-2731", "<=", [T, N, F, N, N, N, T, N, T]),
                           ("gt_op_bool", "2778-2794", ">", [F, N, T, N, N, N, F, N, F]), ("gt_eq_op_bool", "2841-2857", ">=", [T, N, T, N, N, N, F, N, T]),
                           ("is_distinct_from_op_bool", "2904-2920", "IS DISTINCT FROM", [F, T, T, T, F, T, T, T, F]),
                           ("is_not_distinct_from_op_bool", "2923-2939", "IS NOT DISTINCT FROM", [T, F, F, F, T, F, F, F, T])):
    case(name, ref, op, BA, BB, boo(exp))
# scalar forms over [T, N, F]: (op, scalar, scalar-on-the-left expected, scalar-on-the-right expected)
SA = boo([T, N, F])
for op, ref, rows in (("=", "2553-2590", [(T, [T, N, F], [T, N, F]), (F, [F, N, T], [F, N, T])]), ("!=", "2612-2649", [(T, [F, N, T], [F, N, T]), (F, [T, N, F], [T, N, F])]),
                      ("<", "2671-2712", [(T, [F, N, F], [F, N, T]), (F, [T, N, F], [F, N, F])]), ("<=", "2734-2775", [(T, [T, N, F], [T, N, T]), (F, [T, N, T], [F, N, T])]),
                      (">", "2797-2838", [(T, [F, N, T], [F, N, F]), (F, [F, N, F], [T, N, F])]), (">=", "2860-2901", [(T, [T, N, T], [T, N, F]), (F, [F, N, T], [T, N, T])])):
    for k, left_exp, right_exp in rows:
        case(f"bool_scalar_{k}_{op}_arr", ref, op, boo([k]), SA, boo(left_exp), ls=True)
        case(f"bool_arr_{op}_scalar_{k}", ref, op, SA, boo([k]), boo(right_exp), rs=True)

# ---- Decimal128(25, 3) column against a Decimal128(25, 3) literal; plain and behind a dictionary
DCOL = dec([V, N, V - 1, V + 1], 25, 3); DDCOL = dct([0, N, 2, 3], dec([V, N, V - 1, V + 1], 25, 3))
for op, exp in (("=", [T, N, F, F]), ("!=", [F, N, T, T]), ("<", [F, N, T, F]), ("<=", [T, N, T, F]), (">", [F, N, F, T]), (">=", [T, N, F, T])):
    case(f"comparison_decimal_arr_{op}_scalar", "3081-3155", op, DCOL, dec([V], 25, 3), boo(exp), rs=True)
    case(f"comparison_dict_decimal_arr_{op}_scalar", "2990-3076", op, DDCOL, dec([V], 25, 3), boo(exp), rs=True)
# Decimal128(10, 0) array against array
DL = dec([V, N, V - 1, V + 1], 10, 0); DR = dec([V - 1, V, V + 1, V + 1], 10, 0)
for op, exp in (("=", [F, N, F, T]), ("!=", [T, N, T, F]), ("<", [F, N, T, F]), ("<=", [F, N, T, T]), (">", [T, N, F, F]), (">=", [T, N, F, T])):
    case(f"comparison_decimal_arr_{op}_arr", "3240-3316", op, DL, DR, boo(exp))

# ---- InListExpr (physical-expr/src/expressions/in_list.rs): NULL in the list turns "no match" into NULL; NaN matches NaN of the same sign bit pattern
I = "datafusion/physical-expr/src/expressions/in_list.rs:"
in_list = []


def inl(name, ref, typ, values, lst, negated, expected):
    in_list.append({"name": name, "ref": I + ref, "type": typ, "values": values, "list": lst, "negated": negated, "expected": expected})


for name, ref, typ, vals, l2 in (("utf8", "515-567", "utf8", ["a", "d", N], ["a", "b"]), ("int64", "629-681", "int64", [0, 2, N], [0, 1]), ("date32", "909-975", "date32", [0, 2, N], [0, 1])):
    inl(f"in_list_{name}", ref, typ, vals, l2, False, [T, F, N]); inl(f"not_in_list_{name}", ref, typ, vals, l2, True, [F, T, N])
    inl(f"in_list_{name}_with_null", ref, typ, vals, l2 + [N], False, [T, N, N]); inl(f"not_in_list_{name}_with_null", ref, typ, vals, l2 + [N], True, [F, N, N])
FV = [0.0, 0.2, N, "NaN", "-NaN"]
for lst, neg, exp in (([0.0, 0.1], False, [T, F, N, F, F]), ([0.0, 0.1], True, [F, T, N, T, T]), ([0.0, 0.1, N], False, [T, N, N, N, N]), ([0.0, 0.1, N], True, [F, N, N, N, N]),
                      ([0.0, 0.1, "NaN"], False, [T, F, N, T, F]), ([0.0, 0.1, "NaN"], True, [F, T, N, F, T]), ([0.0, 0.1, "-NaN"], False, [T, F, N, F, T]), ([0.0, 0.1, "-NaN"], True, [F, T, N, T, F])):
    inl(f"{'not_' if neg else ''}in_list_float64_{len(in_list)}", "683-785", "float64", FV, lst, neg, exp)
for lst, neg, exp in (([T], False, [T, N]), ([T], True, [F, N]), ([T, N], False, [T, N]), ([T, N], True, [F, N])):
    inl(f"{'not_' if neg else ''}in_list_bool_{len(in_list)}", "787-839", "bool", [T, N], lst, neg, exp)
# Decimal128(13, 4) column [100.0000, NULL, 200.5000]; the Int32 literals 100 / 200 are cast to the column type by in_list_cast (in_list.rs:459-513)
DT = {"decimal128": [13, 4]}
for lst, neg, exp in (([1000000, 2000000], False, [T, N, F]), ([1000000, 2000000], True, [F, N, T]), ([1000000, N], False, [T, N, N]), ([1000000, N], True, [F, N, N]),
                      ([v * 10000 for v in range(99, 300)], False, [T, N, F]), ([v * 10000 for v in range(99, 300)], True, [F, N, T])):
    inl(f"{'not_' if neg else ''}in_list_decimal_{len(in_list)}", "977-1075", DT, [1000000, N, 2005000], lst, neg, exp)

# ---- NestedLoopJoinExec (physical-plan/src/joins/nested_loop_join.rs:772-1130): left (a1, b1, c1), right (a2, b2, c2), filter left.b1 != 8 AND right.b2 != 10
NL = "datafusion/physical-plan/src/joins/nested_loop_join.rs:"
nlj = {"left": {"ref": NL + "782-788", "columns": {"a1": [5, 9, 11], "b1": [5, 8, 8], "c1": [50, 90, 110]}}, "right": {"ref": NL + "790-796", "columns": {"a2": [12, 2, 10], "b2": [10, 2, 10], "c2": [40, 80, 100]}},
       "filter": {"ref": NL + "798-840", "column_indices": [["left", 1], ["right", 1]], "expression": "x0 != 8 AND x1 != 10"},
       "cases": [
           {"name": "join_inner_with_filter", "ref": NL + "892-917", "join_type": "Inner", "expected_sorted": [[5, 5, 50, 2, 2, 80]]},
           {"name": "join_left_with_filter", "ref": NL + "920-948", "join_type": "Left", "expected_sorted": [[11, 8, 110, N, N, N], [5, 5, 50, 2, 2, 80], [9, 8, 90, N, N, N]]},
           {"name": "join_right_with_filter", "ref": NL + "951-979", "join_type": "Right", "expected_sorted": [[N, N, N, 10, 10, 100], [N, N, N, 12, 10, 40], [5, 5, 50, 2, 2, 80]]},
           {"name": "join_full_with_filter", "ref": NL + "982-1012", "join_type": "Full", "expected_sorted": [[N, N, N, 10, 10, 100], [N, N, N, 12, 10, 40], [11, 8, 110, N, N, N], [5, 5, 50, 2, 2, 80], [9, 8, 90, N, N, N]]},
           {"name": "join_left_semi_with_filter", "ref": NL + "1015-1041", "join_type": "LeftSemi", "expected_sorted": [[5, 5, 50]]},
           {"name": "join_left_anti_with_filter", "ref": NL + "1044-1071", "join_type": "LeftAnti", "expected_sorted": [[11, 8, 110], [9, 8, 90]]},
           {"name": "join_right_semi_with_filter", "ref": NL + "1074-1100", "join_type": "RightSemi", "expected_sorted": [[2, 2, 80]]},
           {"name": "join_right_anti_with_filter", "ref": NL + "1103-1130", "join_type": "RightAnti", "expected_sorted": [[10, 10, 100], [12, 10, 40]]},
       ]}

# ---- SortMergeJoinExec (physical-plan/src/joins/sort_merge_join.rs:1787-2448): rows AND their order are asserted (assert_batches_eq)
SM = "datafusion/physical-plan/src/joins/sort_merge_join.rs:"
T3 = lambda a, b, c: [a, b, c]
smj = [
    {"name": "join_inner_one", "ref": SM + "1787-1818", "join_type": "Inner", "left": T3([1, 2, 3], [4, 5, 5], [7, 8, 9]), "right": T3([10, 20, 30], [4, 5, 6], [70, 80, 90]), "on": [[1, 1]],
     "expected": [[1, 4, 7, 10, 4, 70], [2, 5, 8, 20, 5, 80], [3, 5, 9, 20, 5, 80]]},
    {"name": "join_inner_two", "ref": SM + "1821-1856", "join_type": "Inner", "left": T3([1, 2, 2], [1, 2, 2], [7, 8, 9]), "right": T3([1, 2, 3], [1, 2, 2], [70, 80, 90]), "on": [[0, 0], [1, 1]],
     "expected": [[1, 1, 7, 1, 1, 70], [2, 2, 8, 2, 2, 80], [2, 2, 9, 2, 2, 80]]},
    {"name": "join_inner_with_nulls", "ref": SM + "1898-1933", "join_type": "Inner", "left": T3([1, 1, 2, 2], [N, 1, 2, 2], [1, N, 8, 9]), "right": T3([1, 1, 2, 3], [N, 1, 2, 2], [10, 70, 80, 90]), "on": [[0, 0], [1, 1]],
     "expected": [[1, 1, N, 1, 1, 70], [2, 2, 8, 2, 2, 80], [2, 2, 9, 2, 2, 80]]},
    {"name": "join_inner_with_nulls_with_options", "ref": SM + "1936-1985", "join_type": "Inner", "descending": True, "nulls_first": False, "null_equals_null": True,
     "left": T3([2, 2, 1, 1], [2, 2, 1, N], [9, 8, N, 1]), "right": T3([3, 2, 1, 1], [2, 2, 1, N], [90, 80, 70, 10]), "on": [[0, 0], [1, 1]],
     "expected": [[2, 2, 9, 2, 2, 80], [2, 2, 8, 2, 2, 80], [1, 1, N, 1, 1, 70], [1, N, 1, 1, N, 10]]},
    {"name": "join_left_one", "ref": SM + "2030-2059", "join_type": "Left", "left": T3([1, 2, 3], [4, 5, 7], [7, 8, 9]), "right": T3([10, 20, 30], [4, 5, 6], [70, 80, 90]), "on": [[1, 1]],
     "expected": [[1, 4, 7, 10, 4, 70], [2, 5, 8, 20, 5, 80], [3, 7, 9, N, N, N]]},
    {"name": "join_right_one", "ref": SM + "2062-2091", "join_type": "Right", "left": T3([1, 2, 3], [4, 5, 7], [7, 8, 9]), "right": T3([10, 20, 30], [4, 5, 6], [70, 80, 90]), "on": [[1, 1]],
     "expected": [[1, 4, 7, 10, 4, 70], [2, 5, 8, 20, 5, 80], [N, N, N, 30, 6, 90]]},
    {"name": "join_anti", "ref": SM + "2126-2154", "join_type": "LeftAnti", "left": T3([1, 2, 2, 3, 5], [4, 5, 5, 7, 7], [7, 8, 8, 9, 11]), "right": T3([10, 20, 30], [4, 5, 6], [70, 80, 90]), "on": [[1, 1]],
     "expected": [[3, 7, 9], [5, 7, 11]]},
    {"name": "join_semi", "ref": SM + "2157-2186", "join_type": "LeftSemi", "left": T3([1, 2, 2, 3], [4, 5, 5, 7], [7, 8, 8, 9]), "right": T3([10, 20, 30], [4, 5, 6], [70, 80, 90]), "on": [[1, 1]],
     "expected": [[1, 4, 7], [2, 5, 8], [2, 5, 8]]},
    {"name": "join_left_sort_order", "ref": SM + "2285-2318", "join_type": "Left", "left": T3([0, 1, 2, 3, 4, 5], [3, 4, 5, 6, 6, 7], [4, 5, 6, 7, 8, 9]), "right": T3([0, 10, 20, 30, 40], [2, 4, 6, 6, 8], [50, 60, 70, 80, 90]), "on": [[1, 1]],
     "expected": [[0, 3, 4, N, N, N], [1, 4, 5, 10, 4, 60], [2, 5, 6, N, N, N], [3, 6, 7, 20, 6, 70], [3, 6, 7, 30, 6, 80], [4, 6, 8, 20, 6, 70], [4, 6, 8, 30, 6, 80], [5, 7, 9, N, N, N]]},
    {"name": "join_right_sort_order", "ref": SM + "2321-2350", "join_type": "Right", "left": T3([0, 1, 2, 3], [3, 4, 5, 7], [6, 7, 8, 9]), "right": T3([0, 10, 20, 30], [2, 4, 5, 6], [60, 70, 80, 90]), "on": [[1, 1]],
     "expected": [[N, N, N, 0, 2, 60], [1, 4, 7, 10, 4, 70], [2, 5, 8, 20, 5, 80], [N, N, N, 30, 6, 90]]},
    {"name": "join_left_multiple_batches", "ref": SM + "2353-2399", "join_type": "Left", "left_batches": [3, 4], "right_batches": [3, 2],
     "left": T3([0, 1, 2, 3, 4, 5, 6], [3, 4, 5, 6, 6, 7, 9], [4, 5, 6, 7, 8, 9, 9]), "right": T3([0, 10, 20, 30, 40], [2, 4, 6, 6, 8], [50, 60, 70, 80, 90]), "on": [[1, 1]],
     "expected": [[0, 3, 4, N, N, N], [1, 4, 5, 10, 4, 60], [2, 5, 6, N, N, N], [3, 6, 7, 20, 6, 70], [3, 6, 7, 30, 6, 80], [4, 6, 8, 20, 6, 70], [4, 6, 8, 30, 6, 80], [5, 7, 9, N, N, N], [6, 9, 9, N, N, N]]},
    {"name": "join_right_multiple_batches", "ref": SM + "2402-2448", "join_type": "Right", "left_batches": [3, 2], "right_batches": [3, 4],
     "left": T3([0, 10, 20, 30, 40], [2, 4, 6, 6, 8], [50, 60, 70, 80, 90]), "right": T3([0, 1, 2, 3, 4, 5, 6], [3, 4, 5, 6, 6, 7, 9], [4, 5, 6, 7, 8, 9, 9]), "on": [[1, 1]],
     "expected": [[N, N, N, 0, 3, 4], [10, 4, 60, 1, 4, 5], [N, N, N, 2, 5, 6], [20, 6, 70, 3, 6, 7], [30, 6, 80, 3, 6, 7], [20, 6, 70, 4, 6, 8], [30, 6, 80, 4, 6, 8], [N, N, N, 5, 7, 9], [N, N, N, 6, 9, 9]]},
]

S = "datafusion/physical-plan/src/sorts/sort.rs:"
# JoinType::Full: the reference's two tests compare SORTED rows (assert_batches_sorted_eq), so "expected" is a multiset here ("sorted": True)
smj_full = [
    {"name": "join_full_one", "ref": SM + "2094-2121", "join_type": "Full", "sorted": True, "left": T3([1, 2, 3], [4, 5, 7], [7, 8, 9]), "right": T3([10, 20, 30], [4, 5, 6], [70, 80, 90]), "on": [[1, 1]],
     "expected": [[N, N, N, 30, 6, 90], [1, 4, 7, 10, 4, 70], [2, 5, 8, 20, 5, 80], [3, 7, 9, N, N, N]]},
    {"name": "join_full_multiple_batches", "ref": SM + "2451-2497", "join_type": "Full", "sorted": True, "left_batches": [3, 4], "right_batches": [3, 2],
     "left": T3([0, 1, 2, 3, 4, 5, 6], [3, 4, 5, 6, 6, 7, 9], [4, 5, 6, 7, 8, 9, 9]), "right": T3([0, 10, 20, 30, 40], [2, 4, 6, 6, 8], [50, 60, 70, 80, 90]), "on": [[1, 1]],
     "expected": [[N, N, N, 0, 2, 50], [N, N, N, 40, 8, 90], [0, 3, 4, N, N, N], [1, 4, 5, 10, 4, 60], [2, 5, 6, N, N, N], [3, 6, 7, 20, 6, 70], [3, 6, 7, 30, 6, 80], [4, 6, 8, 20, 6, 70],
                  [4, 6, 8, 30, 6, 80], [5, 7, 9, N, N, N], [6, 9, 9, N, N, N]]},
]
# JoinFilter: sqllogictest/test_files/sort_merge_join.slt (prefer_hash_join = false).  Tables as the file creates them (INT columns widened to Int64, which is what the
# planner's CAST in the file's own EXPLAIN output does before the arithmetic); "filter" is the ON clause's non-equi part as a small tree over ["l", i] / ["r", i] = column i
# of the left / right table; "project" = the SELECT list as indices into left columns ++ right columns; rows compared sorted (the file says rowsort).
SLT = "datafusion/sqllogictest/test_files/sort_merge_join.slt:"
TA1 = {"types": ["utf8", "int64"], "columns": [["Alice", "Alice", "Bob"], [50, 100, 1]]}
TA2 = {"types": ["utf8", "int64"], "columns": [["Alice", "Alice"], [2, 1]]}
TB1 = {"types": ["int64", "utf8", "int64"], "columns": [[11, 22, 33, 44], ["a", "b", "c", "d"], [1, 2, 3, 4]]}
TB2 = {"types": ["int64", "utf8", "int64"], "columns": [[11, 22, 44, 55], ["z", "y", "x", "w"], [3, 1, 3, 3]]}
L, Rr = (lambda i: ["l", i]), (lambda i: ["r", i])
A = "Alice"
smj_filter = [
    {"name": "inner_times_50", "ref": SLT + "51-56", "join_type": "Inner", "left": TA1, "right": TA2, "on": [[0, 0]], "filter": ["<=", ["*", Rr(1), 50], L(1)],
     "expected": [[A, 100, A, 1], [A, 100, A, 2], [A, 50, A, 1]]},
    {"name": "inner_less", "ref": SLT + "58-64", "join_type": "Inner", "left": TA1, "right": TA2, "on": [[0, 0]], "filter": ["<", Rr(1), L(1)],
     "expected": [[A, 100, A, 1], [A, 100, A, 2], [A, 50, A, 1], [A, 50, A, 2]]},
    {"name": "inner_none_pass", "ref": SLT + "66-68", "join_type": "Inner", "left": TA1, "right": TA2, "on": [[0, 0]], "filter": [">", Rr(1), L(1)], "expected": []},
    {"name": "left_times_50", "ref": SLT + "81-88", "join_type": "Left", "left": TA1, "right": TA2, "on": [[0, 0]], "filter": ["<=", ["*", Rr(1), 50], L(1)],
     "expected": [[A, 100, A, 1], [A, 100, A, 2], [A, 50, A, 1], [A, 50, N, N], ["Bob", 1, N, N]]},
    {"name": "left_less", "ref": SLT + "90-97", "join_type": "Left", "left": TA1, "right": TA2, "on": [[0, 0]], "filter": ["<", Rr(1), L(1)],
     "expected": [[A, 100, A, 1], [A, 100, A, 2], [A, 50, A, 1], [A, 50, A, 2], ["Bob", 1, N, N]]},
    {"name": "right_times_50", "ref": SLT + "109-115", "join_type": "Right", "left": TA1, "right": TA2, "on": [[0, 0]], "filter": ["<=", ["*", Rr(1), 50], L(1)],
     "expected": [[A, 100, A, 1], [A, 100, A, 2], [A, 50, A, 1], [N, N, A, 2]]},
    {"name": "right_greater", "ref": SLT + "117-123", "join_type": "Right", "left": TA1, "right": TA2, "on": [[0, 0]], "filter": [">", L(1), Rr(1)],
     "expected": [[A, 100, A, 1], [A, 100, A, 2], [A, 50, A, 1], [A, 50, A, 2]]},
    {"name": "full_times_50", "ref": SLT + "136-146", "join_type": "Full", "left": TA1, "right": TA2, "on": [[0, 0]], "filter": [">", ["*", Rr(1), 50], L(1)],
     "expected": [[A, 100, N, N], [A, 100, N, N], [A, 50, A, 2], [A, 50, N, N], ["Bob", 1, N, N], [N, N, A, 1], [N, N, A, 1], [N, N, A, 2]]},
    {"name": "full_plus_50", "ref": SLT + "148-157", "join_type": "Full", "left": TA1, "right": TA2, "on": [[0, 0]], "filter": [">", L(1), ["+", Rr(1), 50]],
     "expected": [[A, 100, A, 1], [A, 100, A, 2], [A, 50, N, N], [A, 50, N, N], ["Bob", 1, N, N], [N, N, A, 1], [N, N, A, 2]]},
    {"name": "inner_int_ge", "ref": SLT + "180-185", "join_type": "Inner", "left": TB1, "right": TB2, "on": [[0, 0]], "filter": [">=", L(2), Rr(2)], "project": [0, 2, 5],
     "expected": [[22, 2, 1], [44, 4, 3]]},
    {"name": "equijoin_multiple_condition_ordering", "ref": SLT + "187-193", "join_type": "Inner", "left": TB1, "right": TB2, "on": [[0, 0]], "filter": ["!=", L(1), Rr(1)], "project": [0, 1, 4],
     "expected": [[11, "a", "z"], [22, "b", "y"], [44, "d", "x"]]},
    {"name": "equijoin_right_and_condition_from_left", "ref": SLT + "195-202", "join_type": "Right", "left": TB1, "right": TB2, "on": [[0, 0]], "filter": [">=", L(0), 22], "project": [0, 1, 4],
     "expected": [[22, "b", "y"], [44, "d", "x"], [N, N, "w"], [N, N, "z"]]},
    {"name": "equijoin_left_and_condition_from_left", "ref": SLT + "204-211", "join_type": "Left", "left": TB1, "right": TB2, "on": [[0, 0]], "filter": [">=", L(0), 44], "project": [0, 1, 4],
     "expected": [[11, "a", N], [22, "b", N], [33, "c", N], [44, "d", "x"]]},
    {"name": "equijoin_left_and_condition_from_both", "ref": SLT + "213-220", "join_type": "Left", "left": TB1, "right": TB2, "on": [[0, 0]], "filter": [">=", L(2), Rr(2)], "project": [0, 2, 5],
     "expected": [[11, 1, N], [22, 2, 1], [33, 3, N], [44, 4, 3]]},
    {"name": "equijoin_right_and_condition_from_right", "ref": SLT + "222-229", "join_type": "Right", "left": TB1, "right": TB2, "on": [[0, 0]], "filter": [">=", Rr(0), 22], "project": [0, 1, 4],
     "expected": [[22, "b", "y"], [44, "d", "x"], [N, N, "w"], [N, N, "z"]]},
    {"name": "equijoin_right_and_condition_from_both", "ref": SLT + "231-238", "join_type": "Right", "left": TB1, "right": TB2, "on": [[0, 0]], "filter": ["<=", Rr(2), L(2)], "project": [2, 5, 3],
     "expected": [[2, 1, 22], [4, 3, 44], [N, 3, 11], [N, 3, 55]]},
    {"name": "equijoin_full_and_condition_from_both", "ref": SLT + "250-259", "join_type": "Full", "left": TB1, "right": TB2, "on": [[0, 0]], "filter": ["<=", Rr(2), L(2)],
     "expected": [[11, "a", 1, N, N, N], [22, "b", 2, 22, "y", 1], [33, "c", 3, N, N, N], [44, "d", 4, 44, "x", 3], [N, N, N, 11, "z", 3], [N, N, N, 55, "w", 3]]},
]
sort = [
    {"name": "test_in_mem_sort", "ref": S + "1022-1049 (test::scan_partitioned(4): 4 partitions of make_partition(100), column i = 0..100)", "type": "int32",
     "partitions": [list(range(100))] * 4, "descending": False, "nulls_first": True, "expected_rows": 400, "expected_batches": 1},
    {"name": "test_sort_metadata", "ref": S + "1152-1197", "type": "uint64", "partitions": [[3, 2, 1]], "descending": False, "nulls_first": True, "expected": [1, 2, 3]},
]
R = "datafusion/physical-plan/src/repartition/mod.rs:"
repartition = [
    {"name": "one_to_many_round_robin", "ref": R + "952-969", "inputs": [50], "scheme": "RoundRobinBatch", "n": 4, "expected_batches": [13, 13, 12, 12]},
    {"name": "many_to_one_round_robin", "ref": R + "972-986", "inputs": [50, 50, 50], "scheme": "RoundRobinBatch", "n": 1, "expected_batches": [150]},
    {"name": "many_to_many_round_robin", "ref": R + "989-1007", "inputs": [50, 50, 50], "scheme": "RoundRobinBatch", "n": 5, "expected_batches": [30, 30, 30, 30, 30]},
    {"name": "many_to_many_hash_partition", "ref": R + "1010-1033", "inputs": [50, 50, 50], "scheme": "Hash", "n": 8, "expected_total_rows": 8 * 50 * 3},
]
json.dump({"binary": cases, "in_list": in_list, "nested_loop_join": nlj, "sort_merge_join": smj, "sort_merge_join_full": smj_full, "sort_merge_join_filter": smj_filter, "sort": sort, "repartition": {"batch": {"ref": R + "1440-1447 create_batch", "type": "uint32", "column": "c0", "values": [1, 2, 3, 4, 5, 6, 7, 8]}, "cases": repartition}},
          open(__file__.rsplit("/", 1)[0] + "/unit_vectors.json", "w"), indent=1)
print(len(cases), "binary cases,", len(in_list), "in_list cases")
