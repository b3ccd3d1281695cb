"""Writes hash_join.json: known-answer tables transcribed BY HAND from the reference's own unit tests
(datafusion/physical-plan/src/joins/hash_join.rs, `mod tests`).  Only data (inputs / expected rows / expected
batch counts) is recorded, each case citing the test it comes from.  Run: python transcribe_hash_join.py"""
import json

T3 = lambda a, b, c: {"names": [a[0], b[0], c[0]], "batches": [[a[1], b[1], c[1]]]}
N = None
BS = [8192, 10, 5, 2, 1]
cases = []

def case(name, ref, left, right, on, jt, expected, ordered, nen=False, filt=None, batch_sizes=BS, batch_count=None, dtype="int32", per_partition_expected=None):
    cases.append({"name": name, "ref": "datafusion/physical-plan/src/joins/hash_join.rs:" + ref, "left": left, "right": right, "on": on,
                  "join_type": jt, "null_equals_null": nen, "filter": filt, "expected": expected, "ordered": ordered,
                  "batch_sizes": batch_sizes, "batch_count": batch_count, "dtype": dtype, "per_partition_expected": per_partition_expected})

dc = lambda n, bs: (n + bs - 1) // bs
L1 = T3(("a1", [1, 2, 3]), ("b1", [4, 5, 5]), ("c1", [7, 8, 9]))
R1 = T3(("a2", [10, 20, 30]), ("b1", [4, 5, 6]), ("c2", [70, 80, 90]))
case("join_inner_one", "1553-1597", L1, R1, [["b1", "b1"]], "Inner",
     [[1, 4, 7, 10, 4, 70], [2, 5, 8, 20, 5, 80], [3, 5, 9, 20, 5, 80]], True)
case("join_inner_one_randomly_ordered", "1684-1721",
     T3(("a1", [0, 3, 2, 1]), ("b1", [4, 5, 5, 4]), ("c1", [6, 9, 8, 7])),
     T3(("a2", [20, 30, 10]), ("b2", [5, 6, 4]), ("c2", [80, 90, 70])), [["b1", "b2"]], "Inner",
     [[3, 5, 9, 20, 5, 80], [2, 5, 8, 20, 5, 80], [0, 4, 6, 10, 4, 70], [1, 4, 7, 10, 4, 70]], True, batch_sizes=[8192])
L2 = T3(("a1", [1, 2, 2]), ("b2", [1, 2, 2]), ("c1", [7, 8, 9]))
R2 = T3(("a1", [1, 2, 3]), ("b2", [1, 2, 2]), ("c2", [70, 80, 90]))
case("join_inner_two", "1725-1775", L2, R2, [["a1", "a1"], ["b2", "b2"]], "Inner",
     [[1, 1, 7, 1, 1, 70], [2, 2, 8, 2, 2, 80], [2, 2, 9, 2, 2, 80]], True,
     batch_count={str(bs): dc(3, bs) + (1 if bs == 1 else 0) for bs in BS})
case("join_inner_one_two_parts_left", "1780-1837",
     {"names": ["a1", "b2", "c1"], "batches": [[[1, 2], [1, 2], [7, 8]], [[2], [2], [9]]]}, R2, [["a1", "a1"], ["b2", "b2"]], "Inner",
     [[1, 1, 7, 1, 1, 70], [2, 2, 8, 2, 2, 80], [2, 2, 9, 2, 2, 80]], True,
     batch_count={str(bs): dc(3, bs) + (1 if bs == 1 else 0) for bs in BS})
case("join_inner_one_two_parts_right", "1892-1966", L1,
     {"names": ["a2", "b1", "c2"], "batches": [[[10, 20], [4, 6], [70, 80]], [[30], [5], [90]]]}, [["b1", "b1"]], "Inner",
     [[1, 4, 7, 10, 4, 70], [2, 5, 8, 30, 5, 90], [3, 5, 9, 30, 5, 90]], True,
     per_partition_expected=[[[1, 4, 7, 10, 4, 70]], [[2, 5, 8, 30, 5, 90], [3, 5, 9, 30, 5, 90]]])
L3 = T3(("a1", [1, 2, 3]), ("b1", [4, 5, 7]), ("c1", [7, 8, 9]))
R3two = {"names": ["a2", "b1", "c2"], "batches": [[[10, 20, 30], [4, 5, 6], [70, 80, 90]]] * 2}
case("join_left_multi_batch", "1982-2020", L3, R3two, [["b1", "b1"]], "Left",
     [[1, 4, 7, 10, 4, 70], [1, 4, 7, 10, 4, 70], [2, 5, 8, 20, 5, 80], [2, 5, 8, 20, 5, 80], [3, 7, 9, N, N, N]], False)
R3two_b2 = {"names": ["a2", "b2", "c2"], "batches": [[[10, 20, 30], [4, 5, 6], [70, 80, 90]]] * 2}
case("join_full_multi_batch", "2024-2065", L3, R3two_b2, [["b1", "b2"]], "Full",
     [[N, N, N, 30, 6, 90], [N, N, N, 30, 6, 90], [1, 4, 7, 10, 4, 70], [1, 4, 7, 10, 4, 70], [2, 5, 8, 20, 5, 80], [2, 5, 8, 20, 5, 80], [3, 7, 9, N, N, N]], False)
EMPTY_R = {"names": ["a2", "b1", "c2"], "batches": [[[], [], []]]}
case("join_left_empty_right", "2069-2102", L3, EMPTY_R, [["b1", "b1"]], "Left",
     [[1, 4, 7, N, N, N], [2, 5, 8, N, N, N], [3, 7, 9, N, N, N]], False)
case("join_full_empty_right", "2106-2139", L3, {"names": ["a2", "b2", "c2"], "batches": [[[], [], []]]}, [["b1", "b2"]], "Full",
     [[1, 4, 7, N, N, N], [2, 5, 8, N, N, N], [3, 7, 9, N, N, N]], False)
case("join_left_one", "2143-2183", L3, R1, [["b1", "b1"]], "Left",
     [[1, 4, 7, 10, 4, 70], [2, 5, 8, 20, 5, 80], [3, 7, 9, N, N, N]], False)
SL = T3(("a1", [1, 3, 5, 7, 9, 11, 13]), ("b1", [1, 3, 5, 7, 8, 8, 10]), ("c1", [10, 30, 50, 70, 90, 110, 130]))
SR = T3(("a2", [8, 12, 6, 2, 10, 4]), ("b2", [8, 10, 6, 2, 10, 4]), ("c2", [20, 40, 60, 80, 100, 120]))
ON_S = [["b1", "b2"]]
F = lambda side, idx, op, lit: {"column_indices": [[side, idx]], "op": op, "rhs_literal": lit}
case("join_left_semi", "2251-2282", SL, SR, ON_S, "LeftSemi", [[11, 8, 110], [13, 10, 130], [9, 8, 90]], False)
case("join_left_semi_with_filter_ne10", "2286-2341", SL, SR, ON_S, "LeftSemi", [[11, 8, 110], [13, 10, 130], [9, 8, 90]], False, filt=F("right", 0, "!=", 10))
case("join_left_semi_with_filter_gt10", "2343-2369", SL, SR, ON_S, "LeftSemi", [[13, 10, 130]], False, filt=F("right", 0, ">", 10))
case("join_right_semi", "2373-2406", SL, SR, ON_S, "RightSemi", [[8, 8, 20], [12, 10, 40], [10, 10, 100]], True)
case("join_right_semi_with_filter_ne9", "2410-2469", SL, SR, ON_S, "RightSemi", [[8, 8, 20], [12, 10, 40], [10, 10, 100]], True, filt=F("left", 0, "!=", 9))
case("join_right_semi_with_filter_gt11", "2471-2496", SL, SR, ON_S, "RightSemi", [[12, 10, 40], [10, 10, 100]], True, filt=F("left", 0, ">", 11))
case("join_left_anti", "2500-2531", SL, SR, ON_S, "LeftAnti", [[1, 1, 10], [3, 3, 30], [5, 5, 50], [7, 7, 70]], False)
case("join_left_anti_with_filter_ne8", "2534-2592", SL, SR, ON_S, "LeftAnti",
     [[1, 1, 10], [11, 8, 110], [3, 3, 30], [5, 5, 50], [7, 7, 70], [9, 8, 90]], False, filt=F("right", 0, "!=", 8))
case("join_right_anti", "2628-2659", SL, SR, ON_S, "RightAnti", [[6, 6, 60], [2, 2, 80], [4, 4, 120]], True)
case("join_right_anti_with_filter_left_a1_ne13", "2662-2722", SL, SR, ON_S, "RightAnti",
     [[12, 10, 40], [6, 6, 60], [2, 2, 80], [10, 10, 100], [4, 4, 120]], True, filt=F("left", 0, "!=", 13))
case("join_right_anti_with_filter_right_b2_ne8", "2724-2759", SL, SR, ON_S, "RightAnti",
     [[8, 8, 20], [6, 6, 60], [2, 2, 80], [4, 4, 120]], True, filt=F("right", 1, "!=", 8))
case("join_right_one", "2763-2798", L3, R1, [["b1", "b1"]], "Right",
     [[N, N, N, 30, 6, 90], [1, 4, 7, 10, 4, 70], [2, 5, 8, 20, 5, 80]], False)
case("join_full_one", "2842-2880", L3, T3(("a2", [10, 20, 30]), ("b2", [4, 5, 6]), ("c2", [70, 80, 90])), [["b1", "b2"]], "Full",
     [[N, N, N, 30, 6, 90], [1, 4, 7, 10, 4, 70], [2, 5, 8, 20, 5, 80], [3, 7, 9, N, N, N]], False)
FL = T3(("a", [0, 1, 2, 2]), ("b", [4, 5, 7, 8]), ("c", [7, 8, 9, 1]))
FR = T3(("a", [10, 20, 30, 40]), ("b", [2, 2, 3, 4]), ("c", [7, 5, 6, 4]))
FJ = {"column_indices": [["left", 2], ["right", 2]], "op": ">", "rhs_column": 1}     # prepare_join_filter :2985-3007: left.c > right.c
case("join_inner_with_filter", "3011-3048", FL, FR, [["a", "b"]], "Inner", [[2, 7, 9, 10, 2, 7], [2, 7, 9, 20, 2, 5]], False, filt=FJ)
case("join_left_with_filter", "3052-3092", FL, FR, [["a", "b"]], "Left",
     [[0, 4, 7, N, N, N], [1, 5, 8, N, N, N], [2, 7, 9, 10, 2, 7], [2, 7, 9, 20, 2, 5], [2, 8, 1, N, N, N]], False, filt=FJ)
case("join_right_with_filter", "3096-3135", FL, FR, [["a", "b"]], "Right",
     [[N, N, N, 30, 3, 6], [N, N, N, 40, 4, 4], [2, 7, 9, 10, 2, 7], [2, 7, 9, 20, 2, 5]], False, filt=FJ)
case("join_full_with_filter", "3139-3180", FL, FR, [["a", "b"]], "Full",
     [[N, N, N, 30, 3, 6], [N, N, N, 40, 4, 4], [2, 7, 9, 10, 2, 7], [2, 7, 9, 20, 2, 5], [0, 4, 7, N, N, N], [1, 5, 8, N, N, N], [2, 8, 1, N, N, N]], False, filt=FJ)
case("join_date32", "3184-3224", {"names": ["date", "n"], "batches": [[[19107, 19108, 19109], [1, 2, 3]]]},
     {"names": ["date", "n"], "batches": [[[19108, 19108, 19109], [4, 5, 6]]]}, [["date", "date"]], "Inner",
     [[19108, 2, 19108, 4], [19108, 2, 19108, 5], [19109, 3, 19109, 6]], False, batch_sizes=[8192], dtype="date32,int32")
# join_splitted_batch :3283-3415 -- every join type x batch_size 20..1, ORDER asserted, batch counts asserted
SPL = T3(("a1", [1, 2, 3, 4]), ("b1", [1, 1, 1, 1]), ("c1", [0, 0, 0, 0]))
SPR = T3(("a2", [10, 20, 30, 40, 50]), ("b2", [1, 1, 1, 1, 1]), ("c2", [0, 0, 0, 0, 0]))
common = [[a, 1, 0, r, 1, 0] for r in [10, 20, 30, 40, 50] for a in [1, 2, 3, 4]]
lb = [[a, 1, 0] for a in [1, 2, 3, 4]]
rb = [[r, 1, 0] for r in [10, 20, 30, 40, 50]]
for jt, exp in [("Inner", common), ("Left", common), ("Right", common), ("Full", common), ("RightSemi", rb), ("RightAnti", []), ("LeftSemi", lb), ("LeftAnti", [])]:
    extra = 0 if jt in ("Inner", "Right", "RightSemi", "RightAnti") else 1
    case("join_splitted_batch_" + jt, "3283-3415", SPL, SPR, [["b1", "b2"]], jt, exp, True, batch_sizes=list(range(20, 0, -1)),
         batch_count={str(bs): dc(20, bs) + extra for bs in range(20, 0, -1)})

json.dump({"source": "transcribed from datafusion/physical-plan/src/joins/hash_join.rs (mod tests); data only", "cases": cases},
          open("hash_join.json", "w"), indent=1)
print(len(cases), "cases")
