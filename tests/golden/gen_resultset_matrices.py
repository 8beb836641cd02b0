#!/usr/bin/env python3
"""Writes tests/golden/resultset_matrices.json: the case matrices of the reference's layout / reduction unit
tests, as DATA (descriptor parameters, target lists, generators, expected-value rule) -- no reference code.

Source of every row: omniscidb/Tests/ResultSetTest.cpp (line ranges in "ref") with the descriptor builders
and target lists of omniscidb/Tests/ResultSetTestUtils.cpp:484-601 and ResultSetTest.cpp:933-992.
Run from the repo root:  python tests/golden/gen_resultset_matrices.py
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))

# target lists: (is_agg, agg, type, arg_type); every TargetInfo has skip_null_val = true, is_distinct = false
TARGET_SETS = {
    # generate_test_target_infos(), ResultSetTest.cpp:933-952
    "test": [[False, "min", "int32", None], [True, "avg", "int32", "int32"], [True, "sum", "int32", "int32"],
             [False, "min", "fp64", None], [False, "min", "dict32", None]],
    # generate_random_groups_target_infos(), :954-970 (NOT NULL types)
    "random_groups": [[True, "min", "int32nn", "int32nn"], [True, "max", "int32nn", "int32nn"],
                      [True, "sum", "int32nn", "int32nn"], [True, "count", "int32nn", "int32nn"],
                      [True, "avg", "int32nn", "fp64nn"]],
    # generate_random_groups_nullable_target_infos(), :972-990
    "random_groups_nullable": [[True, "min", "int32", "int32"], [True, "max", "int32", "int32"],
                               [True, "sum", "int32", "int32"], [True, "count", "int32", "int32"],
                               [True, "avg", "int32", "fp64"]],
    # generate_custom_agg_target_infos({8}, {max,min,count,sum,avg}, ...), :1692-1730 (2- and 1-byte slots)
    "custom16": [[False, "min", "int64", None], [True, "max", "int16", "int16"], [True, "min", "int16", "int16"],
                 [True, "count", "int32", "int16"], [True, "sum", "int64", "int16"], [True, "avg", "fp64", "int16"]],
    "custom8": [[False, "min", "int64", None], [True, "max", "int8", "int8"], [True, "min", "int8", "int8"],
                [True, "count", "int32", "int8"], [True, "sum", "int64", "int8"], [True, "avg", "fp64", "int8"]],
}

# descriptor builders, ResultSetTestUtils.cpp:484-601: (hash kind, min, max, group col widths, entry count rule)
DESCRIPTORS = {
    "perfect_hash_one_col_desc_0_99": {"kind": "perfect", "min": 0, "max": 99, "group_col_widths": [8], "entry_count": 100},
    "perfect_hash_one_col_desc_small": {"kind": "perfect", "min": 0, "max": 19, "group_col_widths": [8], "entry_count": 20},
    "perfect_hash_two_col_desc": {"kind": "perfect", "min": 0, "max": 36, "group_col_widths": [8, 8], "entry_count": 36},
    "baseline_hash_two_col_desc": {"kind": "baseline", "min": 0, "max": 3, "group_col_widths": [8, 8], "entry_count": 4},
    "baseline_hash_two_col_desc_large": {"kind": "baseline", "min": 0, "max": 19, "group_col_widths": [8, 8],
                                         "entry_count": 20},
}


def reduce_cases():
    out = []

    def add(name, ref, desc, num_bytes, columnar=False, keyless=False, targets="test", gen2="even", step=2, sort=False):
        out.append({"name": name, "ref": ref, "desc": desc, "num_bytes": num_bytes, "columnar": columnar,
                    "keyless": keyless, "target_idx_for_key": 2 if keyless else None, "targets": targets,
                    "gen1": "even", "gen2": gen2, "step": step, "sort": sort})

    one, two, base = "perfect_hash_one_col_desc_0_99", "perfect_hash_two_col_desc", "baseline_hash_two_col_desc"
    add("Reduce.PerfectHashOneCol", "1658-1664", one, 8)
    add("Reduce.PerfectHashOneCol32", "1666-1672", one, 4)
    add("Reduce.PerfectHashOneColColumnar", "1674-1681", one, 8, columnar=True)
    add("Reduce.PerfectHashOneColColumnar32", "1683-1690", one, 4, columnar=True)
    add("Reduce.PerfectHashOneColColumnar16", "1692-1710", one, 2, columnar=True, targets="custom16")
    add("Reduce.PerfectHashOneColColumnar8", "1712-1730", one, 1, columnar=True, targets="custom8")
    add("Reduce.PerfectHashOneColKeyless", "1732-1740", one, 8, keyless=True)
    add("Reduce.PerfectHashOneColKeyless32", "1742-1750", one, 4, keyless=True)
    add("Reduce.PerfectHashOneColColumnarKeyless", "1752-1761", one, 8, columnar=True, keyless=True)
    add("Reduce.PerfectHashOneColColumnarKeyless32", "1763-1772", one, 4, columnar=True, keyless=True)
    add("Reduce.PerfectHashTwoCol", "1818-1824", two, 8)
    add("Reduce.PerfectHashTwoCol32", "1826-1832", two, 4)
    add("Reduce.PerfectHashTwoColColumnar", "1834-1841", two, 8, columnar=True)
    add("Reduce.PerfectHashTwoColColumnar32", "1843-1850", two, 4, columnar=True)
    add("Reduce.PerfectHashTwoColKeyless", "1852-1860", two, 8, keyless=True)
    add("Reduce.PerfectHashTwoColKeyless32", "1862-1870", two, 4, keyless=True)
    add("Reduce.PerfectHashTwoColColumnarKeyless", "1872-1881", two, 8, columnar=True, keyless=True)
    add("Reduce.PerfectHashTwoColColumnarKeyless32", "1883-1892", two, 4, columnar=True, keyless=True)
    # generator2 = ReverseOddOrEvenNumberGenerator(2 * entry_count - 1)
    add("Reduce.BaselineHash", "1894-1900", base, 8, gen2="reverse_odd", step=1, sort=True)
    add("Reduce.BaselineHashColumnar", "1902-1909", base, 8, columnar=True, gen2="reverse_odd", step=1, sort=True)
    return out


def random_group_cases():
    out = []
    small, large = "perfect_hash_one_col_desc_small", "baseline_hash_two_col_desc_large"

    def fam(prefix, lines, desc, columnar, targets, flow, pcts):
        assert len(lines) == len(pcts)
        for line, (p1, p2, tag) in zip(lines, pcts):
            out.append({"name": f"ReduceRandomGroups.{prefix}_{tag}", "ref": str(line), "desc": desc,
                        "num_bytes": 8, "columnar": columnar, "targets": targets, "prct1": p1, "prct2": p2, "flow": flow})

    P_SMALL = [(25, 25, "2525"), (25, 75, "2575"), (50, 50, "5050"), (75, 25, "7525"), (25, 100, "25100"),
               (100, 25, "10025"), (95, 5, "9505"), (100, 100, "100100"), (25, 0, "2500"), (0, 75, "0075")]
    P_BASE = [(50, 50, "5050"), (75, 25, "7525"), (25, 75, "2575"), (10, 20, "1020"), (100, 100, "100100"),
              (25, 0, "2500"), (0, 75, "0075")]
    P_COL = [(50, 50, "5050"), (25, 100, "25100"), (100, 25, "10025"), (100, 100, "100100"), (25, 0, "2500"),
             (0, 75, "0075")]
    P_NULL = [p for p in P_SMALL if p[2] != "9505"]
    # first line of each TEST(ReduceRandomGroups, ...) in ResultSetTest.cpp
    fam("PerfectHashOneCol_Small", [2239, 2252, 2265, 2278, 2291, 2304, 2317, 2329, 2341, 2353], small, False,
        "random_groups", 0, P_SMALL)
    fam("BaselineHash_Large", [2366, 2378, 2390, 2402, 2414, 2426, 2438], large, False, "random_groups", 0, P_BASE)
    fam("PerfectHashOneColColumnar_Small", [2451, 2464, 2477, 2490, 2503, 2516], small, True, "random_groups", 0, P_COL)
    fam("BaselineHashColumnar_Large", [2530, 2543, 2556, 2569, 2582, 2595], large, True, "random_groups", 0, P_COL)
    fam("PerfectHashOneCol_NullVal", [2609, 2623, 2637, 2651, 2665, 2679, 2693, 2707, 2721], small, False,
        "random_groups_nullable", 2, P_NULL)
    fam("PerfectHashOneColColumnar_NullVal", [2736, 2749, 2762, 2775, 2788, 2801], small, True,
        "random_groups_nullable", 2, P_COL)
    fam("BaselineHash_Large_NullVal", [2815, 2827, 2839, 2851, 2863, 2875, 2887], large, False,
        "random_groups_nullable", 2, P_BASE)
    fam("BaselineHashColumnar_Large_NullVal", [2900, 2913, 2926, 2939, 2952, 2965], large, True,
        "random_groups_nullable", 2, P_COL)
    return out


def main():
    doc = {
        "source": "omniscidb/Tests/ResultSetTest.cpp + omniscidb/Tests/ResultSetTestUtils.cpp (case matrices as data)",
        "notes": [
            "fill procedures, generators and expected-value rules are restated in tests/rs_matrix.py with file:line cites",
            "group membership of the ReduceRandomGroups cases is random in the reference (std::random_device); the port "
            "draws it from a seeded generator with the same percentages",
            "rows are placed at QueryMemoryDescriptor::getRowSize() stride with slots at getColOffInBytes() "
            "(the reference's fill_one_entry_no_collisions advances by the logical slot widths)",
        ],
        "target_sets": TARGET_SETS,
        "descriptors": DESCRIPTORS,
        "reduce_cases": reduce_cases(),
        "random_group_cases": random_group_cases(),
    }
    with open(os.path.join(HERE, "resultset_matrices.json"), "w") as f:
        json.dump(doc, f, indent=1)
    print(len(doc["reduce_cases"]), "reduce cases,", len(doc["random_group_cases"]), "random-group cases")


if __name__ == "__main__":
    main()
