#!/usr/bin/env python3
"""Hand traces of Go 1.14's sort.Sort on small, tie-heavy inputs — vectors for tests/golden/reference_tests.json["go_sort_small"].

cpMerge / intersector sort the posting lists by length with sort.Sort (pkg/merger/cp_merge.go:24, list_intersector.go:30);
the sort is unstable, and WHICH order it leaves equal lengths in shows in the secondary rows of documents that repeat a
term.  No Go toolchain exists in the build image, so the three restatements of sort.Sort in this repository
(oracle C++, tests/gosort.py, the device's PairSort) are pinned against each other only.  This script is deliberately NOT a
fourth restatement of the whole algorithm: for n <= 12 elements sort.Sort runs exactly two straight-line steps
(go1.14 src/sort/sort.go, quickSort: `for b-a > 12 { ... }` is skipped, then

        if b-a > 1 {
            // Do ShellSort pass with gap 6
            // It could be written in this simplified form cause b-a <= 12
            for i := a + 6; i < b; i++ {
                if data.Less(i, i-6) {
                    data.Swap(i, i-6)
                }
            }
            insertionSort(data, a, b)
        }

 with insertionSort = `for i := a + 1; i < b; i++ { for j := i; j > a && data.Less(j, j-1); j-- { data.Swap(j, j-1) } }`),
and it writes every compare-and-swap of those two steps out as a line a reader can follow with pencil and paper
(tests/golden/go_sort_small_traces.txt).  Elements are written key:tag, tag = original position; Less compares keys only.

    python tools/gosort_hand_traces.py            # prints the vectors as JSON, writes the trace file
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

INPUTS = [   # tie-heavy list lengths, n <= 12 (equal keys are where an unstable sort shows)
    [5, 5], [7, 3, 7], [2, 2, 2], [4, 1, 4, 1], [3, 3, 1, 3, 1],
    [9, 9, 9, 9, 9, 9, 1], [1, 9, 9, 9, 9, 9, 9], [6, 5, 4, 3, 2, 1, 0], [0, 1, 2, 3, 4, 5, 6, 7],
    [5, 5, 5, 5, 5, 5, 5, 5], [8, 8, 8, 8, 8, 8, 3, 3], [3, 3, 8, 8, 8, 8, 8, 8], [2, 7, 2, 7, 2, 7, 2, 7],
    [7, 2, 7, 2, 7, 2, 7, 2, 7], [4, 4, 4, 9, 9, 9, 1, 1, 1], [1, 1, 1, 9, 9, 9, 4, 4, 4],
    [6, 6, 6, 6, 6, 6, 6, 6, 6, 2], [2, 6, 6, 6, 6, 6, 6, 6, 6, 6], [5, 3, 5, 3, 5, 3, 5, 3, 5, 3],
    [10, 20, 10, 20, 10, 20, 10, 20, 10, 20, 10], [20, 10, 20, 10, 20, 10, 20, 10, 20, 10, 20],
    [3, 1, 4, 1, 5, 9, 2, 6, 5, 3, 5], [7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7], [12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1],
    [1, 1, 2, 2, 3, 3, 1, 1, 2, 2, 3, 3], [3, 3, 2, 2, 1, 1, 3, 3, 2, 2, 1, 1], [9, 1, 9, 1, 9, 1, 1, 9, 1, 9, 1, 9],
    [4, 4, 4, 4, 4, 4, 0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 4, 4, 4, 4, 4, 4], [5, 9, 5, 9, 5, 5, 9, 9, 5, 9, 9, 5],
]


def trace(keys, out):
    n = len(keys)
    assert 2 <= n <= 12
    d = ["%d:%d" % (k, i) for i, k in enumerate(keys)]
    key = lambda s: int(s.split(":")[0])
    out.append("input   " + " ".join(d))
    out.append("  shell pass, gap 6")
    for i in range(6, n):
        if key(d[i]) < key(d[i - 6]):
            out.append("    Less(%d,%d): %s < %s  -> swap" % (i, i - 6, d[i], d[i - 6]))
            d[i], d[i - 6] = d[i - 6], d[i]
        else:
            out.append("    Less(%d,%d): %s < %s  no" % (i, i - 6, d[i], d[i - 6]))
    out.append("          " + " ".join(d))
    out.append("  insertion sort")
    for i in range(1, n):
        j = i
        while j > 0:
            if key(d[j]) < key(d[j - 1]):
                out.append("    i=%d Less(%d,%d): %s < %s  -> swap" % (i, j, j - 1, d[j], d[j - 1]))
                d[j], d[j - 1] = d[j - 1], d[j]
                j -= 1
            else:
                out.append("    i=%d Less(%d,%d): %s < %s  no" % (i, j, j - 1, d[j], d[j - 1]))
                break
    out.append("result  " + " ".join(d))
    out.append("")
    return [int(s.split(":")[1]) for s in d]


def main():
    lines, vectors = [], []
    for keys in INPUTS:
        perm = trace(keys, lines)
        assert sorted(perm) == list(range(len(keys))) and all(keys[perm[i]] <= keys[perm[i + 1]] for i in range(len(keys) - 1))
        vectors.append({"keys": keys, "perm": perm})
    with open(os.path.join(ROOT, "tests", "golden", "go_sort_small_traces.txt"), "w") as f:
        f.write("Go 1.14 sort.Sort, n <= 12: ShellSort pass with gap 6, then insertionSort (src/sort/sort.go, quickSort).\n"
                "Elements are key:original-position; written by tools/gosort_hand_traces.py.\n\n" + "\n".join(lines))
    json.dump(vectors, sys.stdout)
    print()
    return vectors


if __name__ == "__main__":
    main()
