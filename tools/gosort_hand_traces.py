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

    python tools/gosort_hand_traces.py [--write]  # prints the vectors as JSON, writes the trace files (--write: also into reference_tests.json)
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


# ---- [r5] 13 .. 40 elements: quickSort's loop runs, doPivot takes the median-of-three branch (`hi-lo > 40` — Tukey's
# ninther — is not taken), the pieces of <= 12 elements end in the two straight-line steps above.  Written out the same
# way: every Less and every Swap of (go1.14 src/sort/sort.go)
#
#   func medianOfThree(data Interface, m1, m0, m2 int) {
#       if data.Less(m1, m0) { data.Swap(m1, m0) }
#       if data.Less(m2, m1) { data.Swap(m2, m1); if data.Less(m1, m0) { data.Swap(m1, m0) } }
#   }
#   func doPivot(data Interface, lo, hi int) (midlo, midhi int) {
#       m := int(uint(lo+hi) >> 1)
#       if hi-lo > 40 { ... ninther ... }
#       medianOfThree(data, lo, m, hi-1)
#       pivot := lo
#       a, c := lo+1, hi-1
#       for ; a < c && data.Less(a, pivot); a++ { }
#       b := a
#       for {
#           for ; b < c && !data.Less(pivot, b); b++ { }
#           for ; b < c && data.Less(pivot, c-1); c-- { }
#           if b >= c { break }
#           data.Swap(b, c-1); b++; c--
#       }
#       protect := hi-c < 5
#       if !protect && hi-c < (hi-lo)/4 {
#           dups := 0
#           if !data.Less(pivot, hi-1) { data.Swap(c, hi-1); c++; dups++ }
#           if !data.Less(b-1, pivot) { b--; dups++ }
#           if !data.Less(m, pivot) { data.Swap(m, b-1); b--; dups++ }
#           protect = dups > 1
#       }
#       if protect {
#           for {
#               for ; a < b && !data.Less(b-1, pivot); b-- { }
#               for ; a < b && data.Less(a, pivot); a++ { }
#               if a >= b { break }
#               data.Swap(a, b-1); a++; b--
#           }
#       }
#       data.Swap(pivot, b-1)
#       return b - 1, c
#   }
#   func quickSort(data Interface, a, b, maxDepth int) {
#       for b-a > 12 {
#           if maxDepth == 0 { heapSort(data, a, b); return }
#           maxDepth--
#           mlo, mhi := doPivot(data, a, b)
#           if mlo-a < b-mhi { quickSort(data, a, mlo, maxDepth); a = mhi } else { quickSort(data, mhi, b, maxDepth); b = mlo }
#       }
#       if b-a > 1 { shell pass with gap 6; insertionSort(data, a, b) }
#   }
#   Sort: quickSort(data, 0, n, maxDepth(n)),  maxDepth(n) = 2 * (number of bits of n)
#
# The inputs are chosen so that maxDepth never reaches 0 (heapSort stays pinned by the restatements agreeing only).
MID_INPUTS = [
    [5] * 13, [3, 7] * 7, [9, 1, 4] * 5, list(range(13, 0, -1)), [2] * 7 + [8] * 7, [8] * 7 + [2] * 7,
    [1, 1, 5, 5, 9, 9] * 3, [6, 2, 6, 2, 6, 2, 6, 2, 6, 2, 6, 2, 6, 2, 6, 2, 6], [4] * 20, [7, 7, 7, 1] * 5,
    [3, 1, 4, 1, 5, 9, 2, 6, 5, 3, 5, 8, 9, 7, 9, 3, 2, 3, 8, 4, 6], [10, 20] * 12, [20, 10] * 12 + [10],
    [1] * 12 + [2] * 12 + [1], [5, 4, 3, 2, 1] * 6, [1, 2, 3, 4, 5] * 6, [9] * 15 + [0] * 15 + [9],
    [2, 2, 2, 7, 7, 7, 4, 4, 4, 4] * 3 + [4, 7], [6] * 33, [0, 1] * 17, [3, 3, 1, 1, 2, 2, 2] * 5, list(range(37)),
    [8, 1, 8, 1, 8, 8, 1, 1] * 5, [40 - i // 3 for i in range(40)], [i % 4 for i in range(40)], [11] * 39 + [3],
]


def trace_mid(keys, out):
    n = len(keys)
    assert 13 <= n <= 40
    d = ["%d:%d" % (k, i) for i, k in enumerate(keys)]
    key = lambda s: int(s.split(":")[0])

    def less(i, j, pad):
        r = key(d[i]) < key(d[j])
        out.append("%sLess(%d,%d): %s < %s  %s" % (pad, i, j, d[i], d[j], "yes" if r else "no"))
        return r

    def swap(i, j, pad):
        out.append("%sSwap(%d,%d): %s <-> %s" % (pad, i, j, d[i], d[j]))
        d[i], d[j] = d[j], d[i]

    def median_of_three(m1, m0, m2, pad):
        out.append("%smedianOfThree(m1=%d, m0=%d, m2=%d)" % (pad, m1, m0, m2))
        if less(m1, m0, pad + "  "):
            swap(m1, m0, pad + "  ")
        if less(m2, m1, pad + "  "):
            swap(m2, m1, pad + "  ")
            if less(m1, m0, pad + "  "):
                swap(m1, m0, pad + "  ")

    def do_pivot(lo, hi, pad):
        out.append("%sdoPivot(lo=%d, hi=%d)" % (pad, lo, hi))
        p = pad + "  "
        m = (lo + hi) >> 1
        assert hi - lo <= 40
        median_of_three(lo, m, hi - 1, p)
        pivot = lo
        a, c = lo + 1, hi - 1
        while a < c and less(a, pivot, p):
            a += 1
        b = a
        while True:
            while b < c and not less(pivot, b, p):
                b += 1
            while b < c and less(pivot, c - 1, p):
                c -= 1
            if b >= c:
                break
            swap(b, c - 1, p)
            b += 1
            c -= 1
        protect = hi - c < 5
        out.append("%sa=%d b=%d c=%d  hi-c=%d  protect=%s" % (p, a, b, c, hi - c, protect))
        if not protect and hi - c < (hi - lo) // 4:
            dups = 0
            if not less(pivot, hi - 1, p):
                swap(c, hi - 1, p)
                c += 1
                dups += 1
            if not less(b - 1, pivot, p):
                b -= 1
                dups += 1
            if not less(m, pivot, p):
                swap(m, b - 1, p)
                b -= 1
                dups += 1
            protect = dups > 1
            out.append("%sdups=%d  protect=%s  b=%d c=%d" % (p, dups, protect, b, c))
        if protect:
            while True:
                while a < b and not less(b - 1, pivot, p):
                    b -= 1
                while a < b and less(a, pivot, p):
                    a += 1
                if a >= b:
                    break
                swap(a, b - 1, p)
                a += 1
                b -= 1
        swap(pivot, b - 1, p)
        out.append("%s-> midlo=%d midhi=%d   %s" % (p, b - 1, c, " ".join(d[lo:hi])))
        return b - 1, c

    def quick_sort(a, b, max_depth, pad):
        out.append("%squickSort(a=%d, b=%d, maxDepth=%d)" % (pad, a, b, max_depth))
        while b - a > 12:
            assert max_depth != 0, "heapSort is not traced: pick another input"
            max_depth -= 1
            mlo, mhi = do_pivot(a, b, pad + "  ")
            if mlo - a < b - mhi:
                quick_sort(a, mlo, max_depth, pad + "  ")
                a = mhi
            else:
                quick_sort(mhi, b, max_depth, pad + "  ")
                b = mlo
            out.append("%s  continue with a=%d b=%d maxDepth=%d" % (pad, a, b, max_depth))
        if b - a > 1:
            p = pad + "  "
            out.append("%sshell pass, gap 6, [%d,%d)" % (p, a, b))
            for i in range(a + 6, b):
                if less(i, i - 6, p + "  "):
                    swap(i, i - 6, p + "  ")
            out.append("%sinsertion sort [%d,%d)" % (p, a, b))
            for i in range(a + 1, b):
                j = i
                while j > a and less(j, j - 1, p + "  "):
                    swap(j, j - 1, p + "  ")
                    j -= 1

    out.append("input   " + " ".join(d))
    depth, i = 0, n
    while i > 0:
        depth += 1
        i >>= 1
    quick_sort(0, n, 2 * depth, "  ")
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
    mid_lines, mid_vectors = [], []
    for keys in MID_INPUTS:
        perm = trace_mid(keys, mid_lines)
        assert sorted(perm) == list(range(len(keys))) and all(keys[perm[i]] <= keys[perm[i + 1]] for i in range(len(keys) - 1))
        mid_vectors.append({"keys": keys, "perm": perm})
    with open(os.path.join(ROOT, "tests", "golden", "go_sort_mid_traces.txt"), "w") as f:
        f.write("Go 1.14 sort.Sort, 13 <= n <= 40: quickSort -> doPivot (median of three; no ninther) -> pieces of <= 12 elements by the\n"
                "ShellSort pass with gap 6 and insertionSort (src/sort/sort.go).  Elements are key:original-position; every Less and Swap\n"
                "in the order sort.Sort makes them; written by tools/gosort_hand_traces.py.\n\n" + "\n".join(mid_lines))
    if "--write" in sys.argv:                                   # merge both sets into tests/golden/reference_tests.json
        path = os.path.join(ROOT, "tests", "golden", "reference_tests.json")
        ref = json.load(open(path))
        ref["go_sort_small"]["vectors"] = vectors
        ref["go_sort_mid"] = {"_comment": "Go 1.14 sort.Sort on 13 .. 40 tie-heavy keys: quickSort / doPivot with the median of three, pieces of <= 12 "
                              "elements by the ShellSort gap-6 pass + insertionSort, traced Less by Less in go_sort_mid_traces.txt "
                              "(tools/gosort_hand_traces.py); perm[i] = original position of the element that ends at i", "vectors": mid_vectors}
        json.dump(ref, open(path, "w"), indent=1, ensure_ascii=False)
    json.dump({"go_sort_small": vectors, "go_sort_mid": mid_vectors}, sys.stdout)
    print()
    return vectors


if __name__ == "__main__":
    main()
