"""A third restatement of Go 1.14's sort.Sort (src/sort/sort.go of go1.14: insertionSort, siftDown, heapSort,
medianOfThree, doPivot, quickSort, maxDepth), written separately from the oracle's (oracle/suggest_oracle.cpp GoSort)
and the device's (engine.hip PairSort): recursive like the original, on a Python list of (key, tag) pairs.  The sort is
unstable; what matters for parity is WHICH unstable order it produces for equal keys (cp_merge.go:24 sorts posting
lists by length).  Test infrastructure."""


def go_sort(keys):
    """-> permutation p: p[i] = original index of the element that ends at position i"""
    data = [(k, i) for i, k in enumerate(keys)]

    def less(i, j):
        return data[i][0] < data[j][0]

    def swap(i, j):
        data[i], data[j] = data[j], data[i]

    def insertion_sort(a, b):
        for i in range(a + 1, b):
            j = i
            while j > a and less(j, j - 1):
                swap(j, j - 1)
                j -= 1

    def sift_down(lo, hi, first):
        root = lo
        while True:
            child = 2 * root + 1
            if child >= hi:
                return
            if child + 1 < hi and less(first + child, first + child + 1):
                child += 1
            if not less(first + root, first + child):
                return
            swap(first + root, first + child)
            root = child

    def heap_sort(a, b):
        first, lo, hi = a, 0, b - a
        for i in range((hi - 1) // 2, -1, -1):
            sift_down(i, hi, first)
        for i in range(hi - 1, -1, -1):
            swap(first, first + i)
            sift_down(lo, i, first)

    def median_of_three(m1, m0, m2):
        if less(m1, m0):
            swap(m1, m0)
        if less(m2, m1):
            swap(m2, m1)
            if less(m1, m0):
                swap(m1, m0)

    def do_pivot(lo, hi):
        m = (lo + hi) >> 1
        if hi - lo > 40:
            s = (hi - lo) // 8
            median_of_three(lo, lo + s, lo + 2 * s)
            median_of_three(m, m - s, m + s)
            median_of_three(hi - 1, hi - 1 - s, hi - 1 - 2 * s)
        median_of_three(lo, m, hi - 1)
        pivot = lo
        a, c = lo + 1, hi - 1
        while a < c and less(a, pivot):
            a += 1
        b = a
        while True:
            while b < c and not less(pivot, b):
                b += 1
            while b < c and less(pivot, c - 1):
                c -= 1
            if b >= c:
                break
            swap(b, c - 1)
            b += 1
            c -= 1
        protect = hi - c < 5
        if not protect and hi - c < (hi - lo) // 4:
            dups = 0
            if not less(pivot, hi - 1):
                swap(c, hi - 1)
                c += 1
                dups += 1
            if not less(b - 1, pivot):
                b -= 1
                dups += 1
            if not less(m, pivot):
                swap(m, b - 1)
                b -= 1
                dups += 1
            protect = dups > 1
        if protect:
            while True:
                while a < b and not less(b - 1, pivot):
                    b -= 1
                while a < b and less(a, pivot):
                    a += 1
                if a >= b:
                    break
                swap(a, b - 1)
                a += 1
                b -= 1
        swap(pivot, b - 1)
        return b - 1, c

    def quick_sort(a, b, max_depth):
        while b - a > 12:
            if max_depth == 0:
                heap_sort(a, b)
                return
            max_depth -= 1
            mlo, mhi = do_pivot(a, b)
            if mlo - a < b - mhi:
                quick_sort(a, mlo, max_depth)
                a = mhi
            else:
                quick_sort(mhi, b, max_depth)
                b = mlo
        if b - a > 1:
            for i in range(a + 6, b):
                if less(i, i - 6):
                    swap(i, i - 6)
            insertion_sort(a, b)

    n = len(data)
    depth, i = 0, n
    while i > 0:
        depth += 1
        i >>= 1
    quick_sort(0, n, depth * 2)
    return [t for _, t in data]
