"""pkg/metric mirrored: JaccardMetric(), CosineMetric(), DiceMetric(), ExactMetric(), OverlapMetric() are handles of the five
implementations the device has twins of (suggest_amd/csrc/engine.hip d_min_y..d_score: that is where the search runs).
`Metric` is also the reference's INTERFACE (pkg/metric/metric.go:7-16): any object with MinY / MaxY / Threshold / Distance —
a subclass, or any duck — is accepted wherever a metric is, and reaches the engine as tables its four methods fill
(NGramIndex.metric_tables -> sg_metric_tables_create).  The five built-ins carry the same four methods (IEEE doubles,
the evaluation order of pkg/metric/*.go) so that the tabulated path can be held against the native one."""
import math


class Metric:
    """metric.Metric (pkg/metric/metric.go:7-16).  code: the engine's enum for the built-ins, None for anything else."""
    name, code = "custom", None

    def __init__(self, name=None, code=None):
        if name is not None:
            self.name, self.code = name, code

    def MinY(self, alpha, size):                      # minimum cardinality of a matching set
        raise NotImplementedError

    def MaxY(self, alpha, size):
        raise NotImplementedError

    def Threshold(self, alpha, size_a, size_b):       # minimum overlap
        raise NotImplementedError

    def Distance(self, inter, size_a, size_b):
        raise NotImplementedError

    def __repr__(self):
        return "%sMetric()" % self.name.capitalize()


class _Jaccard(Metric):                               # pkg/metric/jaccard.go
    name, code = "jaccard", 0

    def MinY(self, alpha, size): return int(math.ceil(alpha * float(size)))
    def MaxY(self, alpha, size): return int(math.floor(float(size) / alpha))
    def Threshold(self, alpha, a, b): return int(math.ceil(alpha * float(a + b) / (1 + alpha)))
    def Distance(self, inter, a, b): return 1 - float(inter) / float(a + b - inter)


class _Cosine(Metric):                                # pkg/metric/cosine.go
    name, code = "cosine", 1

    def MinY(self, alpha, size): return int(math.ceil(alpha * alpha * float(size)))
    def MaxY(self, alpha, size): return int(math.floor(float(size) / (alpha * alpha)))
    def Threshold(self, alpha, a, b): return int(math.ceil(alpha * math.sqrt(float(a * b))))
    def Distance(self, inter, a, b): return 1 - float(inter) / math.sqrt(float(a * b))


class _Dice(Metric):                                  # pkg/metric/dice.go
    name, code = "dice", 2

    def MinY(self, alpha, size): return int(math.ceil(alpha / (2 - alpha) * float(size)))
    def MaxY(self, alpha, size): return int(math.floor((2 - alpha) / alpha * float(size)))
    def Threshold(self, alpha, a, b): return int(math.ceil(0.5 * alpha * float(a + b)))
    def Distance(self, inter, a, b): return 1 - float(2 * inter) / float(a + b)


class _Exact(Metric):                                 # pkg/metric/exact.go
    name, code = "exact", 3

    def MinY(self, alpha, size): return size
    def MaxY(self, alpha, size): return size
    def Threshold(self, alpha, a, b): return a
    def Distance(self, inter, a, b): return 0.0


class _Overlap(Metric):                               # pkg/metric/overlap.go
    name, code = "overlap", 4

    def MinY(self, alpha, size): return 1
    def MaxY(self, alpha, size): return 32767
    def Threshold(self, alpha, a, b): return int(math.ceil(alpha * min(float(a), float(b))))
    def Distance(self, inter, a, b): return 1 - float(inter) / min(float(a), float(b))


def JaccardMetric():
    return _Jaccard()


def CosineMetric():
    return _Cosine()


def DiceMetric():
    return _Dice()


def ExactMetric():
    return _Exact()


def OverlapMetric():
    return _Overlap()


class _Duck:
    """an object with the four methods of metric.Metric that cannot carry a `code` attribute itself"""
    code = None

    def __init__(self, inner):
        self._inner = inner
        self.MinY, self.MaxY, self.Threshold, self.Distance = inner.MinY, inner.MaxY, inner.Threshold, inner.Distance


BY_NAME = {"jaccard": JaccardMetric, "cosine": CosineMetric, "dice": DiceMetric, "exact": ExactMetric, "overlap": OverlapMetric}


def resolve(m):
    if isinstance(m, Metric) or all(hasattr(m, f) for f in ("MinY", "MaxY", "Threshold", "Distance")):
        if not hasattr(m, "code"):
            # a caller's own implementation of the interface: no device twin — it is tabulated (NGramIndex.metric_tables).  Wrapped, so
            # that every caller can read m.code and the caller's own object is left as it was (round 5 wrote the attribute onto it)
            m = _Duck(m)
        return m
    if isinstance(m, str):
        # metric names of the HTTP handler, internal/suggest/api/suggest_handler.go:26-34
        return BY_NAME[m.lower()]()
    raise TypeError("metric must be a Metric (MinY / MaxY / Threshold / Distance) or a name")
