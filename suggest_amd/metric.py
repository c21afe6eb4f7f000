"""Metric handles mirroring pkg/metric (JaccardMetric(), CosineMetric(), DiceMetric(), ExactMetric(),
OverlapMetric()).  The maths itself runs on the device (suggest_amd/csrc/engine.hip d_min_y..d_score)."""


class Metric:
    def __init__(self, name, code):
        self.name, self.code = name, code

    def __repr__(self):
        return "%sMetric()" % self.name.capitalize()


def JaccardMetric():
    return Metric("jaccard", 0)


def CosineMetric():
    return Metric("cosine", 1)


def DiceMetric():
    return Metric("dice", 2)


def ExactMetric():
    return Metric("exact", 3)


def OverlapMetric():
    return Metric("overlap", 4)


BY_NAME = {"jaccard": JaccardMetric, "cosine": CosineMetric, "dice": DiceMetric, "exact": ExactMetric, "overlap": OverlapMetric}


def resolve(m):
    if isinstance(m, Metric):
        return m
    if isinstance(m, str):
        # metric names of the HTTP handler, internal/suggest/api/suggest_handler.go:26-34
        return BY_NAME[m.lower()]()
    raise TypeError("metric must be a Metric or a name")
