import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

try:      # before libsuggest_hip.so is loaded: torch and the library must resolve the same libamdhip64 (suggest_amd/_lib.py) —
    import torch  # noqa: F401  a test that imports torch after the library finds "no HIP GPUs"
except Exception:
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def reference_tests():
    import json
    with open(os.path.join(GOLDEN, "reference_tests.json"), encoding="utf-8") as f:
        return json.load(f)


@pytest.fixture(scope="session")
def cars_lines():
    # bufio.Scanner semantics (pkg/dictionary/helpers.go:41-45): split on \n, drop one trailing \r
    with open(os.path.join(GOLDEN, "cars.dict"), "rb") as f:
        data = f.read()
    lines = data.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    return [l[:-1] if l.endswith(b"\r") else l for l in lines]


@pytest.fixture(scope="session")
def words_lines():
    import lzma
    with open(os.path.join(GOLDEN, "words.dict.xz"), "rb") as f:
        data = lzma.decompress(f.read())
    lines = data.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    return [l[:-1] if l.endswith(b"\r") else l for l in lines]


CARS_DESC = dict(ngram_size=3, wrap=("$", "$"), pad="$", alphabet=("russian", "english", "numbers", "$"))
WORDS_DESC = dict(ngram_size=3, wrap=("^", "$"), pad="$", alphabet=("english", "numbers", "$^"))
