"""The C++ mirror of the reference's Go API (include/suggest_hip.hpp) run through the reference's own tests
(tests/cpp/service_test.cpp): host-side logic on CPU, the searches on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
BIN = os.path.join(CPP, "_build", "service_test")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _binary():
    if not os.path.exists(BIN):            # normally built by __graft_entry__.build()
        subprocess.run(["make", "-C", CPP], check=True, capture_output=True)
    return BIN


def test_cpp_mirror_host_logic():
    r = subprocess.run([_binary(), "--cpu", GOLDEN], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 failed" in r.stdout


@pytest.mark.gpu
def test_cpp_mirror_reference_tests():
    r = subprocess.run([_binary(), GOLDEN], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 failed" in r.stdout


@pytest.mark.gpu
def test_cpp_mirror_queries_during_index_swaps_stress():
    """service_test.go:36-79, 150 times over: workers on their own streams while the main thread builds, uploads and frees
    indexes on the same device.  (Round 2 found rows of one caller's batch in another's here — about 1 run in 50 — while
    the call's device block came from the stream-ordered pool; the blocks are per-thread now.)"""
    r = subprocess.run([_binary(), GOLDEN], capture_output=True, text=True, timeout=900, env=dict(os.environ, SG_STRESS="150"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 failed" in r.stdout
