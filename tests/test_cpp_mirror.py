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


TWIN = os.path.join(CPP, "_build", "shim_twin_test")


@pytest.mark.gpu
def test_shim_call_sequences_from_a_compiled_host():
    """[r5] go/suggesthip/suggesthip.go has never met a Go compiler: tests/cpp/shim_twin_test.cpp makes the same calls in the same
    order from C++ — the pipelined dispatcher over pinned slots and two tickets, a foreign metric.Metric as tables with the cache's
    and the calls' references, every candidate paged through sg_suggest_batch_from into a foreign collector, k discovery, Close
    while callers are in flight (pkg/suggest/service_test.go:36-79, collector.go:136-191)."""
    _binary()
    r = subprocess.run([TWIN, GOLDEN, "--stress", "12"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 failed" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("san", ["asan", "tsan"])
def test_shim_call_sequences_under_the_sanitizers(san):
    """the same program built with -fsanitize=address,undefined / thread (the reference runs its tests under the race detector,
    SURVEY.md §5).  The host layer of libsuggest_hip.so is not instrumented, so this covers the caller's side of the ABI contract:
    buffers alive until the wait returns, no row written past a slot, no race on what the caller owns."""
    subprocess.run(["make", "-C", CPP, "sanitizers"], check=True, capture_output=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0", TSAN_OPTIONS="report_bugs=1:halt_on_error=0:suppressions=" + os.path.join(CPP, "tsan.supp"))
    import shutil
    cmd = [TWIN + "_" + san, GOLDEN, "--stress", "4"]
    if san == "tsan" and shutil.which("setarch"):            # (ThreadSanitizer wants its fixed mappings: no address-space randomisation)
        cmd = ["setarch", "x86_64", "-R"] + cmd
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    if r.returncode != 0 and not r.stdout and ("unexpected memory mapping" in r.stderr or "Shadow memory range interleaves" in r.stderr or "setarch" in r.stderr):
        pytest.skip("the sanitizer runtime cannot map its shadow on this box: " + r.stderr[-200:])
    assert " 0 failed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert "WARNING: ThreadSanitizer" not in r.stderr and "ERROR: AddressSanitizer" not in r.stderr, r.stderr[-3000:]
