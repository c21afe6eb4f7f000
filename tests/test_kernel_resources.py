"""Guards the search kernel's register budget: one more live value in the hot loop and hipcc spills to scratch or drops
to two waves per SIMD — which cost 40 % of the headline once (the language-model code inlined into the shared kernel).
Compiles the device code with -Rpass-analysis=kernel-resource-usage (no GPU needed)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_search_kernel_has_no_scratch_and_three_waves_per_simd(tmp_path):
    src = os.path.join(ROOT, "suggest_amd", "csrc", "engine.hip")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "--cuda-device-only", "-O3", "-std=c++17", "-ffp-contract=off",
                        "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", str(tmp_path / "engine.o")],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    blocks = re.split(r"remark: Function Name: ", r.stderr)
    found = {}
    for b in blocks[1:]:
        name = b.split()[0]
        m = {k: int(v) for k, v in re.findall(r"remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", b)}
        found[name] = m
    batch = [v for k, v in found.items() if "sg_search_kernel_tILb0ELb0" in k]
    assert len(batch) == 1, list(found)
    assert batch[0]["ScratchSize [bytes/lane]"] == 0, batch[0]
    assert batch[0]["Occupancy [waves/SIMD]"] >= 3, batch[0]
    assert batch[0]["VGPRs"] <= 168, batch[0]
