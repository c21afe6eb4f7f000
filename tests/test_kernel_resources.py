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
def test_search_kernel_registers_and_stream_loop_schedule(tmp_path):
    src = os.path.join(ROOT, "suggest_amd", "csrc", "engine.hip")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "--cuda-device-only", "-O3", "-std=c++17", "-ffp-contract=off",
                        "-Rpass-analysis=kernel-resource-usage", "-S", src, "-o", str(tmp_path / "engine.s")],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    blocks = re.split(r"remark: Function Name: ", r.stderr)
    found = {}
    for b in blocks[1:]:
        name = b.split()[0]
        m = {k: int(v) for k, v in re.findall(r"remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", b)}
        found[name] = m
    # the batch kernel without threshold tightening: full and slim LDS layout (template <kParts, kLM, kTight, kSlim, kG8>)
    batch = [v for k, v in found.items() if re.search(r"sg_search_kernel_tILb0ELb0ELb0ELb[01]ELb0ELb0E", k)]
    assert len(batch) == 2, list(found)
    # ... and its instantiation for indexes with 8-bit-gap terms (the full layout; the registers of two wavefronts per SIMD)
    g8 = [v for k, v in found.items() if "sg_search_kernel_tILb0ELb0ELb0ELb0ELb1ELb0E" in k]
    assert len(g8) == 1 and g8[0]["ScratchSize [bytes/lane]"] <= 64 and g8[0]["Occupancy [waves/SIMD]"] >= 2, g8
    for b in batch:
        assert b["ScratchSize [bytes/lane]"] <= 64, b      # a few spilled dwords in cold code are fine (hot blocks checked below)
        assert b["Occupancy [waves/SIMD]"] >= 3, b
        assert b["VGPRs"] <= 168, b

    # [r5] the pipeline's launches (pipeline.inc).  The stream kernel: 64 registers and 80 scalar registers at most — eight
    # wavefronts per SIMD = four workgroups of eight per CU (at 82 .. 96 scalar registers the hardware admits seven, whatever the
    # compiler reports: three workgroups) — and nothing in scratch.
    res = {}
    for b in blocks[1:]:
        nm = b.split()[0]
        res[nm] = {k: int(v) for k, v in re.findall(r"remark:\s+(TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", b)}
    stream = [v for k, v in res.items() if "sg_stream_kernelILi" in k]
    assert len(stream) == 6, list(res)                                    # 8, 4 or 2 wavefronts (capi.inc tune_choice: by the index's query volume) x 4- / 8-byte sub-row descriptors
    for b in stream:
        assert b["VGPRs"] <= 64 and b["TotalSGPRs"] <= 80 and b["ScratchSize [bytes/lane]"] == 0 and b["Occupancy [waves/SIMD]"] == 8, b
    plan = [v for k, v in res.items() if "sg_plan_kernel" in k]
    assert len(plan) == 1, list(res)
    for b in plan:                                                         # (5 KB of LDS: 32 wavefronts per CU, if the registers allow 8 per SIMD)
        assert b["ScratchSize [bytes/lane]"] == 0 and b["Occupancy [waves/SIMD]"] == 8 and b["VGPRs"] <= 64, b
    plan2 = [v for k, v in res.items() if "sg_plan2_kernel" in k]          # [r6] two queries per wavefront: eight wavefronts per SIMD, nothing in scratch
    assert len(plan2) == 1 and plan2[0]["ScratchSize [bytes/lane]"] == 0 and plan2[0]["Occupancy [waves/SIMD]"] == 8 and plan2[0]["VGPRs"] <= 64, plan2
    verify = [v for k, v in res.items() if "sg_verify_kernel_t" in k]       # top-k rows in LDS / in HBM
    assert len(verify) == 2, list(res)
    for b in verify:
        assert b["ScratchSize [bytes/lane]"] == 0 and b["Occupancy [waves/SIMD]"] == 8 and b["TotalSGPRs"] <= 80, b
    # ... and its row loop: a row's seven counter atomics sit in blocks that wait for the row with vmcnt(1) (the next row's load
    # stays in flight), and the wait that ends the loop — vmcnt(0), before the last groups' barriers clear counters out of
    # registers the compiler takes for free — names the row registers
    asm_all = open(tmp_path / "engine.s").read().split("\n")
    for variant in ["_ZN2sg16sg_stream_kernelILi%dELb%dE" % (nw, wide) for nw in (8, 4, 2) for wide in (0, 1)]:
        start = next(i for i, l in enumerate(asm_all) if l.startswith(variant))
        end = next(i for i in range(start, len(asm_all)) if asm_all[i].startswith(".Lfunc_end"))
        body = [l.strip() for l in asm_all[start:end]]
        loads = [i for i, l in enumerate(body) if l.startswith("global_load_dwordx4")]
        assert len(loads) == 3, loads                                   # first row, and the two of the ping-pong
        assert sum(l == "s_waitcnt vmcnt(1)" for l in body) == 2
        last_w0 = max(i for i, l in enumerate(body) if l == "s_waitcnt vmcnt(0)")
        assert last_w0 > loads[-1]
        # no compiler-made copy OUT of a row register anywhere between the first row load and that wait (the signature of a value
        # moved while its load is in flight; the decode reads the registers with and / sdwa adds / shifts, never with v_mov)
        regs = set()
        for i in loads:
            m = re.search(r"v\[(\d+):(\d+)\]", body[i])
            regs |= set(range(int(m.group(1)), int(m.group(2)) + 1))
        for l in body[loads[0]:last_w0]:
            if l.startswith("v_mov") and "dpp" not in l:
                dst, src = l.split(None, 1)[1].split(",", 1)

                def named(t):
                    out = {int(x) for x in re.findall(r"\bv(\d+)\b", t)}
                    for a_, b_ in re.findall(r"v\[(\d+):(\d+)\]", t):
                        out |= set(range(int(a_), int(b_) + 1))
                    return out
                # (a row register set from another: the zero-initialisation of a buffer no load has been issued into)
                assert not (named(src) & regs) or (named(dst) <= regs), l

    # the stream loop keeps the next batch's four row loads in flight while it counts the current batch: the blocks
    # that issue the LDS counter atomics wait with vmcnt(4), never with vmcnt(0), and touch no scratch
    asm = open(tmp_path / "engine.s").read().split("\n")
    for variant in ("_ZN2sg18sg_search_kernel_tILb0ELb0ELb0ELb0ELb0ELb0E", "_ZN2sg18sg_search_kernel_tILb0ELb0ELb0ELb1ELb0ELb0E"):
        start = next(i for i, l in enumerate(asm) if l.startswith(variant))
        end = next(i for i in range(start, len(asm)) if asm[i].startswith(".Lfunc_end"))
        blocks, cur = [], []
        for l in asm[start:end]:
            if re.match(r"^\.LBB\d+_\d+:", l):
                blocks.append(cur)
                cur = []
            elif l.startswith("\t") and not l.strip().startswith((";", ".")):
                cur.append(l.strip())
        blocks.append(cur)
        hot = [b for b in blocks if sum("ds_add_rtn_u32" in x for x in b) >= 8]
        assert len(hot) >= 2
        for b in hot:
            assert not any("scratch_" in x for x in b)
            assert not any("s_waitcnt vmcnt(0)" in x for x in b), [x for x in b if "s_waitcnt" in x]
        assert any("s_waitcnt vmcnt(4)" in x for b in hot for x in b)
