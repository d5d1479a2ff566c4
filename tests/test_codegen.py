"""Generated-code invariants the kernels' hand-counted waits rely on (CPU: hipcc cross-compiles gfx950 without a GPU).

csrc/ffn2.hip, projection prologue (instances 21 / 50 / 51): chunk c of to_out's weight is admitted by `s_waitcnt vmcnt(5)` - "at most
the FIVE vector-memory instructions of chunk c + 1 are younger" (three LDS-DMA + two opaque global loads per wave and chunk).  That
only holds if hipcc puts no vector-memory instruction of its own (a spill, a hoisted load) between two of those waits; the q/kv
epilogue (instance 51) likewise counts two DMA instructions per step (`vmcnt(2)`).  This test reads the assembly and counts."""
import os
import re
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(os.path.dirname(HERE), "isp_tts_amd", "csrc", "ffn2.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
VM = re.compile(r"^\s*(global_|scratch_|buffer_|flat_)")


@pytest.fixture(scope="module")
def ffn2_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("asm") / "ffn2.s"
    from isp_tts_amd import build
    cmd = [HIPCC, *build.FLAGS, *build.EXTRA_FLAGS.get("ffn2.hip", []), "-S", "--cuda-device-only", SRC, "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True)
    text = out.read_text()
    kernels = {}
    for m in re.finditer(r"^_ZN12_GLOBAL__N_116ffn2_bf16_kernelILi(\d+)EEEvNS_10Ffn2ParamsE:[^\n]*\n(.*?)\n\s*\.end_amdhsa_kernel", text, re.S | re.M):
        kernels[int(m.group(1))] = m.group(2).split("\n")
    return kernels


def _between(lines, marker):
    """VM instruction counts between consecutive lines that contain `marker`."""
    idx = [i for i, l in enumerate(lines) if marker in l and i > 0 and "#ASMSTART" in lines[i - 1]]    # the hand-written waits only
    return idx, [sum(1 for l in lines[a + 1:b] if VM.match(l)) for a, b in zip(idx, idx[1:])]


@pytest.mark.parametrize("mode", [21, 50, 51])
def test_projection_prologue_waits_count_what_is_in_flight(ffn2_asm, mode):
    lines = ffn2_asm[mode]
    idx, counts = _between(lines, "s_waitcnt vmcnt(5)")
    # the prologue exists twice (once per stage order of the main loop): 2 x 11 counted waits; consecutive waits of ONE copy are
    # separated by exactly the five instructions of one chunk (steps 0 .. 9 request chunks 2 .. 11; step 11 waits for everything)
    assert len(idx) == 22, len(idx)
    per_copy = [counts[:10], counts[11:21]]
    for c in per_copy:
        assert c == [5] * 10, c
    # and no private-segment traffic anywhere between the first and the last counted wait of a copy
    for a, b in ((idx[0], idx[10]), (idx[11], idx[21])):
        assert not any("scratch_" in l for l in lines[a:b])


def test_qkv_epilogue_waits_count_what_is_in_flight(ffn2_asm):
    lines = ffn2_asm[51]
    idx = [i for i, l in enumerate(lines) if "s_waitcnt vmcnt(2)" in l and "#ASMSTART" in lines[i - 1]]
    assert len(idx) == 1                                   # the loop body exists once
    # one iteration of the (rotated) loop = from the label its back edge targets to that back edge: the two DMA instructions of
    # chunk c + 2 and no other vector-memory instruction
    labels = {l.split(":")[0]: i for i, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:", l)}
    back = None
    for i in range(idx[0], min(len(lines), idx[0] + 400)):      # (the step without a request branches back early: take the LAST back edge)
        m = re.match(r"^\s*s_c?branch\w* (\.LBB\d+_\d+)", lines[i])
        if m and labels.get(m.group(1), len(lines)) < idx[0] and (back is None or labels[m.group(1)] == back[0]):
            back = (labels[m.group(1)], i)
    assert back is not None
    vm = [l.strip().split()[0] for l in lines[back[0]:back[1]] if VM.match(l)]
    assert vm == ["global_load_lds_dwordx4", "global_load_lds_dwordx4"], vm


def test_no_instance_of_the_kernel_spills_in_its_main_loop(ffn2_asm):
    """The main loop (between the first `s_setprio 1` after the prologue and the epilogue) of every product instance is free of
    private-segment traffic: a spill there costs a memory round trip per 32-hidden chunk."""
    for mode in (0, 20, 21, 50, 51):
        lines = ffn2_asm[mode]
        loops = [i for i, l in enumerate(lines) if re.match(r"^\s*s_cbranch_scc\d \.LBB", l)]
        spills = [i for i, l in enumerate(lines) if "scratch_" in l]
        # every spill / reload sits outside the two big loop bodies: none between a loop label and its backward branch
        labels = {l[:-1]: i for i, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:$", l)}
        for b in loops:
            tgt = lines[b].split()[-1]
            a = labels.get(tgt)
            if a is not None and a < b and b - a > 400:        # a backward branch over a long body: a main-loop copy
                assert not any(a < s < b for s in spills), (mode, a, b)
