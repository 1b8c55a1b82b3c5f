"""The hand-placed wait states of the inline-assembly DPP blocks (aac.js_amd/csrc/devport.h: dp_mirror8/16_valu,
dp_window_mirror) are invisible to the compiler's hazard recogniser, so they are checked here on the code that ships: the
gfx950 code objects inside libaacgpu.so are disassembled and every DPP instruction must sit
  * at least 2 wait states behind a VALU write of the register its DPP operand reads, and
  * at least 5 wait states behind a write of EXEC
(an instruction is one wait state, `s_nop N` is N + 1) — within the straight-line code in front of it, up to the nearest
label.  A CPU test: it needs llvm-objdump from the ROCm toolchain and the built library, no GPU."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "aac.js_amd", "csrc", "libaacgpu.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs(token):
    out = set()
    for m in REG.finditer(token):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def disassemble(tmp):
    shutil.copy(LIB, os.path.join(tmp, "lib.so"))
    subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=tmp, check=True, capture_output=True)
    text = []
    for f in sorted(os.listdir(tmp)):
        if "amdgcn" in f:
            text.append(subprocess.run([OBJDUMP, "-d", f], cwd=tmp, check=True, capture_output=True, text=True).stdout)
    return "\n".join(text)


@pytest.mark.skipif(not (os.path.exists(OBJDUMP) and os.path.exists(LIB)), reason="needs llvm-objdump and the built library")
def test_dpp_blocks_keep_their_wait_states(tmp_path):
    lines = [l.split("//")[0].strip() for l in disassemble(str(tmp_path)).splitlines()]
    n_dpp = 0
    for i, ins in enumerate(lines):
        if "_dpp " not in ins:
            continue
        n_dpp += 1
        ops = ins.split(None, 1)[1].split(",")
        src = regs(ops[1])                                    # the operand the DPP control applies to
        states = 0
        for back in range(i - 1, max(-1, i - 12), -1):
            prev = lines[back]
            if not prev or prev.endswith(":") or prev.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
                break                                         # a label / a branch: another path joins, nothing more is known
            op = prev.split()[0]
            if op.startswith("v_") and not op.startswith("v_cmp") and "," in prev:
                dst = regs(prev.split(None, 1)[1].split(",")[0])
                assert not (dst & src) or states >= 2, "line %d: %s\n  reads a register written %d wait states earlier by\n  %s" % (i, ins, states, prev)
            writes_exec = (op.startswith("s_") and re.search(r"^\S+\s+exec", prev)) or op.startswith("v_cmpx")
            assert not writes_exec or states >= 5, "line %d: %s\n  is %d wait states behind an EXEC write:\n  %s" % (i, ins, states, prev)
            states += int(prev.split()[1]) + 1 if op == "s_nop" else 1
            if states >= 5:
                break
    assert n_dpp >= 64, "the DPP blocks are gone from the library? (%d DPP instructions found)" % n_dpp


@pytest.mark.skipif(not (os.path.exists(OBJDUMP) and os.path.exists(LIB)), reason="needs llvm-objdump and the built library")
def test_exec_masked_blocks_restore_exec(tmp_path):
    """M/S and intensity stereo run under an EXEC mask per coefficient group (devport.h: dp_sumdiff_where, dp_scale_where):
    inline assembly that narrows EXEC behind the compiler's back.  In the shipped code every such block — an
    s_and_saveexec_b64 followed at once by packed arithmetic — must hold nothing but v_pk_* instructions and end by putting
    back the very register pair it saved EXEC in."""
    lines = [l.split("//")[0].strip() for l in disassemble(str(tmp_path)).splitlines()]
    n_blocks = 0
    for i, ins in enumerate(lines):
        if not ins.startswith("s_and_saveexec_b64") or i + 1 >= len(lines) or not lines[i + 1].startswith(("v_pk_add_f32", "v_pk_mul_f32")):
            continue
        saved = ins.split(None, 1)[1].split(",")[0].strip()
        n_blocks += 1
        for j in range(i + 1, i + 10):
            nxt = lines[j]
            if nxt.startswith("s_mov_b64 exec"):
                assert nxt.split(",")[1].strip() == saved, "line %d: %s saved EXEC in %s, line %d restores %s" % (i, ins, saved, j, nxt)
                break
            assert nxt.startswith("v_pk_"), "line %d: %s inside an EXEC-masked block that began at line %d" % (j, nxt, i)
        else:
            raise AssertionError("line %d: %s is never undone" % (i, ins))
    assert n_blocks >= 8, "the EXEC-masked M/S and intensity blocks are gone from the library? (%d found)" % n_blocks
