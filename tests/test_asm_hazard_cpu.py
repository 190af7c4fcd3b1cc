"""attention_fwd.h (built by attention.hip and attention128.hip) issues its in-loop LDS reads as inline asm with hand-counted `s_waitcnt lgkmcnt` (the compiler would
otherwise order every LDS read behind all outstanding direct-to-LDS loads).  The compiler does not know those registers
are in flight, so it must never copy / spill / use one between the read and the wait that covers it.  This test
compiles the file to gfx950 assembly (hipcc cross-compiles without a GPU) and checks exactly that on both kernels."""
import os
import re
import subprocess
import tempfile

import pytest

from bmhrl_amd import build as B

CSRC = B.CSRC


def _regs(text):
    out = set()
    for a, b, c in re.findall(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", text):
        if c:
            out.add(int(c))
        else:
            out.update(range(int(a), int(b) + 1))
    return out


def _early_uses(lines):
    """Every LDS instruction (reads AND writes) and scalar memory read counts on lgkmcnt; LDS operations return in order."""
    ops, bad = [], []            # ops: outstanding lgkm operations, oldest first: (is_smem, registers it will write)
    for i, l in enumerate(lines):
        s = l.strip()
        if not s or s.startswith(";") or s.endswith(":"):
            continue
        op = s.split()[0]
        if op.startswith("ds_"):
            dst = _regs(s.split()[1].rstrip(",")) if op.startswith("ds_read") or "_rtn" in op or "permute" in op else set()
            ops.append((False, dst))
        elif op.startswith("s_load") or op.startswith("s_memtime") or op.startswith("s_buffer_load"):
            ops.append((True, set()))
        elif op == "s_waitcnt" and "lgkmcnt" in s:
            n = int(re.search(r"lgkmcnt\((\d+)\)", s).group(1))
            if n == 0:
                ops = []
            elif not any(sm for sm, _ in ops):        # scalar reads return out of order: only lgkmcnt(0) covers them
                ops = ops[len(ops) - n:] if n < len(ops) else ops
        elif op == "s_endpgm":
            ops = []
        else:
            pending = set().union(*[d for _, d in ops]) if ops else set()
            hit = sorted(r for r in _regs(" ".join(s.split()[1:])) if r in pending)
            if hit:
                bad.append((i, s, hit[:4]))
    return bad


@pytest.mark.parametrize("source", ["attention.hip", "attention128.hip"])
def test_no_use_of_asm_loaded_registers_before_their_wait(source):
    src = os.path.join(CSRC, source)
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "attention.s")
        cmd = [B.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, src] + \
            B.EXTRA_FLAGS.get(source, [])
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC)
        if r.returncode != 0:
            pytest.fail("hipcc failed:\n" + r.stderr[-2000:])
        text = open(out).read()
    kernels = re.findall(r"^(_ZN\S*attn_fwd_kernel\S*):[^\n]*\n(.*?)s_endpgm", text, flags=re.S | re.M)
    assert len(kernels) >= 2                      # every (QW, KW) split the entry point of the file can launch
    for name, body in kernels:
        lines = body.split("\n")
        assert sum("ds_read_b64_tr_b16" in l for l in lines) >= 16, name
        bad = _early_uses(lines)
        assert not bad, (name, bad[:5])


def test_checker_flags_an_early_use():
    hazard = ["ds_read_b64_tr_b16 v[10:11], v2 offset:64", "ds_read_b128 v[12:15], v3",
              "s_waitcnt lgkmcnt(1)", "v_mov_b32_e32 v20, v10", "v_mov_b32_e32 v21, v12"]
    bad = _early_uses(hazard)
    assert len(bad) == 1 and bad[0][2] == [12]
    assert not _early_uses(hazard[:2] + ["s_waitcnt lgkmcnt(0)"] + hazard[3:])
