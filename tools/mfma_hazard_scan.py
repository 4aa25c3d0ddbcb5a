"""Scans the gfx950 disassembly of the library's objects for data hazards around matrix instructions that the compiler
cannot see because they are issued from inline asm (conv3x3_f16x3_v3.hip, attention_f16x3.hip, ...):

  R1  a VALU instruction writes a VGPR that a following v_mfma reads as SrcA / SrcB within < 2 wait states;
  R2  an instruction reads an accumulator register (v_accvgpr_read, a VALU / memory op with an a[..] source) that a
      v_mfma wrote less than passes + 2 wait states earlier (16x16x32_f16: 4 passes, 32x32x16_f16: 8).

Every instruction counts as one wait state, `s_nop k` as k + 1; waits that depend on s_waitcnt / barriers are ignored
(conservative: they only add time).  The scan follows straight-line code and stops looking back at a branch target
label or a branch.  Usage: python tools/mfma_hazard_scan.py [object ...] (default: every object of the build);
tests/test_abi_and_host.py runs it on every build (`scan_objects`), so a compiler bump that moves a register copy in
front of an asm MFMA fails the CPU suite instead of corrupting a launch silently.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
REG = re.compile(r"\b([va])(?:\[(\d+):(\d+)\]|(\d+)\b)")


def _regs(operand):
    """set of ('v' | 'a', index) named by one operand"""
    out = set()
    for m in REG.finditer(operand):
        kind = m.group(1)
        if m.group(4) is not None:
            out.add((kind, int(m.group(4))))
        else:
            out.update((kind, i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def _parse(line):
    """'\tv_mfma_f32_16x16x32_f16 a[0:3], v[2:5], v[6:9], a[0:3]  // 0000..' -> (mnemonic, [operands])"""
    text = line.split("//")[0].strip()
    if not text or text.endswith(":"):
        return None
    parts = text.split(None, 1)
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    return parts[0], ops


PASSES = {"32x32x16_f16": 8, "32x32x16_bf16": 8, "16x16x32_f16": 4, "16x16x32_bf16": 4, "32x32x2_f32": 16,
          "16x16x4_f32": 8, "32x32x8_f16": 16, "16x16x16_f16": 8}      # 4 clocks per pass (gfx950)


def _mfma_latency(mn):
    """wait states between a matrix instruction and a read of its result by another unit: passes + 2 (the distance
    hipcc keeps for the MFMAs it knows about); unknown shapes are priced as 16 passes"""
    for shape, passes in PASSES.items():
        if mn.endswith(shape):
            return passes + 2
    return 18


def disassemble(obj):
    with tempfile.TemporaryDirectory() as td:
        local = os.path.join(td, os.path.basename(obj))
        with open(obj, "rb") as f, open(local, "wb") as g:
            g.write(f.read())
        subprocess.run([OBJDUMP, "--offloading", local], cwd=td, check=True, capture_output=True)
        dev = [f for f in os.listdir(td) if "amdgcn" in f]
        if not dev:
            return ""
        return subprocess.run([OBJDUMP, "-d", os.path.join(td, dev[0])], check=True, capture_output=True, text=True).stdout


def scan_text(asm):
    """-> list of (kernel, rule, offending line, the instruction it conflicts with)"""
    findings = []
    kernel = "?"
    window = []      # (wait states since, mnemonic, operands, line) most recent last
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            name = m.group(1)
            if not name.startswith("L") and "BB" not in name:
                kernel = name
            window = []          # a label: control may arrive from elsewhere
            continue
        ins = _parse(line)
        if ins is None:
            continue
        mn, ops = ins
        if mn.startswith("v_mfma") or mn.startswith("v_smfmac"):
            src = _regs(ops[1]) | _regs(ops[2])
            dist = 0
            for w_states, w_mn, w_ops, w_line in reversed(window):
                if dist >= 2:
                    break
                if w_mn.startswith("v_") and not w_mn.startswith(("v_mfma", "v_smfmac", "v_cmp")) and w_ops:
                    if _regs(w_ops[0]) & src:
                        findings.append((kernel, "R1", line.strip(), w_line.strip()))
                dist += w_states
        else:
            # accumulator reads by anything that is not a matrix instruction
            reads = set()
            for i, o in enumerate(ops):
                if i == 0 and not mn.startswith(("buffer_store", "global_store", "ds_write", "scratch_store", "flat_store")):
                    continue
                reads |= {r for r in _regs(o) if r[0] == "a"}
            if reads:
                dist = 0
                for w_states, w_mn, w_ops, w_line in reversed(window):
                    if w_mn.startswith("v_mfma") and _regs(w_ops[0]) & reads and dist < _mfma_latency(w_mn):
                        findings.append((kernel, "R2", line.strip(), w_line.strip()))
                        break
                    dist += w_states
                    if dist >= 19:
                        break
        states = 1
        if mn == "s_nop":
            states = int(ops[0], 0) + 1
        window.append((states, mn, ops, line))
        if len(window) > 40:
            window.pop(0)
        if mn.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc")):
            window = []
    return findings


def scan_objects(objs=None):
    if objs is None:
        d = os.path.join(ROOT, "mulan_amd", "csrc", "_obj")
        objs = sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(".o"))
    out = []
    for obj in objs:
        for f in scan_text(disassemble(obj)):
            out.append((os.path.basename(obj),) + f)
    return out


if __name__ == "__main__":
    found = scan_objects(sys.argv[1:] or None)
    for f in found:
        print(" | ".join(f))
    print(f"{len(found)} hazard(s)")
    sys.exit(1 if found else 0)
