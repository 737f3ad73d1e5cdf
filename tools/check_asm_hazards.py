#!/usr/bin/env python3
"""Static check of the hand-written inline assembly in the persistent kernel's ISA (no GPU needed):

  hipcc pads data hazards only between instructions it emitted itself -- it does not look inside `asm volatile` blocks.  Two hazards
  of gfx950 can therefore sit at the EDGE of a block:
    * a VALU write of an SGPR (v_readlane / v_readfirstlane: SGPR spill reloads, uniform values) followed within 5 wait states by a
      vector-memory instruction inside the block that uses that SGPR as its address -> the access goes to a stale address;
    * the register a returning atomic of a block writes asynchronously being read, moved or overwritten by compiler code before
      the instruction that consumes it (tools/check_asm_hazards.py --drawn: the tile queue's draw in igemm_persist.hip).
usage: check_asm_hazards.py [file.hip ...]   (default: yolo-v1_amd/csrc/igemm_persist.hip); exit code 1 on a finding."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "yolo-v1_amd", "csrc")
files = [a for a in sys.argv[1:] if not a.startswith("--")] or [os.path.join(CSRC, "igemm_persist.hip")]
WAIT = 5


def sregs(tok):
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"s(\d+)", tok)
    return {int(m.group(1))} if m else set()


def nops(ins):
    m = re.match(r"s_nop (\d+)", ins)
    return int(m.group(1)) + 1 if m else 1


bad = 0
for f in files:
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
                               "-S", "--cuda-device-only", f, "-o", out], stderr=subprocess.DEVNULL)
        lines = open(out).read().splitlines()
    kernel, in_asm = "?", False
    body = []          # (instruction text, inside an asm block) of the current kernel, labels dropped
    for ln in lines:
        t = ln.strip()
        if re.match(r"^_Z\w+:", t):
            kernel, body = t.split(":")[0], []
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        ins = t.split(";")[0].strip()
        if in_asm and re.match(r"(global|buffer|flat)_", ins):
            used = set()
            for tok in re.split(r"[,\s]+", ins):
                used |= sregs(tok)
            dist = 0
            for prev, prev_asm in reversed(body):
                if dist >= WAIT:
                    break
                m = re.match(r"v_(readlane|readfirstlane)_b32 (s\d+)", prev)
                if m and sregs(m.group(2)) & used:
                    print(f"{os.path.basename(f)}: {kernel}: `{prev}` only {dist} wait state(s) before `{ins}`")
                    bad += 1
                dist += nops(prev)
        body.append((ins, in_asm))
print("inline-assembly hazards:", bad)
sys.exit(1 if bad else 0)
