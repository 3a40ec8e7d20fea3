#!/usr/bin/env python
"""The sweep kernels issue their matrix instructions (v_mfma_f64_4x4x4_4b_f64)
from inline asm, which the compiler's hazard recogniser does not look into.  This
scans the generated ISA of every kernel instance for the data hazards that can then
go unpadded (wait states as in LLVM's GCNHazardRecognizer for gfx90a+ DGEMM 4x4):

  R1  VALU write of a VGPR -> MFMA reads it              needs 2 wait states
  R2  MFMA write of a VGPR -> VALU reads / overwrites it needs 6
  R3  MFMA write of a VGPR -> VMEM / LDS / FLAT reads it needs 9  (e.g. a SPILL of
      an accumulator right behind the slot sequence)
  R4  MFMA write -> MFMA reads it as SrcC                needs 4
  A2  a scratch access (spill) in an instance of a sweep kernel (sweep.hip, sweep_pair.hip,
      sweep_mid.hip)
  A1  (k_sweep only) an AccVGPR named by an instruction outside the inline asm
      blocks: the accumulators live in hand-assigned AccVGPRs (csrc/sweep_slots.h)

    python scripts/dev/check_mfma_hazards.py [extra hipcc flags ...]

Exit status 1 when a hazard is found; tests/test_abi.py runs it over the shipped
sources.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "safeopt_amd", "csrc")
NEED = {"R1": 2, "R2": 6, "R3": 9, "R4": 4, "A1": 0, "A2": 0}
COUNT = {}
HORIZON = 10


_ARITH = re.compile(r"^[0-9+\-*() ]+$")


def _num(expr):
    """Register indices in hand-written asm are assembler expressions (8*(15)+2*0)."""
    expr = expr.strip()
    if not _ARITH.match(expr):
        raise ValueError(expr)
    return int(eval(expr, {"__builtins__": {}}, {}))


def regs(tok):
    """VGPRs / AccVGPRs an operand names, as ('v' | 'a', index) pairs."""
    tok = tok.strip(",")
    m = re.match(r"-?\|?([va])\[([^:\]]+):([^\]]+)\]", tok)
    if m:
        try:
            return {(m.group(1), i) for i in range(_num(m.group(2)), _num(m.group(3)) + 1)}
        except ValueError:
            return set()
    m = re.match(r"-?\|?([va])(\d+)\b", tok)
    return {(m.group(1), int(m.group(2)))} if m else set()


def split_args(rest):
    """Operands of one instruction: commas outside brackets separate them."""
    out, depth, cur = [], 0, ""
    for ch in rest:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if (ch == "," or ch.isspace()) and depth == 0:
            if cur:
                out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur:
        out.append(cur)
    return out


def isa(path, flags):
    extra = ["-Wno-inline-asm"] + (["-mllvm", "-amdgpu-spill-vgpr-to-agpr=0"]
                                   if path.endswith("sweep.hip") else [])
    r = subprocess.run(
        ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I",
         os.path.join(ROOT, "include"), "-I", CSRC, "-S", "--cuda-device-only", "-o", "-",
         path] + extra + flags, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed on %s:\n%s" % (path, r.stderr[-2000:]))
    return r.stdout


def scan(asm, verbose=True):
    bad, func = [], "?"
    valu_w, mfma_w = [], []     # [wait states since, vgprs, text]
    in_asm = False              # only MFMAs inside inline asm are unknown to the compiler
    for line in asm.split("\n"):
        if "#ASMSTART" in line:
            in_asm = True
        elif "#ASMEND" in line:
            in_asm = False
        t = line.split(";")[0].strip()
        if not t:
            continue
        if t.endswith(":"):
            if t.startswith("_Z"):
                func = t[:-1]
                valu_w, mfma_w = [], []
            continue
        if t.startswith("."):
            continue
        toks = t.split(None, 1)
        op, args = toks[0], split_args(toks[1] if len(toks) > 1 else "")
        if op.startswith("v_mfma") and in_asm:
            COUNT["asm_mfma"] = COUNT.get("asm_mfma", 0) + 1
        # A1: the accumulators of the 4-wave sweep are hand-assigned AccVGPRs
        # (csrc/sweep_slots.h): in its instances no instruction of the COMPILER's may
        # name an AccVGPR (as spill space, as a copy target, as a load destination)
        if not in_asm and "k_sweepI" in func and (
                "accvgpr" in op or any(r[0] == "a" for a in args for r in regs(a))):
            bad.append((func, "A1", 0, "(compiler-generated)", t))
        # A2: ... and no instance of either sweep may spill: a scratch access between the
        # asm statements is a register copy the hazard rules below would have to know
        # about (round 3, finding (b)), and costs 5 x in the stage loop besides
        if op.startswith("scratch_") and ("k_sweepI" in func or "k_sweep_pairI" in func or
                                          "k_sweep_midI" in func):
            bad.append((func, "A2", 0, "(spill)", t))
        is_mfma = op.startswith("v_mfma") and in_asm
        is_valu = op.startswith("v_") and not op.startswith("v_mfma")
        is_mem = op.startswith(("ds_", "global_", "scratch_", "flat_", "buffer_"))
        rd = set()
        wr = set()
        if is_mfma:
            wr = regs(args[0])
            srcs = [regs(a) for a in args[1:4]]
            for ws, w, txt in valu_w:
                if ws < NEED["R1"] and any(w & s for s in srcs):
                    bad.append((func, "R1", ws, txt, t))
            for ws, w, txt in mfma_w:
                if ws < NEED["R4"] and (w & srcs[2]):
                    bad.append((func, "R4", ws, txt, t))
        elif is_valu or is_mem:
            allr = [regs(a) for a in args]
            if is_valu and not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
                wr = allr[0] if allr else set()
                rd = set().union(*allr[1:]) if len(allr) > 1 else set()
            elif is_valu:
                rd = set().union(*allr) if allr else set()
            else:
                # loads write their first operand, stores / addresses are reads
                if re.match(r"(ds_read|ds_bpermute|ds_swizzle|global_load|scratch_load|flat_load|buffer_load)", op) \
                        and "lds" not in op:
                    wr = allr[0] if allr else set()
                    rd = set().union(*allr[1:]) if len(allr) > 1 else set()
                else:
                    rd = set().union(*allr) if allr else set()
            need = NEED["R2"] if is_valu else NEED["R3"]
            for ws, w, txt in mfma_w:
                if ws < need and (w & (rd | wr)):
                    bad.append((func, "R2" if is_valu else "R3", ws, txt, t))
        n = 1
        if op == "s_nop":
            n = int(args[0]) + 1
        valu_w = [[ws + n, w, txt] for ws, w, txt in valu_w if ws + n < HORIZON]
        mfma_w = [[ws + n, w, txt] for ws, w, txt in mfma_w if ws + n < HORIZON]
        if is_mfma:
            mfma_w.append([0, wr, t])
        elif is_valu and wr:
            valu_w.append([0, wr, t])
    if verbose:
        for func, rule, ws, a, b in bad[:40]:
            print("%s in %s (%d wait states, needs %d):\n    %s\n    %s" %
                  (rule, func, ws, NEED[rule], a, b))
    return bad


def main(flags):
    total = 0
    for f in ("sweep.hip", "sweep_pair.hip"):
        COUNT.clear()
        total += len(scan(isa(os.path.join(CSRC, f), flags)))
        # (an empty or truncated listing must not pass as "0 hazards")
        n = COUNT.get("asm_mfma", 0)
        print("%s: %d inline-asm MFMAs scanned" % (f, n))
        if n < 1000:
            print("too few matrix instructions in the ISA of %s: scan is void" % f)
            total += 1
    # sweep_mid.hip issues its matrix instructions through the builtin (the compiler pads
    # them itself): only rule A2 -- no instance may spill
    COUNT.clear()
    total += len(scan(isa(os.path.join(CSRC, "sweep_mid.hip"), flags)))
    print("sweep_mid.hip: scanned for spills")
    print("hazards found: %d" % total)
    return total


if __name__ == "__main__":
    sys.exit(1 if main(sys.argv[1:]) else 0)
