"""Build libsafeopt_hip.so (hipcc, gfx950 only) in-tree.

    python -m safeopt_amd.build [--force] [--verbose] [--scan]

The library is the whole device side of the product: hand-written HIP kernels
plus the C ABI declared in include/safeopt_hip.h.  hipcc cross-compiles for
gfx950 without a GPU, so this also is the "does it build" check.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OUT = os.path.join(PKG, "libsafeopt_hip.so")
SOURCES = ["api.hip", "sweep.hip", "sweep_pair.hip", "sweep_mid.hip", "sweep_tiny.hip",
           "step_small.hip",
           "factor.hip", "sets.hip", "swarm.hip"]
HEADERS = [os.path.join(CSRC, h) for h in ("common.h", "kern_eval.h", "fitness.h",
                                            "small_path.h", "sweep_shared.h",
                                            "sweep_slots.h", "set_order.h",
                                            "tiny_row.h", "sets_front.h")] + \
          [os.path.join(REPO, "include", "safeopt_hip.h")]
BASE = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
        "-I", os.path.join(REPO, "include"), "-I", CSRC, "-Wall",
        "-Wno-unused-function"]
# sweep.hip: the accumulators of the 4-wave sweep live in hand-assigned AccVGPRs
# (csrc/sweep_slots.h); the compiler must not park spilled VGPRs there.
EXTRA = {"sets.hip": ["-ffp-contract=off"], "swarm.hip": ["-ffp-contract=off"],
         "sweep.hip": ["-mllvm", "-amdgpu-spill-vgpr-to-agpr=0", "-Wno-inline-asm"]}
# e.g. SGP_HIPCC_FLAGS=-DSGP_INSTRUMENT for scripts/ablate.py (use --force)
USER = os.environ.get("SGP_HIPCC_FLAGS", "").split()


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, scan=False):
    hipcc = _hipcc()
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + HEADERS):
            jobs.append([hipcc] + BASE + EXTRA.get(src, []) + USER + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=6) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(OUT, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-ldl"])
    # The sweep kernels keep state in hand-assigned registers across inline-asm matrix
    # instructions the compiler does not look into: with flags other than the shipped
    # ones (SGP_HIPCC_FLAGS), or on request (--scan / SGP_BUILD_SCAN=1), the ISA of every
    # instance is scanned for hazards, stray AccVGPRs and spills before the library is
    # handed out (tests/test_abi.py runs the same scan over the shipped sources).
    if scan or USER or os.environ.get("SGP_BUILD_SCAN") == "1":
        import importlib.util
        spec = importlib.util.spec_from_file_location(
            "check_mfma_hazards", os.path.join(REPO, "scripts", "dev", "check_mfma_hazards.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        if mod.main(list(USER)):
            raise RuntimeError("ISA scan of the sweep kernels failed (scripts/dev/"
                               "check_mfma_hazards.py): the library must not be used")
    # test harness for the device-side merges of the N-rank step (tests/native): its own
    # translation unit around csrc/sets.hip, linked against the library for the rest
    hs = os.path.join(REPO, "tests", "native", "merge_check.hip")
    if os.path.exists(hs):
        hb = os.path.join(REPO, "tests", "native", "merge_check")
        if force or _stale(hb, [hs, os.path.join(CSRC, "sets.hip"), OUT] + HEADERS):
            run([hipcc] + BASE + EXTRA["sets.hip"] +
                [hs, "-o", hb, "-L", PKG, "-l:libsafeopt_hip.so", "-Wl,-rpath," + PKG])
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv,
                scan="--scan" in sys.argv))
