"""Build libasd_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python adaptive-speculative-decoding_amd/build.py [--force] [--verbose]

Output: adaptive-speculative-decoding_amd/lib/libasd_hip.so (git-ignored, travels with gpurun).
decision.hip and predictor.hip are compiled with -ffp-contract=off: they restate CPython /
numpy float64 arithmetic and must round once per operator (see decision_device.hpp).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INC = os.path.join(os.path.dirname(HERE), "include")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libasd_hip.so")

ARCH = "gfx950"
COMMON = ["-std=c++17", "-O3", f"--offload-arch={ARCH}", "-fPIC", "-fvisibility=hidden",
          "-fno-fast-math", "-Wall", "-Wextra", "-Wno-unused-parameter", f"-I{INC}", f"-I{CSRC}"]
SOURCES = {
    "api.hip": [],
    # hosts the in-kernel epilogue (predictor_device.hpp); the streaming prologue's arguments are preloaded into SGPRs
    "verify_accept.hip": ["-ffp-contract=off", "-mllvm", "-amdgpu-kernarg-preload-count=11"],
    "residual_sample.hip": [],
    "draft_sample.hip": [],
    "decoder.hip": [],
    "commit.hip": [],
    "lm_head_verify.hip": ["-ffp-contract=off"],  # ends in the same finish_row arithmetic as verify_accept.hip
    "decision.hip": ["-ffp-contract=off"],
    "predictor.hip": ["-ffp-contract=off", "-mllvm", "-amdgpu-kernarg-preload-count=11"],   # k_predictor_stop_w64x32's leading arguments
}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libasd_hip.so cannot be built (ROCm toolchain missing)")


def _stale(out: str, deps) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


# The TEST build of the same sources: -DASD_TEST_HOOKS adds the asd_debug_* switches (process-global test / fault-injection hooks,
# include/asd_hip.h) that the product library does not contain.  Only the translation units that host a hook are compiled a
# second time; the others are linked from the product objects.
TEST_LIB = os.path.join(LIBDIR, "libasd_hip_test.so")
TEST_HOOK_SOURCES = ("verify_accept.hip", "residual_sample.hip", "draft_sample.hip", "lm_head_verify.hip")

ASAN_LIB = os.path.join(LIBDIR, "libasd_hip_asan.so")
ASAN_FLAGS = ["-Xarch_host", "-fsanitize=address,undefined", "-Xarch_host", "-fno-omit-frame-pointer", "-g"]


def build(force: bool = False, verbose: bool = False, asan: bool = False) -> str:
    """asan=True: a SECOND library, lib/libasd_hip_asan.so, whose HOST code (argument checks, geometry heuristics, workspace
    layout, launch plumbing) is compiled with AddressSanitizer + UBSan (`-Xarch_host`: the device code is untouched -- GPU
    sanitizers are not available on this pool).  `make asan-host` runs the CPU-side ABI tests against it."""
    hipcc = _hipcc()
    global OBJDIR, LIB
    objdir, lib = (os.path.join(LIBDIR, "obj_asan"), ASAN_LIB) if asan else (OBJDIR, LIB)
    return _build(hipcc, objdir, lib, ASAN_FLAGS if asan else [], force, verbose)


def build_test_hooks(force: bool = False, verbose: bool = False) -> str:
    """lib/libasd_hip_test.so: the product objects, with the hook-hosting translation units recompiled under -DASD_TEST_HOOKS."""
    build(force=force, verbose=verbose)            # the product objects the test library links
    return _build(_hipcc(), os.path.join(LIBDIR, "obj_test"), TEST_LIB, ["-DASD_TEST_HOOKS"], force, verbose,
                  only=TEST_HOOK_SOURCES, fallback_objdir=OBJDIR)


def _build(hipcc, OBJDIR, LIB, more, force, verbose, only=None, fallback_objdir=None) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers += [os.path.join(INC, "asd_hip.h"), os.path.abspath(__file__)]
    objs, jobs = [], []
    for src, extra in SOURCES.items():
        s = os.path.join(CSRC, src)
        if only is not None and src not in only:   # linked from another build's objects
            objs.append(os.path.join(fallback_objdir, src.replace(".hip", ".o")))
            continue
        o = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc, *COMMON, *extra, *more, "-c", s, "-o", o])
    if jobs:                                   # one hipcc per translation unit, side by side (each is single-threaded)
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=min(len(jobs), max(1, (os.cpu_count() or 2) - 1))) as pool:
            list(pool.map(run, jobs))
    if force or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *([a for a in more if a not in ("-Xarch_host", "-g")]), "-o", LIB, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    if "--test-hooks" in sys.argv:
        print(build_test_hooks(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv))
    else:
        print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv, asan="--asan" in sys.argv))
