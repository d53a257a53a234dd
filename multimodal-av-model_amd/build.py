"""Build libavhip.so for gfx950 with hipcc (cross-compiles without a GPU).  In-tree artefact, git-ignored."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libavhip.so")
LIB_F16 = os.path.join(HERE, "libavhip_f16.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value"]


def _newer(src: str, dst: str) -> bool:
    return (not os.path.exists(dst)) or os.path.getmtime(src) > os.path.getmtime(dst)


def _build_one(lib: str, objdir: str, extra: list, force: bool, verbose: bool) -> str:
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "av_hip.h")]
    hdr_m = max(os.path.getmtime(h) for h in hdrs)
    os.makedirs(objdir, exist_ok=True)
    objs, jobs = [], []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s[:-4] + ".o")
        objs.append(obj)
        if force or _newer(src, obj) or os.path.getmtime(obj) < hdr_m:
            jobs.append(["hipcc", *FLAGS, *extra, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=6) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(lib) or any(os.path.getmtime(o) > os.path.getmtime(lib) for o in objs):
        run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
    return lib


def build(force: bool = False, verbose: bool = True) -> str:
    """libavhip.so (bfloat16 operands, the default perf mode) and libavhip_f16.so (the same sources with IEEE-half operands, -DAV_HALF=1:
    the reference's fp16-autocast arithmetic, precision mode "fp16").  Returns the path of the default library."""
    _build_one(LIB_F16, os.path.join(HERE, "build_f16"), ["-DAV_HALF=1"], force, verbose)
    return _build_one(LIB, os.path.join(HERE, "build"), [], force, verbose)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
