"""Builds libseaice_hip.so (gfx950) in-tree with hipcc."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libseaice_hip.so")
SOURCES = ["sih_core.hip", "evp_init.hip", "evp_step.hip", "evp_weak.hip", "ir_advect.hip", "upwind_advect.hip",
           "halo_exchange.hip", "evp_persist.hip"]


def hipcc():
    for p in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if p and (os.path.isabs(p) and os.path.exists(p) or not os.path.isabs(p)):
            return p
    return "hipcc"


def _headers():
    d = os.path.join(HERE, "csrc")
    return [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith(".h")] + [os.path.join(ROOT, "include", "seaice_hip.h")]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    d = os.path.join(HERE, "csrc")
    deps = [os.path.join(d, f) for f in os.listdir(d) if f.endswith(".hip")] + _headers()
    return any(os.path.getmtime(x) > t for x in deps if os.path.exists(x))


def build(force=False, verbose=True):
    extra = [s for s in os.listdir(os.path.join(HERE, "csrc")) if s.endswith(".hip") and s not in SOURCES]
    srcs = [os.path.join(HERE, "csrc", s) for s in SOURCES + sorted(extra)]
    if not force and not needs_build():
        return LIB
    objs = []
    flags = ["-O3", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-std=c++17",
             "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
             "-I", os.path.join(ROOT, "include"), "-I", os.path.join(HERE, "csrc")]
    if os.environ.get("SIH_DEBUG_ALLOC"):   # allocation checker (sih_internal.h); build with force=True to switch
        flags.append("-DSIH_DEBUG_ALLOC")
    os.makedirs(os.path.join(HERE, "csrc", "obj"), exist_ok=True)
    procs = []
    for s in srcs:
        o = os.path.join(HERE, "csrc", "obj", os.path.basename(s) + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(
                [os.path.getmtime(s)] + [os.path.getmtime(h) for h in _headers()]):
            cmd = [hipcc()] + flags + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append(subprocess.Popen(cmd))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed")
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
