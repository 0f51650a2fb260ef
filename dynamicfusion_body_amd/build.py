"""Build libdfusion_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m dynamicfusion_body_amd.build [--force]

The library is the product: there is no CPU fallback, `_lib.load()` raises if it is
missing.  Objects are rebuilt when a source or header is newer than the object.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(os.path.dirname(PKG), "include")
LIB = os.path.join(PKG, "libdfusion_hip.so")
OBJDIR = os.path.join(PKG, "build")

# -ffp-contract=off: the fp64 mask arithmetic must round operation by operation exactly
# like the reference's numpy expressions (no fused multiply-add).
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off",
               "-fno-fast-math", "-Wall", "-Wno-unused-function", "-I" + INCLUDE]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP library cannot be built")
    return exe


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, obj, extra):
    cmd = [hipcc()] + HIPCC_FLAGS + extra + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    return r.stderr


def build_library(force=False, verbose=False, extra_flags=()):
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = sources()
    hdr_m = _deps_mtime()
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(OBJDIR, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_m):
            jobs.append((s, o))
    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for warn in ex.map(lambda so: _compile(so[0], so[1], list(extra_flags)), jobs):
                if verbose and warn:
                    print(warn, file=sys.stderr)
    if jobs or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
