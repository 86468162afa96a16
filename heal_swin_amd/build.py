"""Build libhealswin.so (HIP kernels + C ABI, include/healswin.h) for gfx950 with hipcc.

In-tree build: objects under heal_swin_amd/build/, the shared library at heal_swin_amd/lib/libhealswin.so
(git-ignored, but it travels to the GPU box with the repo snapshot).  hipcc cross-compiles gfx950 code
objects without a GPU present.
"""
import concurrent.futures
import hashlib
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
BUILD_DIR = os.path.join(PKG_DIR, "build")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libhealswin.so")

ARCH = "gfx950"
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function"]
CXXFLAGS += os.environ.get("HS_EXTRA_CXXFLAGS", "").split()  # e.g. -DHS_ATTN_ABLATION for tools/attn_bwd_ablation.py


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (expected /opt/rocm/bin/hipcc)")


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _stamp():
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)) + ["../../include/healswin.h"]:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    h.update(" ".join(CXXFLAGS + [ARCH]).encode())
    return h.hexdigest()


def _unit_stamp(src):
    """Hash of one translation unit: its source, every header it may include, the flags."""
    h = hashlib.sha256()
    for f in [src] + sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + ["../../include/healswin.h"]:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    h.update(" ".join(CXXFLAGS + [ARCH]).encode())
    return h.hexdigest()


def _compile(src):
    obj = os.path.join(BUILD_DIR, os.path.splitext(src)[0] + ".o")
    unit_file, unit = obj + ".stamp", _unit_stamp(src)
    if os.path.exists(obj) and os.path.exists(unit_file) and open(unit_file).read() == unit:
        return obj, ""  # unchanged since its object was made (incremental rebuilds while iterating on one kernel)
    cmd = [_hipcc(), f"--offload-arch={ARCH}", *CXXFLAGS, "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    with open(unit_file, "w") as fh:
        fh.write(unit)
    return obj, r.stderr


def build_library(force=False, verbose=True):
    """Compile every translation unit under csrc/ for gfx950 and link libhealswin.so.  Returns its path."""
    os.makedirs(BUILD_DIR, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    stamp_file = os.path.join(BUILD_DIR, "stamp")
    stamp = _stamp()
    if not force and os.path.exists(LIB_PATH) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        if verbose:
            print(f"[healswin build] up to date: {LIB_PATH}")
        return LIB_PATH
    srcs = _sources()
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        results = list(ex.map(_compile, srcs))
    for src, (_, err) in zip(srcs, results):
        if verbose and err.strip():
            print(f"[healswin build] {src}:\n{err}", file=sys.stderr)
    objs = [o for o, _ in results]
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH, *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    with open(stamp_file, "w") as fh:
        fh.write(stamp)
    if verbose:
        print(f"[healswin build] built {LIB_PATH} from {len(srcs)} sources")
    return LIB_PATH


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
