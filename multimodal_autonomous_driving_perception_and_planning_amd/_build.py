"""Builds libavhot.so (hipcc, gfx950 only) in-tree.  hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "obj")
LIB = os.path.join(HERE, "libavhot.so")

ARCH = "gfx950"
CXXFLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall",
            "-Wno-unused-function", "-I", os.path.join(ROOT, "include"), "-I", CSRC]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found; libavhot.so cannot be built")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime(src, seen=None):
    """Newest modification time of `src` and of everything it includes from csrc/ or include/ (step.hip includes the stage
    files themselves, not only headers)."""
    import re
    seen = set() if seen is None else seen
    if src in seen or not os.path.exists(src):
        return 0.0
    seen.add(src)
    m = os.path.getmtime(src)
    for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(src).read(), flags=re.M):
        for d in (CSRC, os.path.join(ROOT, "include")):
            m = max(m, _deps_mtime(os.path.join(d, inc), seen))
    return m


def build(force=False, verbose=True, jobs=None):
    """Compile every csrc/*.hip for gfx950 and link libavhot.so.  Returns the library path."""
    cc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    todo, objs = [], []
    for src in sources():
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < _deps_mtime(src):
            todo.append((src, obj))

    def compile_one(so):
        src, obj = so
        cmd = [cc] + CXXFLAGS + ["-c", src, "-o", obj]
        if verbose:
            print("[avhot] hipcc", os.path.basename(src), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    if todo:
        with ThreadPoolExecutor(max_workers=jobs or min(6, len(todo))) as ex:
            list(ex.map(compile_one, todo))
    if todo or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [cc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print("[avhot] link", os.path.basename(LIB), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
