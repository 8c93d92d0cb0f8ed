"""Build libpgt_hip.so (gfx950) in-tree with hipcc: `python -m pgtformer_amd.build`.

One object per translation unit under csrc/, rebuilt when the CONTENT of its source or of any header changed (a sha256 next to
every object; file times play no part); the shared library lands in pgtformer_amd/lib/ so it travels to the GPU box with the
repo snapshot.  The sha256 over all sources is compiled into the library (pgt_version() = "pgt_hip <ver> (gfx950) src:<sha16>"):
measurement files quote the stamp of the BINARY that ran (binary_sha16), not of whatever sources lie next to it.
"""
import hashlib
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libpgt_hip.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-I", INCLUDE, "-I", CSRC,
         "-Wno-unused-result", "-ffp-contract=on"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _source_files():
    return [os.path.join(d, f) for d in (CSRC, INCLUDE) for f in sorted(os.listdir(d)) if f.endswith((".hip", ".cpp", ".h", ".inc"))]


def source_sha16():
    """sha256 (first 16 hex digits) over the sources libpgt_hip.so is built from: csrc/*.{hip,cpp,h,inc} and include/*.h"""
    h = hashlib.sha256()
    for path in _source_files():
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


_STAMP = re.compile(rb"pgt_hip [0-9.]+ \(gfx950\) src:([0-9a-f]{16})")


def binary_sha16(lib=None):
    """the source sha compiled into a built libpgt_hip.so (read from the file: no dlopen), or None if there is no stamped library"""
    try:
        m = _STAMP.search(open(lib or LIB, "rb").read())
    except OSError:
        return None
    return m.group(1).decode() if m else None


def build(force=False, verbose=True):
    """compile what changed and link; concurrent callers (the ranks of one node) take turns behind a file lock"""
    import fcntl

    os.makedirs(LIBDIR, exist_ok=True)
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose):
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    hdr = hashlib.sha256()
    for path in _source_files():
        if path.endswith((".h", ".inc")):
            hdr.update(os.path.basename(path).encode())
            hdr.update(open(path, "rb").read())
    sha = source_sha16()
    objs, todo, keys = [], [], {}
    for src in sources():
        sp = os.path.join(CSRC, src)
        op = os.path.join(objdir, src.rsplit(".", 1)[0] + ".o")
        objs.append(op)
        extra = ['-DPGT_SOURCE_SHA16="%s"' % sha] if src == "capi.cpp" else []      # the stamp: capi.o follows EVERY source
        key = hashlib.sha256(hdr.digest() + open(sp, "rb").read() + " ".join(FLAGS + extra).encode()).hexdigest()
        try:
            have = open(op + ".sha").read().strip()
        except OSError:
            have = ""
        if force or not os.path.exists(op) or have != key:
            keys[op] = key
            todo.append([_hipcc()] + FLAGS + extra + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", sp, "-o", op])
    rebuilt = bool(todo)
    if todo:   # translation units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print("[pgt build]", " ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            with open(cmd[-1] + ".sha", "w") as f:
                f.write(keys[cmd[-1]])
        with ThreadPoolExecutor(max_workers=min(len(todo), max(1, (os.cpu_count() or 2) // 2))) as ex:
            list(ex.map(run, todo))
    if rebuilt or not os.path.exists(LIB) or binary_sha16() != sha:
        cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print("[pgt build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
