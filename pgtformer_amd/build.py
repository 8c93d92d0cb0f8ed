"""Build libpgt_hip.so (gfx950) in-tree with hipcc: `python -m pgtformer_amd.build`.

One object per translation unit under csrc/, rebuilt only when the source (or a header) is newer;
the shared library lands in pgtformer_amd/lib/ so it travels to the GPU box with the repo snapshot.
"""
import os
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


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    hdr_m = max(os.path.getmtime(os.path.join(d, f)) for d in (CSRC, INCLUDE) for f in os.listdir(d)
                if f.endswith((".h", ".inc")))
    objs, todo = [], []
    for src in sources():
        sp = os.path.join(CSRC, src)
        op = os.path.join(objdir, src.rsplit(".", 1)[0] + ".o")
        objs.append(op)
        if force or not os.path.exists(op) or os.path.getmtime(op) < max(os.path.getmtime(sp), hdr_m):
            todo.append([_hipcc()] + FLAGS + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", sp, "-o", op])
    rebuilt = bool(todo)
    if todo:   # translation units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print("[pgt build]", " ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=min(len(todo), max(1, (os.cpu_count() or 2) // 2))) as ex:
            list(ex.map(run, todo))
    if rebuilt or not os.path.exists(LIB):
        cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print("[pgt build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
