"""In-tree build of libchordvis.so (HIP kernels + C-ABI + host mirror) for gfx950.

hipcc cross-compiles without a GPU.  Parity kernels are built with
-ffp-contract=off (no FMA contraction) and HIP's default correctly rounded
fp32 divide/sqrt — the canonical arithmetic of SURVEY.md §8c.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "_build")
BUILD_DIR = os.path.join(HERE, "_build", "obj")
LIB = os.path.join(OUT_DIR, "libchordvis.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fvisibility=hidden: the dynamic symbols are the C entry points of include/chordvis.h and nothing else (the header pushes default
# visibility around its declarations; tests/test_abi.py compares `nm -D` with the header)
COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
          "-I", os.path.join(ROOT, "include"), "-I", CSRC]
# HIP defaults kept on purpose: -fhip-fp32-correctly-rounded-divide-sqrt (IEEE / and sqrt), denormals preserved.
DEVICE = ["--offload-arch=gfx950"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _headers_digest():
    h = hashlib.sha1()
    for d in (CSRC, os.path.join(ROOT, "include")):
        for f in sorted(os.listdir(d)):
            if f.endswith((".h", ".hpp", ".hip.h", ".map")):
                with open(os.path.join(d, f), "rb") as fh:
                    h.update(fh.read())
    h.update(" ".join(COMMON + DEVICE).encode())
    return h.hexdigest()


def build(force=False, verbose=True, defines=(), tag=""):
    """defines / tag: a measurement variant (e.g. defines=("-DCHORD_TILE_V=1",), tag="tile_v1") is built next to the
    product library as _build/libchordvis_<tag>.so; chord_amd/lib.py loads it when CHORDVIS_LIB names it."""
    # (object trees of measurement variants whose library has been deleted go with it: they only cost pushes to the GPU box)
    if os.path.isdir(OUT_DIR):
        import shutil
        for d in os.listdir(OUT_DIR):
            if d.startswith("obj_") and not os.path.exists(os.path.join(OUT_DIR, "libchordvis_%s.so" % d[4:])) and d[4:] != tag:
                shutil.rmtree(os.path.join(OUT_DIR, d), ignore_errors=True)
    build_dir = BUILD_DIR + ("_" + tag if tag else "")
    lib_path = os.path.join(OUT_DIR, "libchordvis%s.so" % ("_" + tag if tag else ""))
    hd = _headers_digest() + " ".join(defines)
    # the library carries a digest of everything it was made from: a tree that holds the library but not its object files (the GPU
    # box: .gpurunignore leaves chord_amd/_build/obj*/ at home) is up to date without compiling anything
    whole = hashlib.sha1(hd.encode())
    for src in _sources():
        with open(os.path.join(CSRC, src), "rb") as fh:
            whole.update(fh.read())
    whole = whole.hexdigest()
    digest_path = lib_path + ".digest"
    if not force and os.path.exists(lib_path) and os.path.exists(digest_path) and open(digest_path).read() == whole:
        return lib_path
    os.makedirs(build_dir, exist_ok=True)
    objs, rebuilt = [], False
    for src in _sources():
        path = os.path.join(CSRC, src)
        obj = os.path.join(build_dir, src + ".o")
        stamp = obj + ".stamp"
        with open(path, "rb") as fh:
            digest = hashlib.sha1(fh.read()).hexdigest() + hd
        old = open(stamp).read() if os.path.exists(stamp) else ""
        if force or old != digest or not os.path.exists(obj):
            # hipcc compiles .cpp as HIP too: name the one target for every file (no default-arch code objects)
            cmd = [HIPCC] + COMMON + DEVICE + list(defines)
            if src.endswith(".hip"):
                cmd += ["-x", "hip"]
            cmd += ["-c", path, "-o", obj]
            if verbose:
                print("[chord_amd.build]", " ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            with open(stamp, "w") as fh:
                fh.write(digest)
            rebuilt = True
        objs.append(obj)
    if rebuilt or not os.path.exists(lib_path):
        # -no-hip-rt: libchordvis.so carries NO DT_NEEDED on a particular libamdhip64.  A process must hold
        # exactly one HIP/HSA runtime; PyTorch-ROCm wheels bundle their own (SONAME libamdhip64.so, not
        # libamdhip64.so.7), so the host decides which runtime is live and loads it first (chord_amd/lib.py;
        # a C++ host simply links -lamdhip64 itself).
        # the version script makes `chordvis_*` the only dynamic symbols: hipcc keeps the kernels' host-side handles and the implicit
        # members of types named in the header at default visibility whatever -fvisibility says
        cmd = [HIPCC, "-shared", "-fPIC", "-no-hip-rt", "-Wl,--version-script=" + os.path.join(CSRC, "chordvis.map"), "-o", lib_path] + objs + ["--offload-arch=gfx950"]
        if verbose:
            print("[chord_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with open(digest_path, "w") as fh:
        fh.write(whole)
    return lib_path


if __name__ == "__main__":
    # python chord_amd/build.py [--force] [--tag NAME -DX=1 ...]
    argv = sys.argv[1:]
    tag = argv[argv.index("--tag") + 1] if "--tag" in argv else ""
    print(build(force="--force" in argv, defines=tuple(a for a in argv if a.startswith("-D")), tag=tag))
