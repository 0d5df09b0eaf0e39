"""Build the engine's HIP sources into alphazero-rs_amd/libaz_engine.so for gfx950.

hipcc cross-compiles without a GPU.  -ffp-contract=off is REQUIRED: the tree
kernels reproduce the reference's f32 operation order (src/node.rs:343-357) and
an FMA contraction would change visit counts.
"""
import os
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB = os.path.join(PKG_DIR, "libaz_engine.so")
# The diagnostic twin: the same sources with -DAZ_DIAG, which adds the superseded kernel generations (csrc/az_net_diag.inc), forced
# tiles, clock-stamp builds and the timing ablations that compute WRONG results.  tools/ and the kernel-family bit-identity tests
# load it (engine.Engine(diag=True)); the shipped library above does not contain any of it and refuses those option values.
LIB_DIAG = os.path.join(PKG_DIR, "libaz_engine_diag.so")
SOURCES = ["az_tree.hip", "az_net.hip", "az_train.hip", "az_engine.hip"]
HEADERS = ["az_common.h", "az_game.h", "az_tree.h", "az_net.h", "az_train.h", "az_net_diag.inc", os.path.join("..", "..", "include", "az_engine.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall",
         "-Wno-unused-result"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_engine(force=False, verbose=True, diag=True):
    """Compile each .hip to an object (in parallel) and link the shared library; with diag also the diagnostic twin."""
    lib = _build(LIB, "", [], force, verbose)
    if diag:
        _build(LIB_DIAG, ".diag", ["-DAZ_DIAG"], force, verbose)
    return lib


def _build(LIB, suffix, extra_flags, force, verbose):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs, procs = [], []
    for src in SOURCES:
        if suffix and src == "az_train.hip":          # the trainer has no diagnostic code: one object serves both libraries
            objs.append(os.path.join(CSRC, "az_train.o"))
            continue
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", suffix + ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc] + FLAGS + extra_flags + ["-c", s, "-o", o]
            if verbose:
                print("[build]", " ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out)
            raise RuntimeError(f"hipcc failed on {src}")
        if verbose and out.strip():
            print(out)
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_engine(force="--force" in sys.argv))
