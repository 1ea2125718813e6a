"""Build the gfx950 shared library (csrc/*.hip -> csrc/libpit_hip.so) with hipcc.

In-tree so the .so travels with the repo snapshot to the GPU box.  hipcc cross-compiles
without a GPU.  `-ffp-contract=off`: the mask decision must reproduce the reference's
un-fused fp32 arithmetic; fmas are written explicitly where wanted.
"""
from __future__ import annotations

import os
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
# PIT_LIB_OUT: diagnostic builds (PIT_EXTRA_FLAGS=-DPIT_STAMPS ...) go to their OWN library and object directory, so they
# can never be mistaken for the production library; load them with PIT_LIB_PATH
LIB = os.environ.get("PIT_LIB_OUT") or os.path.join(CSRC, "libpit_hip.so")
SOURCES = ("pit_abi.hip", "pit_select.hip", "pit_posatt.hip", "pit_block.hip", "pit_edge.hip", "pit_fold.hip", "pit_chain.hip", "pit_satt.hip", "pit_mlp.hip", "pit_mlp_slab.hip", "pit_loss.hip", "pit_norm.hip", "pit_optim.hip")
HEADERS = ("pit_common.h", "pit_gemm_rd.h", "pit_block_dev.h", os.path.join("..", "..", "include", "pit_hip.h"))
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17"]
FLAGS += os.environ.get("PIT_EXTRA_FLAGS", "").split()      # diagnostic builds only (e.g. -DPIT_STAMPS, tools/stamp_tiles.py)


OBJDIR = (LIB + ".obj") if os.environ.get("PIT_LIB_OUT") else os.path.join(CSRC, "_obj")
STAMP = os.path.join(OBJDIR, "flags.stamp")


def _stale() -> bool:
    """Missing, older than a source / header, or built with OTHER flags (a -DPIT_STAMPS diagnostic build left in the
    tree must not be taken for the production library: the flags of the last build are kept in _obj/flags.stamp)."""
    if not os.path.exists(LIB):
        return True
    try:
        with open(STAMP) as f:
            if f.read() != " ".join(FLAGS):
                return True
    except OSError:
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile if missing or out of date; returns the library path.  The translation units are
    compiled in parallel (objects under csrc/_obj, not tracked) and linked into one library."""
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if force or _stale():
        from concurrent.futures import ThreadPoolExecutor
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        objdir = OBJDIR
        os.makedirs(objdir, exist_ok=True)
        cflags = [f for f in FLAGS if f != "-shared"]

        flag_text = " ".join(FLAGS)
        # The stamp describes the LIBRARY: it goes away before the first compile and comes back after a successful link, so an
        # interrupted or failed build with other flags (a -DPIT_EDGE_DBG / -DPIT_GELU_FAST diagnostic build) can never leave a
        # stamp that vouches for objects it did not produce.  What vouches for an OBJECT is the flag text kept next to it
        # (obj + ".flags", written after its compile succeeded).
        try:
            os.remove(STAMP)
        except OSError:
            pass
        hdr_time = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS if os.path.exists(os.path.join(CSRC, h)))
        # objects of sources that left SOURCES (a pit_latent.o of round 4) must not be found by tools that link _obj/*.o
        wanted = {s_.replace(".hip", ".o") for s_ in srcs}
        for name in os.listdir(objdir):
            if name.endswith(".o") and name not in wanted:
                for victim in (name, name + ".flags"):
                    try:
                        os.remove(os.path.join(objdir, victim))
                    except OSError:
                        pass

        def obj_flags(obj: str):
            try:
                with open(obj + ".flags") as f:
                    return f.read()
            except OSError:
                return None

        def compile_one(src: str) -> str:
            obj = os.path.join(objdir, src.replace(".hip", ".o"))
            # an object newer than its source and every header, built with THESE flags, is reused (a change to one
            # translation unit recompiles that unit only)
            if not force and obj_flags(obj) == flag_text and os.path.exists(obj) and \
                    os.path.getmtime(obj) > max(hdr_time, os.path.getmtime(os.path.join(CSRC, src))):
                return obj
            try:
                os.remove(obj + ".flags")
            except OSError:
                pass
            cmd = [hipcc] + cflags + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True, cwd=CSRC)
            with open(obj + ".flags", "w") as f:
                f.write(flag_text)
            return obj

        with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 1)) as pool:
            objs = list(pool.map(compile_one, srcs))
        cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True, cwd=CSRC)
        with open(STAMP, "w") as f:
            f.write(flag_text)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
