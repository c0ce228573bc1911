"""In-tree build of libgmx.so and the `gram` executable with hipcc for gfx950.

The shared object is kept next to the sources (gramtools_amd/lib/) so that it travels with the
repository snapshot to the GPU box; it is git-ignored.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
BIN_DIR = os.path.join(HERE, "bin")
LIB = os.path.join(LIB_DIR, "libgmx.so")
GRAM = os.path.join(BIN_DIR, "gram")

LIB_SOURCES = ["gmx_engine.hip", "gmx_ingest.hip", "gmx_multi.hip", "gmx_seedwalk.hip", "gmx_suffixsort.hip", "gmx_capi.cpp", "gmx_index.cpp", "gmx_infer.cpp", "gmx_stock.cpp"]
HEADERS = ["gmx_engine_search.h", "gmx_engine_cover_kernels.h", "gmx_engine_host.h", "gmx_types.h", "gmx_core.h", "gmx_cover.h", "gmx_dfs.h", "gmx_index.h", "gmx_internal.h", "gmx_engine_debug.h", "gmx_gzsource.h", "gmx_pargz.h", "gmx_crc32.h", "libgmx.map", "../../include/gmx.h"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(s) and os.path.getmtime(s) > t for s in sources)


def _link(target, extra_flags, tag, force, verbose):
    """One object per source under lib/obj (rebuilt when the source or a header is newer), then the shared object."""
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    if not force and not _stale(target, [os.path.join(CSRC, name) for name in LIB_SOURCES] + hdrs):
        return target  # (the shared object travels to the GPU box without its objects: nothing to do there)
    obj_dir = os.path.join(LIB_DIR, "obj" + tag)
    os.makedirs(obj_dir, exist_ok=True)
    objs, jobs = [], []
    for name in LIB_SOURCES:
        src = os.path.join(CSRC, name)
        obj = os.path.join(obj_dir, name + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [HIPCC] + FLAGS + extra_flags + ["-c", "-o", obj, src]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            jobs.append((subprocess.Popen(cmd), cmd))
    for proc, cmd in jobs:  # (the sources compile side by side)
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    if jobs or force or _stale(target, objs):
        # (libgmx.map: the library's replacement operator new / delete — gmx_capi.cpp, the allocation-failure test hook — stay local)
        cmd = [HIPCC] + FLAGS + ["-shared", "-o", target] + objs + ["-Wl,--version-script=" + os.path.join(CSRC, "libgmx.map"), "-lpthread", "-ldl", "-lz"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return target


def build_library(force=False, verbose=False):
    return _link(LIB, [], "", force, verbose)


LIB_ALT = os.path.join(LIB_DIR, "libgmx_alt.so")


def build_library_alt(force=False, verbose=False):
    """Test build with -DGMX_SEARCHOUT_ALT: SearchOut in the member order that made the round-2 compiler emit a wrong
    gmx_probe_kernel (HISTORY.md §4.5). tests/test_searchout_layout.py runs the probe pipeline with both builds."""
    return _link(LIB_ALT, ["-DGMX_SEARCHOUT_ALT"], "_alt", force, verbose)


LIB_STATS = os.path.join(LIB_DIR, "libgmx_stats.so")


def build_library_stats(force=False, verbose=False):
    """Debug build with -DGMX_LOOP_STATS (iteration mix and phase clocks of the wave loop and the coverage instances;
    tools/loop_stats.py, tools/coop_stats_c2.py load it through GMX_LIB). Not built by __graft_entry__.build()."""
    return _link(LIB_STATS, ["-DGMX_LOOP_STATS"], "_stats", force, verbose)


def build_gram(force=False, verbose=False):
    src = os.path.join(CSRC, "gram_main.cpp")
    if not os.path.exists(src):
        return None
    build_library(force, verbose)
    deps = [src, LIB] + [os.path.join(CSRC, h) for h in HEADERS]
    if force or _stale(GRAM, deps):
        os.makedirs(BIN_DIR, exist_ok=True)
        cmd = [HIPCC] + FLAGS + ["-o", GRAM, src, "-L" + LIB_DIR, "-lgmx", "-Wl,-rpath,$ORIGIN/../lib",
                                 "-Wl,-rpath,/opt/rocm/lib", "-lz", "-lpthread"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return GRAM


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
    print(build_library_alt(force="--force" in sys.argv, verbose=True))
    g = build_gram(force="--force" in sys.argv, verbose=True)
    if g:
        print(g)
