"""Builds the C-ABI library (HIP kernels + host logic) in-tree with hipcc for gfx950.

Two variants of the same sources:
  librslam_hip.so      the product: exports exactly what include/rslam.h declares
  librslam_hip_dbg.so  -DRSLAM_DEBUG: the same kernels plus the diagnostic / fault-injection entry points
                       (rslam_debug_*), the time-stamp buffers and the RSLAM_* measurement switches read from the
                       environment.  Tests and scripts that inject faults or read stamps load this one; nothing of it
                       is reachable from the product library.
Every source is compiled to an object of its own (in parallel: kernels.hip takes ~2 minutes, the rest seconds), so a
change of the host logic does not recompile the kernels.
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "librslam_hip.so")
LIB_DEBUG = os.path.join(HERE, "librslam_hip_dbg.so")
SOURCES = ["kernels.hip", "staged_kernels.hip", "rank_macro.hip", "map_kernels.hip", "match_kernels.hip", "rslam_api.hip"]
HEADERS = ["kernels.h", "tile_gemm.h", "tile_gemm128.h", "rank_common.h", "camera_model.h", os.path.join("..", "..", "include", "rslam.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc():
    return os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _newest_header():
    return max(os.path.getmtime(os.path.join(CSRC, f)) for f in HEADERS)


def _obj_path(src, debug):
    return os.path.join(OBJ, ("dbg_" if debug else "rel_") + os.path.splitext(src)[0] + ".o")


def _obj_stale(src, debug):
    o = _obj_path(src, debug)
    if not os.path.exists(o):
        return True
    t = os.path.getmtime(o)
    return os.path.getmtime(os.path.join(CSRC, src)) > t or _newest_header() > t


def _compile(job):
    src, debug, extra = job
    cmd = [_hipcc()] + FLAGS + (["-DRSLAM_DEBUG"] if debug else []) + list(extra) + \
          ["-c", os.path.join(CSRC, src), "-o", _obj_path(src, debug)]
    subprocess.check_call(cmd)
    return " ".join(cmd)


def _lib_stale(debug):
    """the library is missing or older than a source / header (objects do not matter: a copy of the tree that carries the
    built libraries but not the object directory -- the GPU boxes -- must not recompile)"""
    lib = LIB_DEBUG if debug else LIB
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False, debug=None):
    """debug: False = the product library, True = the diagnostic variant, None = both.  Returns the product library's path
    (the diagnostic one's when debug is True).  Takes no extra compiler flags on purpose: an instrumented or ablated build
    written over librslam_hip.so would look fresh to the next plain build() (staleness is judged by time stamps) and stay
    the product library -- experiments go through build_dev (a library of their own under _dev/)."""
    variants = [v for v in ([False, True] if debug is None else [bool(debug)]) if force or _lib_stale(v)]
    if variants:
        os.makedirs(OBJ, exist_ok=True)
        jobs = [(src, dbg, ()) for dbg in variants for src in SOURCES if force or _obj_stale(src, dbg)]
        if jobs:
            with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(jobs), max(1, (os.cpu_count() or 2) // 2))) as ex:
                for line in ex.map(_compile, jobs):
                    if verbose:
                        print(line)
        for dbg in variants:
            lib = LIB_DEBUG if dbg else LIB
            cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + [_obj_path(src, dbg) for src in SOURCES] + ["-o", lib]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
    return LIB_DEBUG if debug else LIB


DEV_DIR = os.path.join(HERE, "_dev")
ASM = os.path.join(OBJ, "rel_kernels.s")
FENCED = os.path.join(DEV_DIR, "fenced.so")
FENCED_FLAGS = ["-DCD_HW_FENCES", "-DCDP_DMA_BUILTIN"]


def _older_than_sources(path):
    if not os.path.exists(path):
        return True
    t = os.path.getmtime(path)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build_asm():
    """Device assembly of the product build of kernels.hip (same flags), for scripts/check_pivot_waitcnt.py: an invariant
    of the pivot pipeline that only the instruction stream shows (no compiler-inserted wait for global memory in front of
    the LDS reads of the pivot steps behind the hand-written LDS-DMA)."""
    if _older_than_sources(ASM):
        os.makedirs(OBJ, exist_ok=True)
        subprocess.check_call([_hipcc()] + FLAGS + ["--cuda-device-only", "-S", os.path.join(CSRC, "kernels.hip"), "-o", ASM])
    return ASM


def build_fenced():
    """The conservative twin of the diagnostic library, ransac_slam_amd/_dev/fenced.so: every shape, with the hardware
    fences of the LDS hand-over protocol (-DCD_HW_FENCES) and the in-chain LDS-DMA as the compiler's builtin
    (-DCDP_DMA_BUILTIN, i.e. with the waits the compiler derives for it).  tests/test_gpu_invariants.py requires the product
    library's results to be bit-identical to this one's: the two hand-placed relaxations are then not what the results rest on."""
    if _older_than_sources(FENCED):
        build_dev("fenced", FENCED_FLAGS, full=True)
    return FENCED



def build_dev(name, extra_flags=(), full=False):
    """A kernel experiment as a library of its own, ransac_slam_amd/_dev/<name>.so: the diagnostic variant (-DRSLAM_DEBUG:
    stamps, switches) with `extra_flags`, by default with the C3 shape of the persistent sweep only (-DRSLAM_DEV_ONLY_NJ12,
    a third of the compile time).  For A/B measurements inside ONE gpurun call: RSLAM_HIP_LIB_DEBUG=<path> scripts/ab_frame.py --debug."""
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(DEV_DIR, exist_ok=True)
    flags = ["-DRSLAM_DEBUG"] + ([] if full else ["-DRSLAM_DEV_ONLY_NJ12"]) + list(extra_flags)
    objs = []
    jobs = []
    for src in SOURCES:
        if src in ("kernels.hip", "rslam_api.hip", "rank_macro.hip", "staged_kernels.hip"):
            o = os.path.join(OBJ, "dev_%s_%s.o" % (name, os.path.splitext(src)[0]))
            jobs.append([_hipcc()] + FLAGS + flags + ["-c", os.path.join(CSRC, src), "-o", o])
        else:
            if _obj_stale(src, True):
                _compile((src, True, ()))
            o = _obj_path(src, True)
        objs.append(o)
    with concurrent.futures.ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(subprocess.check_call, jobs))
    lib = os.path.join(DEV_DIR, name + ".so")
    subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib])
    return lib


HOST_DIR = os.path.join(HERE, "host")
HOST_EXAMPLE = os.path.join(HOST_DIR, "track_frame_example")


def build_host_example(force=False):
    """C++ host-side mirror of the reference interface (ransac_slam_hip.hpp) + example driver."""
    srcs = [os.path.join(HOST_DIR, "track_frame_example.cpp"), os.path.join(HOST_DIR, "ransac_slam_hip.hpp")]
    if (not force and os.path.exists(HOST_EXAMPLE)
            and os.path.getmtime(HOST_EXAMPLE) >= max(os.path.getmtime(f) for f in srcs + [LIB])):
        return HOST_EXAMPLE
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.dirname(HERE), srcs[0], "-o", HOST_EXAMPLE,
                           "-L", HERE, "-lrslam_hip", "-Wl,-rpath,$ORIGIN/.."])
    return HOST_EXAMPLE


SHARD_EXAMPLE = os.path.join(HOST_DIR, "shard_frame_example")


def build_shard_example(force=False):
    """C++ driver of the hypothesis-sharded frame: one rank, RCCL communicator owned by the caller (links librccl)."""
    srcs = [os.path.join(HOST_DIR, "shard_frame_example.cpp"), os.path.join(HOST_DIR, "ransac_slam_hip.hpp")]
    if (not force and os.path.exists(SHARD_EXAMPLE)
            and os.path.getmtime(SHARD_EXAMPLE) >= max(os.path.getmtime(f) for f in srcs + [LIB])):
        return SHARD_EXAMPLE
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", os.path.dirname(HERE),
                           "-I", os.path.join(rocm, "include"), srcs[0], "-o", SHARD_EXAMPLE,
                           "-L", HERE, "-lrslam_hip", "-L", os.path.join(rocm, "lib"), "-lrccl", "-lamdhip64",
                           "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath," + os.path.join(rocm, "lib")])
    return SHARD_EXAMPLE


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "dev":          # python -m ransac_slam_amd.build dev <name> [-DFLAG ...]
        print(build_dev(sys.argv[2], [a for a in sys.argv[3:] if a != "--full"], full="--full" in sys.argv))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_host_example())
    print(build_shard_example())
