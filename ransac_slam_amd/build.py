"""Builds librslam_hip.so (HIP kernels + C ABI) in-tree with hipcc for gfx950."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librslam_hip.so")
SOURCES = ["kernels.hip", "map_kernels.hip", "match_kernels.hip", "rslam_api.hip"]
HEADERS = ["kernels.h", "tile_gemm.h", "camera_model.h", os.path.join("..", "..", "include", "rslam.h")]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function"] + [os.path.join(CSRC, f) for f in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


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
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_host_example())
    print(build_shard_example())
