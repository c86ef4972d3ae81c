"""Builds libseekstorm_hip.so (gfx950) in-tree with hipcc.  No CPU fallback is ever built."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.environ.get("SS_OUT_DIR") or os.path.join(HERE, "lib")  # SS_OUT_DIR: experiment builds beside the product build
LIB = os.path.join(OUT_DIR, "libseekstorm_hip.so")
SOURCES = ["ss_api.hip", "vec_scan.hip", "bm25.hip", "synth.hip", "merge.hip", "bm25_fast.hip", "bm25_probe.hip", "ref_format.hip", "vec8_scan.hip", "vec_ann.hip", "facet.hip", "comm.hip", "bm25_phrase.hip", "bm25_scan16.hip", "bm25_sparse.hip", "bm25_small.hip", "bm25_gallop.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result"] + os.environ.get("SS_HIPCC_FLAGS", "").split()


def _newer(dst, srcs):
    if not os.path.exists(dst):
        return False
    t = os.path.getmtime(dst)
    return all(os.path.getmtime(s) <= t for s in srcs)


def build(force=False, verbose=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    # every header of csrc/ (they are few and included widely) + the public header: editing any of them rebuilds every object
    deps = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join(HERE, "..", "include", "seekstorm_hip.h")]
    objs = []
    jobs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OUT_DIR, src.replace(".hip", ".o"))
        objs.append(op)
        if force or not _newer(op, [sp] + deps):
            jobs.append([HIPCC] + FLAGS + ["-c", sp, "-o", op])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r.stderr

    with ThreadPoolExecutor(max_workers=6) as ex:
        for err in ex.map(run, jobs):
            if verbose and err:
                print(err)
    if jobs or not os.path.exists(LIB):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-o", LIB] + objs + ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"])
    build_host(force or bool(jobs), verbose)
    return LIB


HOST_DIR = os.path.join(HERE, "host")
HOST_LIB = os.path.join(OUT_DIR, "libseekstorm_host.so")
HOST_SOURCES = ["seekstorm_host.cpp", "host_capi.cpp"]


def build_host(force=False, verbose=False):
    """C++ host mirror of the reference's search interface (planner, merge, batch coalescer) above the C ABI."""
    srcs = [os.path.join(HOST_DIR, f) for f in HOST_SOURCES]
    deps = srcs + [os.path.join(HOST_DIR, "seekstorm_host.hpp"), os.path.join(HERE, "..", "include", "seekstorm_hip.h")]
    if not force and _newer(HOST_LIB, deps):
        return HOST_LIB
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-pthread", "-o", HOST_LIB] + srcs + \
          ["-L" + OUT_DIR, "-lseekstorm_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("g++ failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
    return HOST_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
