"""Build the gfx950 shared library (hipcc) and the CPU wave simulator (g++), in-tree under lib/."""
import os, subprocess, sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib")
HIP_SO = os.path.join(LIB, "libflashfftconv_hip.so")
SIM_SO = os.path.join(LIB, "libffcsim.so")
HIP_SRCS = ["ffc_hip.hip", "ffc_k_conv.hip", "ffc_k_kfft.hip", "ffc_k_dkf.hip", "ffc_k_dk.hip", "ffc_k_big.hip", "ffc_conv1d.hip", "ffc_plan.cpp"]
SIM_SRCS = ["ffc_sim.cpp", "ffc_plan.cpp"]


def _stale(target, srcs):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "flashfftconv_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.isfile(d))


def build_hip(force=False, verbose=False):
    """Compile every translation unit for gfx950 in parallel, then link the shared library."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(LIB, exist_ok=True)
    obj_dir = os.path.join(LIB, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdr_time = max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith(".h"))
    hdr_time = max(hdr_time, os.path.getmtime(os.path.join(HERE, "..", "include", "flashfftconv_hip.h")))

    def compile_one(f):
        src = os.path.join(CSRC, f)
        obj = os.path.join(obj_dir, f + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time):
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fPIC", "-c", "-x", "hip", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            return obj, True
        return obj, False

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        res = list(ex.map(compile_one, HIP_SRCS))
    objs = [o for o, _ in res]
    if force or any(ch for _, ch in res) or not os.path.exists(HIP_SO):
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", HIP_SO] + objs)
    return HIP_SO


def build_sim(force=False):
    os.makedirs(LIB, exist_ok=True)
    srcs = [os.path.join(CSRC, f) for f in SIM_SRCS]
    if force or _stale(SIM_SO, srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", SIM_SO] + srcs)
    return SIM_SO


if __name__ == "__main__":
    force = "--force" in sys.argv
    print(build_hip(force, verbose=True))
    print(build_sim(force))
