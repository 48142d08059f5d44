"""Build the gfx950 shared library (hipcc) and the CPU wave simulator (g++), in-tree under lib/."""
import os, subprocess, sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib")
HIP_SO = os.path.join(LIB, "libflashfftconv_hip.so")
SIM_SO = os.path.join(LIB, "libffcsim.so")
HIP_SRCS = ["ffc_hip.hip", "ffc_conv1d.hip", "ffc_plan.cpp"]
SIM_SRCS = ["ffc_sim.cpp", "ffc_plan.cpp"]


def _stale(target, srcs):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "flashfftconv_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.isfile(d))


def build_hip(force=False, verbose=False):
    os.makedirs(LIB, exist_ok=True)
    srcs = [os.path.join(CSRC, f) for f in HIP_SRCS if os.path.exists(os.path.join(CSRC, f))]
    if force or _stale(HIP_SO, srcs):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", HIP_SO] + srcs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
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
