"""Build the gfx950 shared library (hipcc) and the CPU wave simulator (g++), in-tree under lib/."""
import os, subprocess, sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib")
HIP_SO = os.path.join(LIB, "libflashfftconv_hip.so")
SIM_SO = os.path.join(LIB, "libffcsim.so")
# -fno-slp-vectorize: packed fp32 math is written explicitly where it pays; auto-formed pairs cost v_mov shuffles.
# --amdgpu-mfma-vgpr-form: MFMA results stay in architectural VGPRs; the accumulation registers a0..a127 are
#   addressed by hand in the backward kernels (dk_f partial sums) and must never be picked by the allocator.
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-mllvm", "--amdgpu-mfma-vgpr-form", "-fPIC"]
AGPR_CHECKED = ["ffc_k_dkf.hip", "ffc_k_bwd.hip", "ffc_k_bwdz.hip"]     # translation units whose device code is scanned by check_agpr()
HIP_SRCS = ["ffc_hip.hip", "ffc_k_conv.hip", "ffc_k_kfft.hip", "ffc_k_dkf.hip", "ffc_k_bwd.hip", "ffc_k_bwdz.hip", "ffc_k_dk.hip", "ffc_k_big.hip", "ffc_conv1d.hip", "ffc_conv1d_t0.hip", "ffc_conv1d_t1.hip", "ffc_conv1d_t2.hip", "ffc_plan.cpp"]
SIM_SRCS = ["ffc_sim.cpp", "ffc_plan.cpp"]


def _stale(target, srcs):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "flashfftconv_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.isfile(d))


def check_agpr(asm_path):
    """The backward kernels keep state in a0..a127 through explicit v_accvgpr_read/write (csrc/ffc_dev.h,
    agpr_get/agpr_set).  Any other reference to an accumulation register means the compiler allocated one
    for its own use and would clobber that state: fail the build."""
    import re
    from collections import Counter
    mine = re.compile(r"^\s*v_accvgpr_(read_b32 v\d+, a(\d+)|write_b32 a(\d+), (?:v\d+|0))\s*$")
    # round 6: the sums are accumulated by MFMAs written in inline asm (DevB::mfma_acc_bf16): destination = C = one aligned 16-register tuple
    # of a0..a127, operands in architectural VGPRs
    mine_mfma = re.compile(r"^\s*v_mfma_f32_32x32x16_bf16 a\[(\d+):(\d+)\], v\[\d+:\d+\], v\[\d+:\d+\], a\[(\d+):(\d+)\]\s*$")
    areg = re.compile(r"\ba\[\d+:\d+\]|\ba\d+\b")
    kernel, reads, writes, mfmas, bad = None, Counter(), Counter(), Counter(), []

    def close():
        # every accumulator is zeroed + updated (2 writes) and read for the update + the final store (2 reads), in the kernels
        # with the dk tail (Modes::dk_tail) once more for it (3 reads); any other count is a compiler spill into the same register
        # the third read is only legitimate where the tail is instantiated: bwd_kernel<Geo<..>, ...> with more than one wave per
        # unit (everything but Geo<16,16,16> = fft 4096); in any other kernel a third read IS the spill this audit exists for,
        # and within one kernel all registers must agree
        tail_ok = kernel is not None and ((kernel.startswith("_Z10bwd_kernelIN3ffc3GeoILi")
                                           and not kernel.startswith("_Z10bwd_kernelIN3ffc3GeoILi16ELi16ELi16EEE"))
                                          or kernel.startswith("_Z13bwd_rp_kernelIN3ffc3GeoILi32ELi32ELi32EEELi0E"))   # bf16 multi-pass: dk_tail_rp
        counts = set()
        # MFMA-accumulated sums (round 6): zeroed (1 write), accumulated by the matrix pipe, read for the final store (1 read) and, in the
        # kernels with the tail, once more; VALU-accumulated sums (FFC_WACC_MFMA=0 builds): one more read and write per register for the update
        upd = 0 if mfmas else 1
        for idx in set(reads) | set(writes) | set(mfmas):
            counts.add(reads[idx])
            if reads[idx] not in ((1 + upd, 2 + upd) if tail_ok else (1 + upd,)) or writes[idx] != 1 + upd:
                bad.append(f"{kernel}: a{idx} read {reads[idx]}x written {writes[idx]}x")
            if mfmas and mfmas[idx] != max(mfmas.values()):
                bad.append(f"{kernel}: a{idx} accumulated by {mfmas[idx]} MFMAs, others by {max(mfmas.values())}")
        if len(counts) > 1:
            bad.append(f"{kernel}: accumulation registers read unevenly ({sorted(counts)} reads)")
        reads.clear(); writes.clear(); mfmas.clear()

    with open(asm_path) as fh:
        for line in fh:
            t = line.split(";")[0].rstrip()
            m = re.match(r"^(_Z\w+):", t)
            if m:
                close(); kernel = m.group(1)
                continue
            if not t.strip() or t.lstrip().startswith("."):
                continue
            if areg.search(t):
                mm = mine.match(t)
                mf = mine_mfma.match(t)
                if mf:
                    lo, hi = int(mf.group(1)), int(mf.group(2))
                    if (lo, hi) != (int(mf.group(3)), int(mf.group(4))) or hi - lo != 15 or lo % 16 or hi > 127:
                        bad.append(f"{kernel}: {t.strip()}")
                    for i in range(lo, hi + 1):
                        mfmas[str(i)] += 1
                    continue
                if not mm:
                    bad.append(f"{kernel}: {t.strip()}")
                elif mm.group(2) is not None:
                    reads[mm.group(2)] += 1
                else:
                    writes[mm.group(3)] += 1
    close()
    if bad:
        raise RuntimeError(f"{asm_path}: compiler-allocated accumulation registers ({len(bad)} findings), e.g. {bad[:3]}")


# Loads-in-flight audit (round 4).  The HBM-level passes are bandwidth kernels whose speed hangs on every row load of a block being
# issued before the first one is waited for; run-time switches inside the row loop once put each load into its own branch with a
# `s_waitcnt vmcnt(0)` behind it (2.95 instead of 4.45 TB/s) and nothing failed.  For the kernels listed here the build checks the
# device assembly: the longest run of 16-byte global loads with no vmcnt wait, label or branch in between must reach the given length.
LOAD_RUNS = {"ffc_k_big.hip": [(r"^_Z10big_kernelILi(16|32)ELi[01]ELb[01]E", 16), (r"^_Z14big_all_kernelILi[01]ELb1E", 16)]}


def check_load_runs(asm_path, rules):
    import re
    kernel, run, best = None, 0, {}
    with open(asm_path) as fh:
        for line in fh:
            t = line.split(";")[0].strip()
            m = re.match(r"^(_Z\w+):", t)
            if m:
                kernel, run = m.group(1), 0
                continue
            if kernel is None or not t:
                continue
            if t.startswith("global_load_dwordx4"):
                run += 1
                best[kernel] = max(best.get(kernel, 0), run)
            elif re.match(r"^(s_waitcnt.*vmcnt|s_cbranch|s_branch|\.LBB|s_barrier)", t):
                run = 0
    bad, seen = [], 0
    for pat, need in rules:
        for k, v in best.items():
            if re.match(pat, k):
                seen += 1
                if v < need:
                    bad.append(f"{k}: longest run of 16-byte loads in flight {v} < {need}")
    if bad or not seen:
        raise RuntimeError(f"{asm_path}: loads-in-flight audit: {bad[:4] if bad else 'no kernel matched the rules'}")


# Register-spill audit (round 6).  The backward kernel of fft 1024 / 2048 had been spilling 34 - 43 registers into scratch memory inside its pair loop
# since round 2, the BLH weight-gradient kernel of conv1d 66 - 77 (a missing __launch_bounds__), and nothing said so: fft 2048 backward -27 % once
# the spills were gone (profiles/r06_ab_fft2048.txt).  Every translation unit is now compiled with -Rpass-analysis=kernel-resource-usage; the
# per-kernel figures go to lib/resource_usage.txt and a kernel with more than SCRATCH_LIMIT bytes of scratch per lane fails the build (the few
# kernels below it spill 2 - 8 registers once per job, outside their loops).
SCRATCH_LIMIT = 64


def parse_resource_usage(text):
    import re
    rows, cur = [], None
    for line in text.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"fn": m.group(1)}
            rows.append(cur)
            continue
        for key, name in (("VGPRs Spill", "spill"), (r"ScratchSize \[bytes/lane\]", "scratch"), ("AGPRs", "agprs"), ("VGPRs", "vgprs"),
                          (r"Occupancy \[waves/SIMD\]", "occupancy"), (r"LDS Size \[bytes/block\]", "lds")):
            m = re.search(key + r": (\d+)", line)
            if m and cur is not None:
                cur[name] = int(m.group(1))
                break
    return rows


def build_hip(force=False, verbose=False, variant=None, extra_flags=(), srcs=None):
    """Compile every translation unit for gfx950 in parallel, then link the shared library.
    variant: tuning build with `extra_flags` under lib/variants/<name>/ (A/B runs: FFC_LIB=<that .so>)."""
    from concurrent.futures import ThreadPoolExecutor
    global HIP_FLAGS
    # knock-out / experiment switches (FFC_KO, FFC_EXP_*, FFC_NO_SETTLE, ...) produce wrong or unguarded results by design: they
    # may only go into a named variant under lib/variants/, never into the library the package loads
    risky = [f for f in list(extra_flags) + os.environ.get("HIPCC_EXTRA_FLAGS", "").split()
             if f.startswith("-DFFC_KO") or f.startswith("-DFFC_EXP_") or f.startswith("-DFFC_NO_SETTLE") or f.startswith("-DFFC_BWD_PROF")]
    if variant is None and (risky or extra_flags):
        raise RuntimeError(f"build.py: extra flags {list(extra_flags)} are only accepted with --variant NAME (the default target is the product)")
    lib_dir = LIB if variant is None else os.path.join(LIB, "variants", variant)
    os.makedirs(lib_dir, exist_ok=True)
    obj_dir = os.path.join(lib_dir, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    hip_so = os.path.join(lib_dir, "libflashfftconv_hip.so")
    flags = HIP_FLAGS + list(extra_flags)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdr_time = max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith(".h"))
    hdr_time = max(hdr_time, os.path.getmtime(os.path.join(HERE, "..", "include", "flashfftconv_hip.h")))

    def compile_one(f):
        # checked translation units keep the device assembly of the SAME compile (-save-temps=obj) for check_agpr
        checked = f in AGPR_CHECKED or f in LOAD_RUNS
        src = os.path.join(CSRC, f)
        out = os.path.join(obj_dir, f + ".o")
        stem = os.path.splitext(f)[0]
        asm = os.path.join(obj_dir, stem + "-hip-amdgcn-amd-amdhsa-gfx950.s")
        marker = os.path.join(obj_dir, stem + ".agpr_ok")       # the check passed for this object (the 50 MB of temporaries are not kept)
        if force or not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), hdr_time) or (checked and not os.path.exists(marker)):
            if os.path.exists(marker):
                os.remove(marker)
            only = os.environ.get("FFC_VARIANT_ONLY")      # tuning builds: extra flags for one translation unit only
            fl = flags if (not only or f == only) else HIP_FLAGS
            is_hip = f.endswith(".hip")
            cmd = [hipcc] + fl + (["-save-temps=obj"] if checked else []) + (["-Rpass-analysis=kernel-resource-usage"] if is_hip else []) + ["-c", "-x", "hip", src, "-o", out]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, cwd=obj_dir, capture_output=True, text=True)
            if r.returncode != 0:
                sys.stderr.write(r.stderr[-8000:])
                raise subprocess.CalledProcessError(r.returncode, cmd)
            if is_hip:
                rows = parse_resource_usage(r.stderr)
                with open(os.path.join(obj_dir, stem + ".ru.txt"), "w") as fh:
                    for k in rows:
                        fh.write(f"{f} {k['fn']} vgprs {k.get('vgprs')} agprs {k.get('agprs')} scratch {k.get('scratch')} spill {k.get('spill')} "
                                 f"occupancy {k.get('occupancy')}\n")
                bad = [k for k in rows if (k.get("scratch") or 0) > SCRATCH_LIMIT]
                if bad and not any(x.startswith("-DFFC_KO") for x in fl):
                    os.remove(out)
                    raise RuntimeError(f"{f}: register spills: " + "; ".join(f"{k['fn'][:90]} scratch {k['scratch']} B/lane ({k.get('spill')} VGPRs)" for k in bad[:4]))
            if checked and not os.environ.get("FFC_SKIP_AGPR_CHECK"):      # (knock-out timing builds skip the check)
                try:
                    if f in AGPR_CHECKED:
                        check_agpr(asm)
                    if f in LOAD_RUNS and not any(x.startswith("-DFFC_KO") for x in fl):
                        check_load_runs(asm, LOAD_RUNS[f])
                except Exception:
                    os.remove(out)          # a failed check must not leave an object the next build would link
                    raise
                open(marker, "w").write("ok\n")
            if checked and not os.environ.get("FFC_KEEP_TEMPS"):      # -save-temps leaves ~50 MB per unit: every gpurun push carried them
                for g in os.listdir(obj_dir):
                    if g.startswith(stem + "-hip-") or g.startswith(stem + "-host-") or g.startswith(stem + ".hip-"):
                        os.remove(os.path.join(obj_dir, g))
            return out, True
        return out, False

    jobs = sorted(HIP_SRCS, key=lambda f: not (f in AGPR_CHECKED or f == "ffc_k_conv.hip" or f.startswith('ffc_conv1d_t')))   # longest first
    with ThreadPoolExecutor(max_workers=min(10, os.cpu_count() or 4)) as ex:
        res = list(ex.map(compile_one, jobs))
    objs = [o for o, _ in res]
    if force or any(ch for _, ch in res) or not os.path.exists(hip_so):
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", hip_so] + objs)
    # per-kernel register / scratch figures of the library as built (one line per kernel)
    with open(os.path.join(lib_dir, "resource_usage.txt"), "w") as fh:
        for f in sorted(HIP_SRCS):
            ru = os.path.join(obj_dir, os.path.splitext(f)[0] + ".ru.txt")
            if os.path.exists(ru):
                fh.write(open(ru).read())
    return hip_so


def build_sim(force=False):
    os.makedirs(LIB, exist_ok=True)
    srcs = [os.path.join(CSRC, f) for f in SIM_SRCS]
    if force or _stale(SIM_SO, srcs):
        # -O0: the simulator is one large translation unit (every kernel mode x 2 backends x 2 dtypes); -O1 took 7.5 min to
        # compile at the end of round 2 for a test run of 23 s, -O0 takes 50 s for 40 s
        subprocess.check_call(["g++", "-O0", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", SIM_SO] + srcs)
    return SIM_SO


def build_all(force=False, verbose=False):
    """HIP library and the CPU simulator side by side (the simulator is one large g++ translation unit)."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=2) as ex:
        sim = ex.submit(build_sim, force)
        hip = build_hip(force, verbose)
        return hip, sim.result()


if __name__ == "__main__":
    force = "--force" in sys.argv
    if "--variant" in sys.argv:      # python build.py --variant NAME -DFFC_X=1 ...
        name = sys.argv[sys.argv.index("--variant") + 1]
        extra = []
        for a in sys.argv[1:]:
            if a.startswith("-D") or a.startswith("-m"):
                extra += a.split("=", 1) if a.startswith("-mllvm=") else [a]      # -mllvm=--flag -> -mllvm --flag
        print(build_hip(force, verbose=False, variant=name, extra_flags=extra))
    else:
        print(build_all(force, verbose=True))
