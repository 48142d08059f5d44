"""Loads-in-flight view of a kernel's device assembly (how round 4's serial-load findings were made; no GPU needed).

    python benchmarks/isa_view.py ffc_k_bwdz.hip 'bwd_kernelIN3ffc3GeoILi32ELi32ELi32EEELi0ELb1ELi1E' [-DFLAG ...]

compiles the translation unit for gfx950 with -save-temps into /tmp/ffc_isa/<unit>/ (cached: delete the directory to recompile) and
prints, for every kernel whose mangled name contains the pattern, the sequence of memory events in program order, run-length coded:

    L  16-byte global load     l  narrower global load     D  LDS-DMA copy (global_load_lds)     G  global store
    S  16-byte LDS write       M  MFMA                     B  workgroup barrier                  Wn s_waitcnt vmcnt(n)
    <slow>  the element-wise arm of a row access (loads of 2 bytes with a wait each)

Both arms of a wave-uniform branch appear one after the other (the text is linear), so `Lx16 W15 ...` reads "16 loads requested, then
waited for one by one", while `L W0 L W0 ...` or `D W0 D W0 ...` is the thing to look for: one request per memory round trip.  Found
this way in round 4: the long-side loads of the HBM-level passes, the gate rows of every gated launch, the rows of the multi-pass
forward kernels, the filter row of k -> k_f, the LDS-DMA prologue of the saved-spectra backward."""
import os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "flash-fft-conv_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-mllvm", "--amdgpu-mfma-vgpr-form", "-fPIC"]


def assembly(unit, extra):
    d = os.path.join("/tmp/ffc_isa", os.path.splitext(unit)[0] + ("_" + "_".join(x.strip("-") for x in extra) if extra else ""))
    asm = os.path.join(d, os.path.splitext(unit)[0] + "-hip-amdgcn-amd-amdhsa-gfx950.s")
    if not os.path.exists(asm):
        os.makedirs(d, exist_ok=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + list(extra) + ["-save-temps=obj", "-c", "-x", "hip", os.path.join(CSRC, unit),
                               "-o", os.path.join(d, "unit.o")], cwd=d)
    return asm


def view(lines):
    out = []
    for l in lines:
        t = l.split(";")[0].strip()
        if t.startswith("global_load_lds"): out.append("D")
        elif t.startswith("global_load_dwordx4"): out.append("L")
        elif t.startswith("global_load"): out.append("l")
        elif t.startswith("global_store"): out.append("G")
        elif t.startswith("ds_write_b128"): out.append("S")
        elif t.startswith("v_mfma"): out.append("M")
        elif t.startswith("s_barrier"): out.append("B")
        else:
            m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", t)
            if m: out.append("W" + m.group(1))
    res, prev, c = [], None, 0
    for k in out + [None]:
        if k == prev:
            c += 1
            continue
        if prev is not None:
            res.append(prev + (f"x{c}" if c > 1 else ""))
        prev, c = k, 1
    return re.sub(r"(lx\d+ W0(x\d)? (Sx2 )?)+", "<slow> ", " ".join(res))


if __name__ == "__main__":
    unit, pat = sys.argv[1], sys.argv[2]
    src = open(assembly(unit, [a for a in sys.argv[3:] if a.startswith("-")])).read().split("\n")
    starts = [i for i, l in enumerate(src) if re.match(r"^_Z\w+:", l)]
    for n, st in enumerate(starts):
        name = src[st].split(":")[0]
        if pat not in name:
            continue
        en = st + next(i for i, l in enumerate(src[st:]) if "s_endpgm" in l)
        print(name + "\n  " + view(src[st:en]) + "\n")
