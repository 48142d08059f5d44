"""Per-kernel PMC digest of the short sizes: reads the rocprofv3 --pmc passes of benchmarks/run_r06_l.sh / run_r06_p.sh
(gpurun_out/<dir>/pmc_<case>_<group>/...counter_collection.csv) -> profiles/r06_pmc_fft1024.txt (dir r06_l) or profiles/r06_pmc_fft4096.txt (dir r06_p)"""
import csv, collections, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "r06_l")
MID = len(sys.argv) > 1 and sys.argv[1] == "r06_p"
CASES = ({"f4k": ("fft 4096 plain bf16 B16 H768 L2048 (sweep row L = 2048)", 16, False, 2048, 4096), "f8k": ("fft 8192 plain bf16 B16 H768 L4096 (sweep row L = 4096)", 16, False, 4096, 8192)}
         if MID else {"g64": ("fft 1024 gated fp16 B64 H768 L1024 (README row)", 64, True, 1024, 1024), "p16": ("fft 1024 plain bf16 B16 H768 L1024", 16, False, 1024, 1024)})
out = open(os.path.join(ROOT, "profiles", "r06_pmc_fft4096.txt" if MID else "r06_pmc_fft1024.txt"), "w")
w = lambda s: (out.write(s + "\n"), print(s))
w("# rocprofv3 --pmc, separate passes per counter group next to --kernel-trace only (benchmarks/run_r06_l.sh), benchmarks/prof_one.py through the module:")
w("# training forward (conv_kernel<..., SZ>: stores the pair's spectrum, gated also the output before the postgate) and fused backward on the saved spectra,")
w("# per dispatch (average over the launches of the run).  FETCH_SIZE x 2 KB (gfx950 correction) / WRITE_SIZE KB = L2 <-> fabric bytes (upper bound on HBM bytes).")
for case, (title, B, gated, L, N) in CASES.items():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(O, f"pmc_{case}_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            key = "conv_kernel" if "conv_kernel" in kn else "bwd_kernel" if "bwd_kernel" in kn else None
            if key:
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    pairs = B // 2 * 768
    for key, c in acc.items():
        v = {k: sum(x) / len(x) for k, x in c.items()}
        w(f"== {title}: {key}")
        w("  " + "  ".join(f"{k}={x:.4g}" for k, x in sorted(v.items())))
        wc = v.get("SQ_WAVE_CYCLES")
        if wc and "SQ_ACTIVE_INST_ANY" in v:
            w(f"  waves {v['SQ_WAVES']:.0f}; wave time: active {v['SQ_ACTIVE_INST_ANY']/wc*100:.1f} %  wait_inst {v['SQ_WAIT_INST_ANY']/wc*100:.1f} %  wait_any {v['SQ_WAIT_ANY']/wc*100:.1f} %; "
              f"VALU issue {v['SQ_ACTIVE_INST_VALU']/wc*100:.1f} % of wave cycles; VALU per pair {v['SQ_INSTS_VALU']/pairs:.0f}; busy cycles per SE-sum {v['SQ_BUSY_CYCLES']:.3g}")
        if wc and "SQ_INSTS_LDS" in v:
            w(f"  LDS per pair {v['SQ_INSTS_LDS']/pairs:.0f}, conflict / active {v['SQ_LDS_BANK_CONFLICT']/max(v['SQ_LDS_IDX_ACTIVE'],1):.2f}, LDS wait {v['SQ_WAIT_INST_LDS']/wc*100:.1f} % of wave cycles; "
              f"VMEM rd / wr per pair {v['SQ_INSTS_VMEM_RD']/pairs:.1f} / {v['SQ_INSTS_VMEM_WR']/pairs:.1f}; MFMA busy {v['SQ_VALU_MFMA_BUSY_CYCLES']:.3g}")
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            rows = B * 768 * L * 2 / 1e6
            zu = N / L      # the kept spectrum in units of one row tensor (4 N bytes per pair against 4 L)
            if not gated and N <= 1024:
                zu = 0 if key == "conv_kernel" else 1      # plain single-tile sizes keep no spectrum: the backward reads u instead
            alg = (rows * ((3 if gated else 1) + (2 if gated else 1) + zu) if key == "conv_kernel" else rows * ((5 if gated else 1) + zu + (3 if gated else 1)))
            w(f"  fabric read {v['FETCH_SIZE']*2048/1e6:.1f} MB, write {v['WRITE_SIZE']*1024/1e6:.1f} MB; algorithmic {alg:.0f} MB (rows of {rows:.0f} MB: "
              + ("u, gates, y, yraw, spectra" if key == "conv_kernel" else "dout, gates, u, yraw, spectra, du, dgates") + ")")
out.close()
