"""Copies the judged summaries of gpurun_out/r06_end (benchmarks/measure_r06.sh) into profiles/r06_*."""
import csv, collections, glob, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(ROOT, "gpurun_out", "r06_end"), os.path.join(ROOT, "profiles")


def find(d, pat):
    r = glob.glob(os.path.join(O, d, "**", pat), recursive=True)
    return r[0] if r else None


# the contract line (last stdout line) and, beside it, every line of the run (table rows) and the complete object
open(f"{P}/r06_bench.json", "w").write(open(f"{O}/bench.json").read().strip().splitlines()[-1] + "\n")
open(f"{P}/r06_bench_stdout.jsonl", "w").write(open(f"{O}/bench.json").read())
if os.path.exists(f"{O}/bench_full.json"):
    shutil.copy(f"{O}/bench_full.json", f"{P}/r06_bench_full.json")
# (cfg4_s / cfg3_s: the last call's runs on the final code, benchmarks/run_r06_s.sh; cfg4 / cfg3: benchmarks/measure_r06.sh)
for d, name in (("stats", "r06_kernel_stats.csv"), ("cfg4", "r06_cfg4_kernel_stats.csv"), ("cfg4_s", "r06_cfg4_kernel_stats.csv"),
                ("cfg3", "r06_cfg3_kernel_stats.csv"), ("cfg3_s", "r06_cfg3_kernel_stats.csv"), ("f1k", "r06_fft2048_kernel_stats.csv")):
    f = find(d, "*kernel_stats.csv")
    if f:
        shutil.copy(f, f"{P}/{name}")
# the kernel trace of the profiled bench run: start / end of every launch of the timed steps (judge item: launch times inside the step)
tr = find("stats", "*kernel_trace.csv")
if tr:
    rows = list(csv.DictReader(open(tr)))
    keep = [r for r in rows if any(s in r["Kernel_Name"] for s in ("conv_kernel", "bwd_kernel", "kfft_kernel", "dkifft_kernel"))]
    with open(f"{P}/r06_step_trace.csv", "w") as fh:
        fh.write("kernel,start_ns,end_ns,duration_us\n")
        for r in keep[-400:]:
            fh.write(f"{r['Kernel_Name'][:60].replace(',', ';')},{r['Start_Timestamp']},{r['End_Timestamp']},{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.1f}\n")
for src, dst in (("hyena_train.jsonl", "r06_hyena_train.jsonl"), ("short_probe.txt", "r06_short_probe.txt"), ("m2_bert_fwd.jsonl", "r06_m2_bert_fwd.jsonl")):
    if os.path.exists(f"{O}/{src}"):
        shutil.copy(f"{O}/{src}", f"{P}/{dst}")
lines = [l[:3000] for l in open(f"{O}/stats.log") if "amdgpu.ids" not in l][-3:]
open(f"{P}/r06_bench_under_rocprof.log", "w").writelines(lines)


def pmc(prefix, kernel_sub, out_name, title, alg_mb, alg_note):
    out = {}
    for d in (prefix + "1", prefix + "2", prefix + "3", prefix + "4"):
        f = find(d, "*counter_collection.csv")
        if not f:
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if kernel_sub in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            out[k] = sum(v) / len(v)
    with open(f"{P}/{out_name}", "w") as fh:
        w = lambda s: fh.write(s + "\n")
        w("# rocprofv3 --pmc (separate passes per counter group, only --kernel-trace next to --pmc; benchmarks/measure_r06.sh),")
        w(f"# {title}, config 2 (B16 H768 L16384, fft 32768, bf16), per dispatch (avg of 4)")
        w("# FETCH_SIZE / WRITE_SIZE count requests between L2 and the fabric (TCC_EA0_RDREQ / WRREQ): Infinity-Cache hits are included,")
        w("# so this is fabric traffic, an UPPER bound on HBM traffic (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is uncalibrated")
        for k, v in out.items():
            w(f"{k:28s}{v:.4e}")
        fs, ws, wc = out.get("FETCH_SIZE"), out.get("WRITE_SIZE"), out.get("SQ_WAVE_CYCLES")
        if fs and ws:
            w(f"L2<->fabric read bytes  (FETCH_SIZE KB x1024 x2 gfx950 correction) = {fs*2048/1e6:.1f} MB")
            w(f"L2<->fabric write bytes (WRITE_SIZE KB x1024)                       = {ws*1024/1e6:.1f} MB")
            w(f"traffic per launch = {(fs*2048+ws*1024)/1e6:.1f} MB  (algorithmic {alg_mb} MB: {alg_note})")
        if wc:
            w(f"wave time split: active {out['SQ_ACTIVE_INST_ANY']/wc*100:.1f}%  wait_inst {out['SQ_WAIT_INST_ANY']/wc*100:.1f}%  wait_any {out['SQ_WAIT_ANY']/wc*100:.1f}% ; VALU issue {out['SQ_ACTIVE_INST_VALU']/wc*100:.1f}% of wave cycles")
            w(f"VALU instructions per wave per pair: {out['SQ_INSTS_VALU']/6144/8:.0f}")
        if "SQ_LDS_IDX_ACTIVE" in out and wc:
            w(f"LDS bank-conflict cycles / LDS active cycles = {out['SQ_LDS_BANK_CONFLICT']/out['SQ_LDS_IDX_ACTIVE']:.2f}; LDS wait = {out['SQ_WAIT_INST_LDS']/wc*100:.1f}% of wave cycles; MFMA busy cycles {out['SQ_VALU_MFMA_BUSY_CYCLES']:.3e}")
    print(open(f"{P}/{out_name}").read()[-900:])


pmc("c", "conv_kernel", "r06_pmc_conv_kernel.txt", "benchmarks/prof_step_kernels.py fwd: conv_kernel<Geo<32,32,32>,bf16,HALF,SZ> training forward incl. k -> k_f of the head (stores the spectra)",
    906 + 805 + 50, "u 403 + y 403 + k_f written and read 101 + k 50 + saved spectra 805")
pmc("b", "bwd_kernel", "r06_pmc_bwd_kernel.txt", "benchmarks/prof_step_kernels.py bwd: bwd_kernel<Geo<32,32,32>,bf16,HALF,ZM=1> fused backward on saved spectra incl. the dk tail (dout rows by LDS-DMA)", 906 + 50 + 805,
    "dout 403 + du 403 + k_f 101 + dk 50 (fp32; the dk_f sums never leave the registers) + saved spectra 805")

# the GPU suite on the measured code (the driver's round-end command)
pg = f"{O}/pytest_gpu.txt"
if os.path.exists(pg):
    keep = [l for l in open(pg) if " passed" in l or " failed" in l or l.startswith("real")]
    open(f"{P}/r06_pytest_gpu.txt", "w").write("# python -m pytest tests -m gpu -x -q on the final round-6 code (benchmarks/measure_r06.sh)\n" + "".join(keep))

rv = f"{O}/reference_verbatim.log"
if os.path.exists(rv):
    keep = [l for l in open(rv) if " passed" in l or " failed" in l or l.startswith("real") or "selection" in l or "case(s)" in l]
    open(f"{P}/r06_reference_verbatim.log", "w").write("# the reference's two test files in FULL (FFC_REF_TESTS_FULL=1), unmodified, on the final round-6 library (benchmarks/measure_r06.sh)\n" + "".join(l[:600] for l in keep))
