"""Copies the judged summaries of gpurun_out/r01_end (benchmarks/measure_r01_end.sh) into profiles/r01_end_*."""
import csv, collections, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(ROOT, "gpurun_out", "r01_end"), os.path.join(ROOT, "profiles")
open(f"{P}/r01_end_bench.json", "w").write(open(f"{O}/bench.json").read().strip().splitlines()[-1] + "\n")
shutil.copy(f"{O}/stats/b_kernel_stats.csv", f"{P}/r01_end_kernel_stats.csv")
shutil.copy(f"{O}/sweep.jsonl", f"{P}/r01_end_sweep.jsonl")
lines = [l[:2000] for l in open(f"{O}/stats.log") if "amdgpu.ids" not in l][-3:]
open(f"{P}/r01_end_bench_under_rocprof.log", "w").writelines(lines)
out = {}
for d in ("p1", "p2", "p3", "p4"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{O}/{d}/p_counter_collection.csv")):
        if "conv_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out[k] = sum(v) / len(v)
with open(f"{P}/r01_end_pmc_conv_kernel.txt", "w") as f:
    w = lambda s: f.write(s + "\n")
    w("# rocprofv3 --pmc (separate passes per counter group, no tracing domains besides --kernel-trace; benchmarks/measure_r01_end.sh),")
    w("# benchmarks/prof_conv.py: conv_kernel<Geo<32,32,32>,bf16,HALF> forward, config 2 (B16 H768 L16384), per dispatch (avg of 4)")
    for k, v in out.items():
        w(f"{k:28s}{v:.4e}")
    fs, ws, wc = out["FETCH_SIZE"], out["WRITE_SIZE"], out["SQ_WAVE_CYCLES"]
    w(f"L2<->fabric read bytes  (FETCH_SIZE KB x1024 x2 gfx950 correction) = {fs*2048/1e6:.1f} MB")
    w(f"L2<->fabric write bytes (WRITE_SIZE KB x1024)                       = {ws*1024/1e6:.1f} MB")
    w(f"traffic per launch = {(fs*2048+ws*1024)/1e6:.1f} MB  (algorithmic 906 MB: u 403 + y 403 + k_f 101; k_f is re-fetched per pair)")
    w(f"wave time split: active {out['SQ_ACTIVE_INST_ANY']/wc*100:.1f}%  wait_inst {out['SQ_WAIT_INST_ANY']/wc*100:.1f}%  wait_any {out['SQ_WAIT_ANY']/wc*100:.1f}% ; VALU issue {out['SQ_ACTIVE_INST_VALU']/wc*100:.1f}% of wave cycles")
    w(f"VALU instructions per wave per pair: {out['SQ_INSTS_VALU']/6144/8:.0f}; LDS bank-conflict cycles / LDS active cycles = {out['SQ_LDS_BANK_CONFLICT']/out['SQ_LDS_IDX_ACTIVE']:.2f}; LDS wait = {out['SQ_WAIT_INST_LDS']/wc*100:.1f}% of wave cycles")
print(open(f"{P}/r01_end_pmc_conv_kernel.txt").read()[-700:])
