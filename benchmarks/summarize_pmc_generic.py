"""Per-kernel PMC digest of any benchmarks/prof_one.py shape: reads gpurun_out/<dir>/pmc_<case>_<group>/**/counter_collection.csv (+ stats_<case> kernel stats)
-> stdout.  usage: summarize_pmc_generic.py <dir> <case> [<case> ...]"""
import csv, collections, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", sys.argv[1])
for case in sys.argv[2:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(O, f"pmc_{case}_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            if any(s in kn for s in ("conv_kernel", "conv_rp_kernel", "bwd_kernel", "bwd_rp_kernel", "big_", "kfft", "dkifft")):
                acc[kn.split("(")[0][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = {}
    for f in glob.glob(os.path.join(O, f"stats_{case}", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Name"].split("(")[0][:90]] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
    for kn, c in acc.items():
        v = {k: sum(x) / len(x) for k, x in c.items()}
        d = dur.get(kn)
        print(f"== {case}: {kn}" + (f"   avg {d[0]:.1f} us over {d[1]} launches" if d else ""))
        wc = v.get("SQ_WAVE_CYCLES")
        if wc and "SQ_ACTIVE_INST_ANY" in v:
            print(f"  waves {v['SQ_WAVES']:.0f}; wave time: active {v['SQ_ACTIVE_INST_ANY']/wc*100:.1f} %  wait_inst {v['SQ_WAIT_INST_ANY']/wc*100:.1f} %  wait_any {v['SQ_WAIT_ANY']/wc*100:.1f} %; "
                  f"VALU issue {v['SQ_ACTIVE_INST_VALU']/wc*100:.1f} % of wave cycles; VALU instructions {v['SQ_INSTS_VALU']:.4g}")
        if wc and "SQ_INSTS_LDS" in v:
            print(f"  LDS instructions {v['SQ_INSTS_LDS']:.4g}, conflict / active {v['SQ_LDS_BANK_CONFLICT']/max(v['SQ_LDS_IDX_ACTIVE'],1):.2f}, LDS wait {v['SQ_WAIT_INST_LDS']/wc*100:.1f} % of wave cycles; "
                  f"VMEM rd / wr {v['SQ_INSTS_VMEM_RD']:.4g} / {v['SQ_INSTS_VMEM_WR']:.4g}; MFMA busy cycles {v['SQ_VALU_MFMA_BUSY_CYCLES']:.3g}")
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            rd, wr = v['FETCH_SIZE'] * 2048 / 1e6, v['WRITE_SIZE'] * 1024 / 1e6
            print(f"  fabric read {rd:.1f} MB (FETCH_SIZE x 2 KB, gfx950 correction), write {wr:.1f} MB" + (f" -> {(rd + wr) / d[0]:.2f} TB/s over the launch" if d else ""))
