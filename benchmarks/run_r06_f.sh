# round-6 GPU call F: short sequences.  (1) the pipelined persistent forward of the single-tile sizes (Body::conv_small: the next tile's rows, gate and k_f in
# flight under the current transform) against the round-5 job loop (lib/variants/nopipe = -DFFC_SMALL_PIPE=0), same box, interleaved; (2) PMC counters of the fft-1024
# forward kernel, gated B64 H768 and plain B16 H768; (3) parity of the small sizes on the new library
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_f; mkdir -p $O
cd $R
V=$R/flash-fft-conv_amd/lib/variants
for i in 1 2 3; do
  for v in product nopipe; do
    if [ $v = product ]; then unset FFC_LIB; else export FFC_LIB=$V/$v/libflashfftconv_hip.so; fi
    echo "== $v" >> $O/ab_small.txt
    timeout 600 python benchmarks/ab_lib.py 1024,64,768,1024,g 1024,16,768,1024 1024,16,768,512 256,64,768,256,g 512,64,768,512,g 512,16,768,256 2048,16,768,1024 2>&1 | grep -v amdgpu.ids | sed 's/digests.*//' >> $O/ab_small.txt
  done
done
unset FFC_LIB
cat $O/ab_small.txt
( time timeout 1500 python -m pytest tests/test_flashfftconv_gpu.py tests/test_spectrum_gpu.py tests/test_graph_gpu.py tests/test_determinism_gpu.py tests/test_robustness_gpu.py -m gpu -x -q -k "256 or 512 or 1024 or 2048 or golden or determin or robust or graph or spectrum" ) > $O/pytest_small.txt 2>&1; tail -4 $O/pytest_small.txt
cd /tmp
pmc() { name=$1; shift; shape=$1; shift; timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -o p -- python $R/benchmarks/prof_one.py $shape > $O/$name.log 2>&1; }
for v in product nopipe; do
  if [ $v = product ]; then unset FFC_LIB; else export FFC_LIB=$V/$v/libflashfftconv_hip.so; fi
  pmc ${v}_g_1 "1024 64 768 1024 fwd gated" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
  pmc ${v}_g_2 "1024 64 768 1024 fwd gated" SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU
  pmc ${v}_g_3 "1024 64 768 1024 fwd gated" FETCH_SIZE
  pmc ${v}_g_4 "1024 64 768 1024 fwd gated" WRITE_SIZE
  pmc ${v}_p_1 "1024 16 768 1024 fwd" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
  pmc ${v}_p_3 "1024 16 768 1024 fwd" FETCH_SIZE
done
unset FFC_LIB
cd $R
python - <<'PY'
import csv, glob, os, collections
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r06_f")
for d in sorted(glob.glob(O + "/*_[gp]_[1-4]")):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    ks = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    if not fs:
        print(os.path.basename(d), "no counters:", open(d + ".log").read()[-300:].replace("\n", " | ")); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "conv_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(ks[0])) if "conv_kernel" in r["Kernel_Name"]] if ks else []
    print(os.path.basename(d), {k: f"{sum(v)/len(v):.4e}" for k, v in acc.items()}, "kernel us (under pmc)", round(sum(dur) / max(len(dur), 1), 1))
PY
