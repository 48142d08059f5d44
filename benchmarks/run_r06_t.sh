# round-6 GPU call T: PMC passes of the fft-2048 kernels (BASELINE sweep row L = 1K) and of the 4-pass fft-131072 kernels at L = 64K: traffic against algorithmic bytes
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_t; mkdir -p $O
cd /tmp
pmc() { name=$1; shift; args="$1"; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -o p -- python $R/benchmarks/prof_one.py $args > $O/$name.log 2>&1; }
for c in "f2k|2048 16 768 1024 both plain bfloat16" "f128k|131072 16 768 65536 both plain bfloat16" "f64k|65536 16 768 32768 both plain bfloat16"; do
  n=${c%%|*}; a=${c#*|}
  pmc pmc_${n}_1 "$a" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
  pmc pmc_${n}_2 "$a" SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU
  pmc pmc_${n}_3 "$a" FETCH_SIZE
  pmc pmc_${n}_4 "$a" WRITE_SIZE
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$n -o s -- python $R/benchmarks/prof_one.py $a > $O/stats_$n.log 2>&1
done
cd $R; python benchmarks/summarize_pmc_generic.py r06_t f2k f64k f128k > $O/pmc_multipass.txt; cat $O/pmc_multipass.txt
