# round-6 GPU call U: PMC passes of an HBM-level size at the sweep's shape (fft 262144 = 16 x 16384, bf16 B16 H384 L = 131072) and of config 4 (fft 4M, B1 H16 L = 1M: runs 2M points, half rows)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_u; mkdir -p $O
cd /tmp
pmc() { name=$1; shift; args="$1"; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -o p -- python $R/benchmarks/prof_one.py $args > $O/$name.log 2>&1; }
for c in "f256k|262144 16 384 131072 both plain bfloat16" "f1m|1048576 16 96 524288 both plain bfloat16"; do
  n=${c%%|*}; a=${c#*|}
  pmc pmc_${n}_1 "$a" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
  pmc pmc_${n}_2 "$a" SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU
  pmc pmc_${n}_3 "$a" FETCH_SIZE
  pmc pmc_${n}_4 "$a" WRITE_SIZE
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$n -o s -- python $R/benchmarks/prof_one.py $a > $O/stats_$n.log 2>&1
done
cd $R; python benchmarks/summarize_pmc_generic.py r06_u f256k f1m > $O/pmc_levels.txt; cat $O/pmc_levels.txt
