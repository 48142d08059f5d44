# round-6 GPU call L: short sequences.  (1) keep the spectra or recompute, single-tile and small fused sizes, module rows, interleaved twice;
# (2) PMC passes of the fft-1024 kernels (README gated fp16 B64 H768; plain bf16 B16 H768), counters as the config-2 files
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_l; mkdir -p $O
cd $R
for i in 1 2; do
  for s in default 0; do
    echo "== FFC_SAVE_SPECTRUM=$s" >> $O/ab_save_small.txt
    for shape in "256 64 768 256 768 gated" "1024 64 768 1024 768 gated" "2048 64 768 2048 768 gated" "4096 64 768 4096 768 gated" "8192 64 768 8192 768 gated" "16384 32 768 16384 768 gated" \
                 "1024 16 768 1024 768" "2048 16 768 1024 768" "4096 16 768 2048 768" "8192 16 768 4096 768" "16384 16 768 8192 768" "32768 16 768 16384 768" "32768 16 768 16384 768 gated" "16384 8 1024 8192 1024 gated"; do
      if [ $s = default ]; then unset FFC_SAVE_SPECTRUM; else export FFC_SAVE_SPECTRUM=$s; fi
      timeout 300 python benchmarks/sweep.py row $shape 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print(r['row'][:30], 'gated' if r['gated'] else 'plain', 'fwd', r['fwd_ms'], 'bwd', r['bwd_ms'], 'fwd+bwd', r['fwd_bwd_ms'], 'infer', r.get('fwd_infer_ms'), 'peak MB', round(r['peak_fwd_bwd'] / 1e6))
" >> $O/ab_save_small.txt
    done
  done
done
unset FFC_SAVE_SPECTRUM
cat $O/ab_save_small.txt
cd /tmp
pmc() { name=$1; shift; args="$1"; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -o p -- python $R/benchmarks/prof_one.py $args > $O/$name.log 2>&1; }
for c in "g64|1024 64 768 1024 both gated float16" "p16|1024 16 768 1024 both plain bfloat16"; do
  n=${c%%|*}; a=${c#*|}
  pmc pmc_${n}_1 "$a" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
  pmc pmc_${n}_2 "$a" SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU
  pmc pmc_${n}_3 "$a" FETCH_SIZE
  pmc pmc_${n}_4 "$a" WRITE_SIZE
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$n -o s -- python $R/benchmarks/prof_one.py $a > $O/stats_$n.log 2>&1
done
cd $R; python benchmarks/summarize_pmc_short.py r06_l; cp profiles/r06_pmc_fft1024.txt $O/
find $O -name "*kernel_stats.csv" | head
