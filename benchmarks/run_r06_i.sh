# round-6 GPU call I: (1) HBM-level sizes with and without the kept inner spectra: module rows, FFC_SAVE_SPECTRUM=0 / default, interleaved;
# (2) bench.py on the final library (contract line + tables)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_i; mkdir -p $O
cd $R
for i in 1 2; do
  for s in default 0; do
    echo "== FFC_SAVE_SPECTRUM=$s" >> $O/ab_save_big.txt
    for shape in "262144 16 768 131072 384" "1048576 16 768 524288 96" "2097152 16 768 1048576 48" "2097152 16 768 2097152 48"; do
      if [ $s = default ]; then unset FFC_SAVE_SPECTRUM; else export FFC_SAVE_SPECTRUM=$s; fi
      timeout 600 python benchmarks/sweep.py row $shape 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print(r['row'][:36], 'H_run', r['H_run'], 'fwd', r['fwd_ms'], 'bwd', r['bwd_ms'], 'fwd+bwd', r['fwd_bwd_ms'], 'peak MB', round(r['peak_fwd_bwd'] / 1e6))
" >> $O/ab_save_big.txt
    done
  done
done
unset FFC_SAVE_SPECTRUM
cat $O/ab_save_big.txt
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json; cp gpurun_out/bench_full.json $O/bench_full.json
