# round-6 GPU call Q: the wide form of the 128- / 64-point level (fft 4M / 2M in one level at any length): parity of the big sizes, then README-shaped rows
# with FFC_BIG_WIDE=0 (round-5 routing) / 1, interleaved twice
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_q; mkdir -p $O
cd $R
( time python -m pytest tests/test_flashfftconv_gpu.py tests/test_robustness_gpu.py tests/test_sharding_gpu.py tests/test_hyena_gpu.py -m gpu -x -q -k "2097152 or 4194304 or big or long or level" ) > $O/pytest_big.txt 2>&1; tail -4 $O/pytest_big.txt
row() { timeout 900 python benchmarks/sweep.py row $1 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print(r['row'][:34], 'H_run', r['H_run'], 'fwd', r['fwd_ms'], 'bwd', r['bwd_ms'], 'fwd+bwd', r['fwd_bwd_ms'], 'infer', r.get('fwd_infer_ms'), 'peak MB', round(r['peak_fwd_bwd'] / 1e6))
" >> $O/ab_wide.txt; }
for i in 1 2; do
  for wv in 0 1; do
    echo "== FFC_BIG_WIDE=$wv" >> $O/ab_wide.txt
    export FFC_BIG_WIDE=$wv
    row "2097152 8 768 2097152 32 gated"; row "4194304 8 768 4194304 16 gated"; row "4194304 8 768 2097152 16 gated"; row "2097152 16 768 2097152 48"; row "4194304 4 768 3000000 16"
  done
done
unset FFC_BIG_WIDE
cat $O/ab_wide.txt
