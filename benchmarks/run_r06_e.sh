# round-6 GPU call E: a batch of one through the HBM-level sizes keeps half of the level rows (real-input symmetry, csrc/ffc_big.h BigArgs::half).
# Parity: every B = 1 case of the reference's matrix + the config-4 / fitted-size tests + the Hyena operator; timing: FFC_BIG_HALF=0 / 1 interleaved.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_e; mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests/test_flashfftconv_gpu.py -m gpu -x -q -k "1-111 or 1-768 or cfg4 or fitted or golden or odd" ) > $O/pytest_b1.txt 2>&1; tail -4 $O/pytest_b1.txt
( time timeout 600 python -m pytest tests/test_hyena_gpu.py tests/test_spectrum_gpu.py -m gpu -x -q ) > $O/pytest_hy.txt 2>&1; tail -4 $O/pytest_hy.txt
for i in 1 2 3; do
  for h in 0 1; do
    echo "== FFC_BIG_HALF=$h" >> $O/ab_half.txt
    for shape in "4194304 1 16 1048576" "2097152 1 32 1048576" "1048576 1 48 524288" "1048576 1 48 1048576" "262144 1 192 131072"; do
      FFC_BIG_HALF=$h timeout 300 python benchmarks/sweep.py row $shape 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print(r['row'][:40], 'fft_run', r.get('fft_run'), 'fwd', r['fwd_ms'], 'bwd', r['bwd_ms'], 'infer', r['fwd_infer_ms'], 'seqlen-point', r.get('fwd_ms_seqlen_points'), r.get('bwd_ms_seqlen_points'))
" >> $O/ab_half.txt
    done
  done
done
cat $O/ab_half.txt
