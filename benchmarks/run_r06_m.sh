# round-6 GPU call M: gated single-tile sizes keeping y_raw alone (FFC_Y_ONLY_MAX): parity tests, then module rows 0 (spectra + y_raw, round 5) / 1024 / 2048, interleaved twice
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_m; mkdir -p $O
cd $R
( time python -m pytest tests/test_spectrum_gpu.py -m gpu -x -q ) > $O/pytest_spectrum.txt 2>&1; tail -4 $O/pytest_spectrum.txt
( time FFC_Y_ONLY_MAX=2048 python -m pytest tests/test_spectrum_gpu.py tests/test_flashfftconv_gpu.py -m gpu -x -q -k "2048 or 1024 or 256 or 512" ) > $O/pytest_y2048.txt 2>&1; tail -4 $O/pytest_y2048.txt
for i in 1 2; do
  for s in 0 1024 2048; do
    echo "== FFC_Y_ONLY_MAX=$s" >> $O/ab_y_only.txt
    for shape in "256 64 768 256 768 gated" "512 64 768 512 768 gated" "1024 64 768 1024 768 gated" "1024 16 768 512 768 gated" "2048 64 768 2048 768 gated" "2048 16 768 1024 768 gated"; do
      FFC_Y_ONLY_MAX=$s timeout 300 python benchmarks/sweep.py row $shape 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print(r['row'][:30], 'gated' if r['gated'] else 'plain', 'fwd', r['fwd_ms'], 'bwd', r['bwd_ms'], 'fwd+bwd', r['fwd_bwd_ms'], 'infer', r.get('fwd_infer_ms'), 'peak MB', round(r['peak_fwd_bwd'] / 1e6))
" >> $O/ab_y_only.txt
    done
  done
done
cat $O/ab_y_only.txt
