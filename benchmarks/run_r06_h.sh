# round-6 GPU call H: fft 2048 -- merged passes in the forward (FFC_IP_MERGE) + the pass tables read from LDS at their use in the backward (FFC_IP_LEAN: no spills)
# against the round-5 forms (lib/variants/ipold), same box, interleaved; module-level rows (sweep.py row) + kernel-level (ab_lib.py); parity of the fft-2048 cases
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_h; mkdir -p $O
cd $R
V=$R/flash-fft-conv_amd/lib/variants
for i in 1 2 3; do
  for v in product ipold; do
    if [ $v = product ]; then unset FFC_LIB; else export FFC_LIB=$V/$v/libflashfftconv_hip.so; fi
    echo "== $v" >> $O/ab_2048.txt
    timeout 600 python benchmarks/ab_lib.py 2048,16,768,1024 2048,16,768,1024,g 2048,64,768,1024,g 2048,16,768,2048 1024,16,768,1024 2>&1 | grep -v amdgpu.ids | grep -v library | sed 's/digests.*//' >> $O/ab_2048.txt
    for shape in "2048 16 768 1024" "2048 16 768 1024 768 g"; do
      timeout 300 python benchmarks/sweep.py row $shape 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('module', r['row'][:34], 'gated' if r['gated'] else 'plain', 'fwd', r['fwd_ms'], 'bwd', r['bwd_ms'], 'infer', r['fwd_infer_ms'], 'graph', r.get('graph_step_ms'))
" >> $O/ab_2048.txt
    done
  done
done
unset FFC_LIB
cat $O/ab_2048.txt
( time timeout 900 python -m pytest tests/test_flashfftconv_gpu.py tests/test_spectrum_gpu.py tests/test_graph_gpu.py tests/test_determinism_gpu.py -m gpu -x -q -k "2048 or 1024" ) > $O/pytest_2048.txt 2>&1; tail -3 $O/pytest_2048.txt
