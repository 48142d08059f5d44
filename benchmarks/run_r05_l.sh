# round-5 GPU call L: fp16 gate multiplies as one v_pk_mul_f16 (product) against the widening form (lib/variants/nopkgate: -DFFC_PK_GATE=0) on the
# reference's README shapes (gated fp16, L = N; benchmarks/sweep.py readme: training forward + backward, scaled to B64 x H768), same box, twice
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_l; mkdir -p $O
cd $R
V=$R/flash-fft-conv_amd/lib/variants
for i in 1 2; do
  for v in nopkgate product; do
    if [ $v = product ]; then unset FFC_LIB; else export FFC_LIB=$V/$v/libflashfftconv_hip.so; fi
    echo "== $v" >> $O/ab_pkgate.txt
    python benchmarks/sweep.py readme 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['fft'], d['fwd_ms_scaled_to_B64_H768'], d['bwd_ms_scaled'], d['speedup_vs_h100_published'])
" >> $O/ab_pkgate.txt
  done
done
unset FFC_LIB
cat $O/ab_pkgate.txt
FFC_LIB= python -m pytest tests/test_flashfftconv_gpu.py -m gpu -q -x -k "float16 or fp16 or dtype1 or unit_scale" 2>&1 | tail -2
