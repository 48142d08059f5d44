# round-6 GPU call O: config 3 (gated fft 16384, B8 H1024 L8192): the in-launch k -> k_f head / dk tail against separate launches (FFC_FLAGS 64 / 32), interleaved
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_o; mkdir -p $O
cd $R
row() { timeout 600 python benchmarks/sweep.py row $1 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print(r['row'][:34], 'fwd', r['fwd_ms'], r['fwd_ms_min'], 'bwd', r['bwd_ms'], r['bwd_ms_min'], 'fwd+bwd', r['fwd_bwd_ms'], 'infer', r.get('fwd_infer_ms'))
" >> $O/ab_cfg3_flags.txt; }
for i in 1 2 3; do
  for f in 0 64 32 96; do
    echo "== FFC_FLAGS=$f" >> $O/ab_cfg3_flags.txt
    FFC_FLAGS=$f row "16384 8 1024 8192 1024 gated"
    FFC_FLAGS=$f row "8192 16 768 4096 768"
  done
done
cat $O/ab_cfg3_flags.txt
