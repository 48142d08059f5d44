# round-6 GPU call N: routing of the README rows N = 2M / 4M at L = N (gated, B8): default against the other factorisations the library has
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_n; mkdir -p $O
cd $R
row() { timeout 600 python benchmarks/sweep.py row $1 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print(r['row'][:34], 'H_run', r['H_run'], 'fwd', r['fwd_ms'], 'bwd', r['bwd_ms'], 'fwd+bwd', r['fwd_bwd_ms'], 'infer', r.get('fwd_infer_ms'))
" >> $O/ab_route_big.txt; }
for i in 1 2; do
  echo "== default (2M: 32 x 65536 [2-pass inner]; 4M: 16 x 16 x 16384)" >> $O/ab_route_big.txt
  row "2097152 8 768 2097152 32 gated"; row "4194304 8 768 4194304 16 gated"
  echo "== FFC_BIG_2LEVEL=1 (2M: 16 x 16 x 8192)" >> $O/ab_route_big.txt
  FFC_BIG_2LEVEL=1 row "2097152 8 768 2097152 32 gated"
  echo "== FFC_BIG_1LEVEL=1 (4M: 32 x 131072 [4-pass inner])" >> $O/ab_route_big.txt
  FFC_BIG_1LEVEL=1 row "4194304 8 768 4194304 16 gated"
done
cat $O/ab_route_big.txt
