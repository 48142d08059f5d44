# round-5 GPU call C: A/B of the level passes (product vs lib/variants/nobigchain = -DFFC_BIG_CHAIN=0 in the level unit), the parity suite
# without the verbatim / 2-rank files, the graphed-step figures, sweep rows of the HBM-level sizes and the bench line
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_c; mkdir -p $O
cd $R
V=$R/flash-fft-conv_amd/lib/variants
for i in 1 2; do
  for v in nobigchain product; do
    if [ $v = product ]; then unset FFC_LIB; else export FFC_LIB=$V/$v/libflashfftconv_hip.so; fi
    echo "== $v" >> $O/level_bw.txt
    python benchmarks/level_bw.py 2>&1 | grep -v amdgpu.ids >> $O/level_bw.txt
  done
done
unset FFC_LIB
cat $O/level_bw.txt
( time python -m pytest tests -m gpu -x -q --deselect tests/test_reference_verbatim_gpu.py --deselect tests/test_sharding_gpu.py ) > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
python -m pytest tests/test_graph_gpu.py -m gpu -q -s 2>&1 | grep -E "eager|passed|failed" > $O/graph.txt; cat $O/graph.txt
for v in nobigchain product; do
  if [ $v = product ]; then unset FFC_LIB; else export FFC_LIB=$V/$v/libflashfftconv_hip.so; fi
  echo "== $v" >> $O/ab_big.txt
  python benchmarks/sweep.py row 262144 16 768 131072 384 2>&1 | grep -v amdgpu.ids >> $O/ab_big.txt
  python benchmarks/sweep.py row 1048576 16 768 524288 96 2>&1 | grep -v amdgpu.ids >> $O/ab_big.txt
  python benchmarks/sweep.py row 4194304 1 16 1048576 2>&1 | grep -v amdgpu.ids >> $O/ab_big.txt
done
unset FFC_LIB
cut -c1-330 $O/ab_big.txt
python bench.py > $O/bench.txt 2> $O/bench.err; tail -c 2800 $O/bench.txt; cp gpurun_out/bench_full.json $O/ 2>/dev/null
