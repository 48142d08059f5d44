# round-5 GPU call B: parity subset on the new kernels (everything but the verbatim reference files and the 2-rank tests), the graphed
# step, then A/B of the round-5 kernel changes on one box: round-4 kernels (lib/variants/kf_late: FFC_KF_LATE measured as no-op) vs the
# product, and the product with one change switched off (nochain: -DFFC_CHAIN16=0, noquad: -DFFC_OUTER_QUAD=0); multi-pass rows at module
# level (fast-only backward + dk tail per pass); the bench line
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_b; mkdir -p $O
cd $R
( time python -m pytest tests -m gpu -x -q --deselect tests/test_reference_verbatim_gpu.py --deselect tests/test_sharding_gpu.py ) > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
python -m pytest tests/test_graph_gpu.py -m gpu -q -s 2>&1 | grep -E "eager|passed|failed" > $O/graph.txt; cat $O/graph.txt
V=$R/flash-fft-conv_amd/lib/variants
SH="32768,16,768,16384 32768,16,768,32768 16384,16,768,8192 8192,16,768,4096 4096,16,768,2048 16384,8,1024,8192,g"
for i in 1 2; do
  for v in kf_late product nochain noquad; do
    if [ $v = product ]; then unset FFC_LIB; else export FFC_LIB=$V/$v/libflashfftconv_hip.so; fi
    [ $v = product ] || [ -f $FFC_LIB ] || continue
    echo "== $v" >> $O/ab_kernels.txt
    python benchmarks/ab_lib.py $SH 2>&1 | grep -v amdgpu.ids >> $O/ab_kernels.txt
  done
done
unset FFC_LIB
cat $O/ab_kernels.txt
for v in kf_late product; do
  if [ $v = product ]; then unset FFC_LIB; else export FFC_LIB=$V/$v/libflashfftconv_hip.so; fi
  echo "== $v" >> $O/ab_multipass.txt
  python benchmarks/sweep.py row 65536 16 768 32768 2>&1 | grep -v amdgpu.ids >> $O/ab_multipass.txt
  python benchmarks/sweep.py row 65536 16 768 65536 2>&1 | grep -v amdgpu.ids >> $O/ab_multipass.txt
  python benchmarks/sweep.py row 131072 16 768 65536 2>&1 | grep -v amdgpu.ids >> $O/ab_multipass.txt
  python benchmarks/sweep.py row 65536 16 768 32768 768 g 2>&1 | grep -v amdgpu.ids >> $O/ab_multipass.txt
done
unset FFC_LIB
cut -c1-400 $O/ab_multipass.txt
python bench.py > $O/bench.txt 2> $O/bench.err; tail -c 2800 $O/bench.txt; cp gpurun_out/bench_full.json $O/ 2>/dev/null
