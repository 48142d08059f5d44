# round-5 GPU call E (last): the measurement pass on the final code (benchmarks/measure_r05.sh: bench line, rocprofv3 stats / trace / PMC,
# configs, caller benches, the full GPU suite), the level passes once more against the per-element-twiddle variant, and the reference's
# two test files in FULL (every case, unmodified) against the final library
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_end
bash $R/benchmarks/measure_r05.sh
cd $R
V=$R/flash-fft-conv_amd/lib/variants
for v in nobigchain product; do
  if [ $v = product ]; then unset FFC_LIB; else export FFC_LIB=$V/$v/libflashfftconv_hip.so; fi
  echo "== $v" >> $O/level_bw.txt
  python benchmarks/level_bw.py 2>&1 | grep -v amdgpu.ids >> $O/level_bw.txt
done
unset FFC_LIB
cat $O/level_bw.txt
( time FFC_REF_TESTS_FULL=1 python -m pytest tests/test_reference_verbatim_gpu.py -m gpu -q -s ) > $O/reference_verbatim.log 2>&1; tail -8 $O/reference_verbatim.log
