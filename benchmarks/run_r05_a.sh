# round-5 GPU call A: the GPU suite, the bench line on the round-4 kernels (first timing of the four late load-in-flight fixes) and the
# FFC_KF_LATE A/B (lib/variants/kf_late)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_a; mkdir -p $O
cd $R
( time python -m pytest tests -m gpu -x -q ) > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
python bench.py > $O/bench.txt 2> $O/bench.err; tail -c 2500 $O/bench.txt; cp gpurun_out/bench_full.json $O/ 2>/dev/null
for i in 1 2; do
  python benchmarks/ab_lib.py 32768,16,768,16384 32768,16,768,32768 16384,16,768,8192 >> $O/ab_kf_late.txt 2>&1
  FFC_LIB=$R/flash-fft-conv_amd/lib/variants/kf_late/libflashfftconv_hip.so python benchmarks/ab_lib.py 32768,16,768,16384 32768,16,768,32768 16384,16,768,8192 >> $O/ab_kf_late.txt 2>&1
done
cat $O/ab_kf_late.txt
