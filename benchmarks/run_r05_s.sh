# round-5 GPU call S (final code): the bench line, rocprofv3 kernel stats of configs 4 and 3 (config 4 runs 2097152 points since call R, the forward of
# fft 16384 folds its outer twiddle since call O), then the GPU tests of every fft size above 32768 as far as the remaining GPU time allows
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_end; mkdir -p $O
cd $R
timeout 150 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json; cp gpurun_out/bench_full.json $O/bench_full.json
cd /tmp
timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg4_s -o s -- python $R/benchmarks/prof_one.py 4194304 1 16 1048576 both > $O/cfg4.log 2>&1
timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg3_s -o s -- python $R/benchmarks/prof_one.py 16384 8 1024 8192 both gated > $O/cfg3.log 2>&1
cd $R
( timeout 100 python -m pytest tests/test_flashfftconv_gpu.py tests/test_hyena_gpu.py -m gpu -x -q -k "65536 or 131072 or 262144 or 524288 or 1048576 or 2097152 or 4194304" ) > $O/pytest_gpu_s.txt 2>&1; tail -2 $O/pytest_gpu_s.txt
