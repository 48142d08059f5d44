# round-5 GPU call P (last): the GPU suite without the verbatim reference files (full verbatim run: call E; the fft-16384 forward kernels are the only
# change since call M) on the final library, then the bench line
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_end; mkdir -p $O
cd $R
( time python -m pytest tests -m gpu -x -q --deselect tests/test_reference_verbatim_gpu.py ) > $O/pytest_gpu_p.txt 2>&1; tail -4 $O/pytest_gpu_p.txt
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json; cp gpurun_out/bench_full.json $O/bench_full.json
