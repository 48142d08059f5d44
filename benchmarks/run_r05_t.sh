# round-5 GPU call T: the bench line once more (call S's rows had lost the fft_run field to bench.py's row filter)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_end; mkdir -p $O
cd $R
timeout 120 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 200 $O/bench.json; cp gpurun_out/bench_full.json $O/bench_full.json
