# round-5 GPU call N: the bench line once more (bench.py now also times the short sweep rows as one HIP graph); the other r05_end profiles stay (same library)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_end; mkdir -p $O
cd $R
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json; cp gpurun_out/bench_full.json $O/bench_full.json
