# round-6 GPU call K: the library after the switch clean-up: full GPU suite + bench line (+ same-box A/B digest run of ab_lib for the record)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_k; mkdir -p $O
cd $R
( time python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json; cp gpurun_out/bench_full.json $O/bench_full.json
timeout 600 python benchmarks/ab_lib.py 2>&1 | grep -v amdgpu.ids > $O/ab_lib.txt; cat $O/ab_lib.txt
