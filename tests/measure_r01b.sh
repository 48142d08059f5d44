# Round-1 re-entry measurement pass (run through gpurun from the repo root): GPU tests, bench, kernel stats, sweep.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01b; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q -n 2 > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 2500 $O/bench.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/stats.log 2>&1
cd $R
python benchmarks/sweep.py all 2>&1 | grep -v amdgpu.ids > $O/sweep.jsonl
cat $O/sweep.jsonl | cut -c1-260
