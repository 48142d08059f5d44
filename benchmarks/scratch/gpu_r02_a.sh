# Round-2 first GPU pass: the whole -m gpu suite (timed) + the bench line with the embedded sweep.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02a; mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 --durations=25 -p no:cacheprovider ) > $O/pytest.log 2>&1
tail -40 $O/pytest.log
( time python bench.py ) > $O/bench.json 2> $O/bench.err
tail -c 1500 $O/bench.json; tail -5 $O/bench.err
