export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01f; mkdir -p $O
cd $R
echo "== driver-style single process"; timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_single.log 2>&1; tail -3 $O/pytest_single.log | cut -c1-200
for i in 1 2; do
echo "== xdist -n 2 run $i"; timeout 600 python -m pytest tests -m gpu -q -n 2 > $O/pytest_x$i.log 2>&1
grep -E "^FAILED|passed|failed|not reproducible|never reproduced" $O/pytest_x$i.log | cut -c1-220 | tail -12
done
