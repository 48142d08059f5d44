export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01d; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q -n 2 > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log; tail -12 $O/pytest_gpu.log | cut -c1-250
