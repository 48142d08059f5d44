export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02b; mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=60 --durations=15 -p no:cacheprovider --deselect tests/test_conv1d_gpu.py::test_conv1d_fwd ) > $O/pytest.log 2>&1
tail -70 $O/pytest.log
