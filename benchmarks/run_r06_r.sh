# round-6 GPU call R: parity of the level's wide form (opt-in FFC_BIG_WIDE=1): the dedicated test, then every 2M / 4M case of the module matrix with it on and off
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_r; mkdir -p $O
cd $R
( time python -m pytest tests/test_flashfftconv_gpu.py -m gpu -x -q -k "one_level_at_any_length" ) > $O/pytest_wide.txt 2>&1; tail -4 $O/pytest_wide.txt
( time FFC_BIG_WIDE=1 python -m pytest tests/test_flashfftconv_gpu.py tests/test_robustness_gpu.py tests/test_sharding_gpu.py tests/test_hyena_gpu.py -m gpu -x -q -k "2097152 or 4194304" ) > $O/pytest_big_wide.txt 2>&1; tail -4 $O/pytest_big_wide.txt
( time python -m pytest tests/test_flashfftconv_gpu.py tests/test_robustness_gpu.py tests/test_sharding_gpu.py tests/test_hyena_gpu.py -m gpu -x -q -k "2097152 or 4194304 or 1048576 or 524288" ) > $O/pytest_big_default.txt 2>&1; tail -4 $O/pytest_big_default.txt
