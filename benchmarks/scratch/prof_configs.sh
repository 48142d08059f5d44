# rocprofv3 kernel stats of the BASELINE configs (benchmarks/sweep.py configs) and of one big size (fft 65536)
export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01_cfg; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg -o c -- python $R/benchmarks/sweep.py configs > $O/cfg.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/big -o b -- python $R/benchmarks/prof_big.py 65536 > $O/big.log 2>&1
ls $O/cfg $O/big
