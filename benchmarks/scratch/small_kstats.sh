# kernel durations (rocprofv3 kernel stats) of one fwd+bwd at the small fft sizes: B16 H768 L=N/2
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/small; cd /tmp; export TMPDIR=/tmp
for N in ${SIZES:-512 1024 2048}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p$N -o s -- python $GRAFT_REPO_ROOT/benchmarks/prof_one.py $N 16 768 $((N/2)) both > /dev/null 2>&1
  cp $(find /tmp/p$N -name "*kernel_stats.csv") $GRAFT_REPO_ROOT/gpurun_out/small/s$N.csv
done
