# round-6 GPU call D: is PC sampling (or thread trace) available on this box?  Stall attribution for the two config-2 kernels.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_d; mkdir -p $O
cd /tmp
for k in bwd fwd; do
  timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit cycles --pc-sampling-method stochastic --pc-sampling-interval 1048576 --kernel-trace --output-format csv -d $O/pcs_$k -o p -- python $R/benchmarks/prof_step_kernels.py $k > $O/pcs_$k.log 2>&1
  tail -n 5 $O/pcs_$k.log
done
timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit time --pc-sampling-method host_trap --pc-sampling-interval 1 --kernel-trace --output-format csv -d $O/pch_bwd -o p -- python $R/benchmarks/prof_step_kernels.py bwd > $O/pch_bwd.log 2>&1
tail -n 5 $O/pch_bwd.log
timeout 200 rocprofv3 --att --kernel-include-regex bwd_kernel --output-format csv -d $O/att_bwd -o p -- python $R/benchmarks/prof_step_kernels.py bwd > $O/att_bwd.log 2>&1
tail -n 5 $O/att_bwd.log
find $O -type f | head -40; du -sh $O
# keep the merge small
find $O -type f -size +20M -delete
