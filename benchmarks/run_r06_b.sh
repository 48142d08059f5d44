# round-6 GPU call B: (1) VALU issue rates of candidate replacements for the half-rate unpack / merge instructions (benchmarks/ubench/valu_rates.hip);
# (2) which side flickers under 4-way GPU time-slicing, torch.fft or the HIP module (benchmarks/reffft_contention.py)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_b; mkdir -p $O
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 benchmarks/ubench/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates > $O/valu_rates.txt 2>&1; cat $O/valu_rates.txt
timeout 1500 python benchmarks/reffft_contention.py 200 4 > $O/reffft_contention.txt 2>&1; tail -n 8 $O/reffft_contention.txt
