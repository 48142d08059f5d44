# round-6 GPU call C: dk_f accumulated by the matrix pipe (FFC_WACC_MFMA = 1, the new default) against the VALU form (lib/variants/wacc0), same box,
# interleaved three times; then parity on the new library (every backward test of the fused sizes, spectra, determinism, robustness), the RCCL
# world-size-1 test and the two-rank bench line
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_c; mkdir -p $O
cd $R
V=$R/flash-fft-conv_amd/lib/variants
for i in 1 2 3; do
  for v in product wacc0; do
    if [ $v = product ]; then unset FFC_LIB; else export FFC_LIB=$V/$v/libflashfftconv_hip.so; fi
    echo "== $v" >> $O/ab_wacc.txt
    timeout 600 python benchmarks/ab_lib.py 32768,16,768,16384 32768,16,768,32768 16384,16,768,8192 65536,16,768,32768 8192,16,768,4096 4096,16,768,2048 16384,8,1024,8192,g 2>&1 | grep -v amdgpu.ids | sed 's/digests.*//' >> $O/ab_wacc.txt
  done
done
unset FFC_LIB
cat $O/ab_wacc.txt
( time timeout 1500 python -m pytest tests/test_flashfftconv_gpu.py tests/test_spectrum_gpu.py tests/test_determinism_gpu.py tests/test_hyena_gpu.py tests/test_graph_gpu.py -m gpu -x -q -n 4 ) > $O/pytest_conv.txt 2>&1; tail -5 $O/pytest_conv.txt
( time timeout 900 python -m pytest tests/test_rccl_gpu.py tests/test_sharding_gpu.py -m gpu -x -q ) > $O/pytest_shard.txt 2>&1; tail -15 $O/pytest_shard.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
