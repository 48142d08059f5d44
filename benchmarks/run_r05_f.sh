# round-5 GPU call F: spectrum-tile prefetch in the saved-spectra backward of fft 32768 (lib/variants/zpf: -DFFC_Z_PREFETCH=1) against the
# product, same box, interleaved three times (benchmarks/ab_lib.py: min of 3 x 20 launches)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_f; mkdir -p $O
cd $R
V=$R/flash-fft-conv_amd/lib/variants
for i in 1 2 3; do
  for v in product zpf; do
    if [ $v = product ]; then unset FFC_LIB; else export FFC_LIB=$V/$v/libflashfftconv_hip.so; fi
    echo "== $v" >> $O/ab_zpf.txt
    python benchmarks/ab_lib.py 32768,16,768,16384 32768,16,768,32768 32768,16,768,16384,g 2>&1 | grep -v amdgpu.ids >> $O/ab_zpf.txt
  done
done
unset FFC_LIB
cat $O/ab_zpf.txt
