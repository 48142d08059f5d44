# round-5 GPU call H: outer twiddle folded into per-tile inner matrices (lib/variants/fold: -DFFC_FOLD_TW=1), saved-spectra backward of
# fft 32768: parity of the variant through the GPU tests of that size, then the A/B against the product (same box, interleaved)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_h; mkdir -p $O
cd $R
V=$R/flash-fft-conv_amd/lib/variants
( FFC_LIB=$V/fold/libflashfftconv_hip.so python -m pytest tests/test_flashfftconv_gpu.py tests/test_spectrum_gpu.py -m gpu -x -q -k "32768 and not equals_recompute" ) > $O/pytest_fold.txt 2>&1; tail -3 $O/pytest_fold.txt
for i in 1 2 3; do
  for v in product fold; do
    if [ $v = product ]; then unset FFC_LIB; else export FFC_LIB=$V/$v/libflashfftconv_hip.so; fi
    echo "== $v" >> $O/ab_fold.txt
    python benchmarks/ab_lib.py 32768,16,768,16384 32768,16,768,32768 32768,16,768,16384,g 2>&1 | grep -v amdgpu.ids >> $O/ab_fold.txt
  done
done
unset FFC_LIB
cat $O/ab_fold.txt
