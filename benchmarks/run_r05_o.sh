# round-5 GPU call O: folded outer twiddle in the cross-unit forward of fft 16384 (lib/variants/fold, -DFFC_FOLD_TW=1: two pairs share a tile's four
# matrices = half the matrix bytes per pair of fft 32768) against the product, same box, interleaved three times (benchmarks/ab_lib.py)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_o; mkdir -p $O
cd $R
V=$R/flash-fft-conv_amd/lib/variants
for i in 1 2 3; do
  for v in product fold; do
    if [ $v = product ]; then unset FFC_LIB; else export FFC_LIB=$V/$v/libflashfftconv_hip.so; fi
    echo "== $v" >> $O/ab_fold16k.txt
    python benchmarks/ab_lib.py 16384,16,768,8192 16384,16,768,16384 16384,8,1024,8192,g 2>&1 | grep -v amdgpu.ids >> $O/ab_fold16k.txt
  done
done
( FFC_LIB=$V/fold/libflashfftconv_hip.so python -m pytest tests/test_flashfftconv_gpu.py -m gpu -x -q -k "16384" ) 2>&1 | tail -2
cat $O/ab_fold16k.txt
