# round-5 GPU call J: what the folded-twiddle kernels would cost WITHOUT fetching their matrices (lib/variants/foldko: -DFFC_FOLD_TW=1 -DFFC_KO=2048,
# results wrong by design) next to the real variant (fold) and the product: separates "fewer VALU instructions" from "more L2 loads"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_j; mkdir -p $O
cd $R
V=$R/flash-fft-conv_amd/lib/variants
for i in 1 2; do
  for v in product fold foldko; do
    if [ $v = product ]; then unset FFC_LIB; else export FFC_LIB=$V/$v/libflashfftconv_hip.so; fi
    echo "== $v" >> $O/ab_foldko.txt
    python benchmarks/ab_lib.py 32768,16,768,16384 32768,16,768,32768 2>&1 | grep -v amdgpu.ids >> $O/ab_foldko.txt
  done
done
unset FFC_LIB
cat $O/ab_foldko.txt
