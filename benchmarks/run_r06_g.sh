# round-6 GPU call G: Body::conv_small without its scratch round trips against the round-5 job loop (lib/variants/nopipe), interleaved
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_g; mkdir -p $O
cd $R
V=$R/flash-fft-conv_amd/lib/variants
for i in 1 2 3; do
  for v in product nopipe; do
    if [ $v = product ]; then unset FFC_LIB; else export FFC_LIB=$V/$v/libflashfftconv_hip.so; fi
    echo "== $v" >> $O/ab_small.txt
    timeout 600 python benchmarks/ab_lib.py 1024,64,768,1024,g 1024,16,768,1024 1024,16,768,512 256,64,768,256,g 512,64,768,512,g 512,16,768,256 256,16,768,128 2>&1 | grep -v amdgpu.ids | grep -v library | sed 's/digests.*//' >> $O/ab_small.txt
  done
done
unset FFC_LIB
cat $O/ab_small.txt
