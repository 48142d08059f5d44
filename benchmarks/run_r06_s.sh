# round-6 GPU call S: same-box check that the config-2 (and neighbouring) kernels of the final library run as the library of call K did (the bench boxes differ by +-3 %)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_s; mkdir -p $O
cd $R
for i in 1 2 3; do
  echo "== final library" >> $O/ab_final_vs_k.txt
  timeout 300 python benchmarks/ab_lib.py 32768,16,768,16384 32768,16,768,32768 16384,8,1024,8192,g 2>&1 | grep -v amdgpu.ids >> $O/ab_final_vs_k.txt
  echo "== library of call K (commit a738eb6)" >> $O/ab_final_vs_k.txt
  FFC_LIB=$R/flash-fft-conv_amd/lib/variants/prevK/libflashfftconv_hip.so timeout 300 python benchmarks/ab_lib.py 32768,16,768,16384 32768,16,768,32768 16384,8,1024,8192,g 2>&1 | grep -v amdgpu.ids >> $O/ab_final_vs_k.txt
done
cat $O/ab_final_vs_k.txt
