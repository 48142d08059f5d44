# kernel-level breakdown of the big sizes (rocprofv3 --stats): cfg4 forward, fft 64K / 1M / 2M forward + backward
export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02c; mkdir -p $O
prof() { name=$1; shift; rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o s -- python $R/benchmarks/prof_one.py "$@" > $O/$name.log 2>&1; cp $O/$name/*/s_kernel_stats.csv $O/$name.csv 2>/dev/null || find $O/$name -name "*kernel_stats.csv" -exec cp {} $O/$name.csv \; ; }
prof cfg4_fwd 4194304 1 16 1048576 fwd
prof cfg4_both 4194304 1 16 1048576 both
prof f64k 65536 16 768 32768 both
prof f1m 1048576 16 96 524288 both
prof f2m 2097152 16 48 1048576 both
for f in cfg4_fwd cfg4_both f64k f1m f2m; do echo "== $f"; cut -d, -f1-4 $O/$f.csv | cut -c1-160 | head -14; done
