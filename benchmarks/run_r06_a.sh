# round-6 GPU call A: what bounds the two headline kernels (config 2: fft 32768 B16 H768 L16384 bf16)?
#  1. knock-out builds (lib/variants/koN: -DFFC_KO=N, wrong results by design), interleaved twice with the product:
#     ko4 = no k_f loads, ko64 = no spectrum traffic, ko2 = no row loads / stores, ko70 = none of the three, ko1 = no twiddle multiplies,
#     ko71 = neither (pure MFMA + LDS + the remaining VALU)
#  2. per-phase cycle budget of the forward (conv_prof_kernel) and of the backward on saved spectra (bwdprof variant)
#  3. L2 policy: fabric reads / L2 hits per policy (k_f nt: FFC_FLAGS=2; stores sc1: variant sc1st)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_a; mkdir -p $O
cd $R
V=$R/flash-fft-conv_amd/lib/variants
rocprofv3 -L > $O/counters.txt 2>&1
for i in 1 2; do
  for v in product ko4 ko64 ko2 ko70 ko1 ko71 sc1st; do
    if [ $v = product ]; then unset FFC_LIB; else export FFC_LIB=$V/$v/libflashfftconv_hip.so; fi
    echo "== $v" >> $O/ab_ko.txt
    timeout 300 python benchmarks/ab_lib.py 32768,16,768,16384 2>&1 | grep -v amdgpu.ids >> $O/ab_ko.txt
  done
done
unset FFC_LIB
for f in 2 4 6; do
  echo "== product FFC_FLAGS=$f" >> $O/ab_ko.txt
  FFC_FLAGS=$f timeout 300 python benchmarks/ab_lib.py 32768,16,768,16384 2>&1 | grep -v amdgpu.ids >> $O/ab_ko.txt
done
echo "== sc1st FFC_FLAGS=2" >> $O/ab_ko.txt
FFC_LIB=$V/sc1st/libflashfftconv_hip.so FFC_FLAGS=2 timeout 300 python benchmarks/ab_lib.py 32768,16,768,16384 2>&1 | grep -v amdgpu.ids >> $O/ab_ko.txt
cat $O/ab_ko.txt
timeout 300 python benchmarks/prof_phases.py 2>&1 | grep -v amdgpu.ids > $O/phases_fwd.txt; cat $O/phases_fwd.txt
FFC_LIB=$V/bwdprof/libflashfftconv_hip.so timeout 300 python benchmarks/prof_bwdz_phases.py 2>&1 | grep -v amdgpu.ids > $O/phases_bwdz.txt; cat $O/phases_bwdz.txt
cd /tmp
pmc() { name=$1; arg=$2; shift 2; timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -o p -- python $R/benchmarks/prof_step_kernels.py $arg > $O/$name.log 2>&1; }
for pol in p0 p2 s0 s2; do
  case $pol in
    p0) unset FFC_LIB; export FFC_FLAGS=0;;
    p2) unset FFC_LIB; export FFC_FLAGS=2;;
    s0) export FFC_LIB=$V/sc1st/libflashfftconv_hip.so; export FFC_FLAGS=0;;
    s2) export FFC_LIB=$V/sc1st/libflashfftconv_hip.so; export FFC_FLAGS=2;;
  esac
  for k in fwd bwd; do
    pmc ${pol}_${k}_f $k FETCH_SIZE
    pmc ${pol}_${k}_h $k TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
    pmc ${pol}_${k}_e $k TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
  done
done
unset FFC_LIB FFC_FLAGS
cd $R
python - <<'PY'
import csv, glob, os, collections
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r06_a")
for d in sorted(glob.glob(O + "/[ps][0-9]_*_[fhe]")):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(os.path.basename(d), "no counters:", open(d + ".log").read()[-300:].replace("\n", " | ")); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "conv_kernel" in r["Kernel_Name"] or "bwd_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(os.path.basename(d), {k: f"{sum(v)/len(v):.4e}" for k, v in acc.items()})
PY
