export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01c; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q -n 2 > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log; tail -8 $O/pytest_gpu.log
python tests/prof_conv1d.py 2>&1 | grep -v amdgpu.ids | tee $O/conv1d.log
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/big64k -o b -- python $R/tests/prof_big.py 65536 > $O/big64k.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/big64k/b_kernel_stats.csv")))
for r in rows[:14]:
    print(r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e6, r["Percentage"])
PY
