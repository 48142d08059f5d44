export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_bwd; mkdir -p $O
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -o p -- python $R/benchmarks/prof_bwd.py > $O/$name.log 2>&1; }
run p3 FETCH_SIZE
run p4 WRITE_SIZE
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
python - <<PY
import csv, collections
out = {}
for d in ("p1","p3","p4"):
    rows = list(csv.DictReader(open("$O/%s/p_counter_collection.csv" % d)))
    acc = collections.defaultdict(list)
    for r in rows:
        if "bwd_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items(): out[k] = sum(v)/len(v)
for k, v in out.items(): print("%-28s%.4e" % (k, v))
PY
