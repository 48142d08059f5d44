# Round-1 measurement pass (run through gpurun from the repo root).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01_final; mkdir -p $O
cd $R
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 2500 $O/bench.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/stats.log 2>&1
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -o p -- python $R/benchmarks/prof_conv.py > $O/$name.log 2>&1; }
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
run p2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU
run p3 FETCH_SIZE
run p4 WRITE_SIZE
cd $R
python benchmarks/sweep.py all 2>&1 | grep -v amdgpu.ids > $O/sweep.jsonl
ls $O $O/stats | head -30
