export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_r01; mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
grep -oE "\b(SQ_[A-Z_0-9]+|TCC_[A-Z_0-9]+|FETCH_SIZE|WRITE_SIZE|GRBM_[A-Z_]+|MfmaUtil|VALUBusy|TCP_[A-Z_0-9]+)\b" $O/counters.txt | sort -u | tr '\n' ' ' | head -c 6000 > $O/counter_names.txt
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -o p -- python $R/benchmarks/prof_conv.py > $O/$name.log 2>&1; tail -2 $O/$name.log | cut -c1-300; }
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
run p2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU
run p3 FETCH_SIZE
run p4 WRITE_SIZE
run p5 SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM GRBM_GUI_ACTIVE SQ_ACTIVE_INST_MISC
find $O -name "*.csv" | head -20
