# Round-6 measurement pass (run through gpurun from the repo root): the bench line (with the embedded sweep / configs / README
# table / cpu baseline / measured peaks), rocprofv3 kernel stats + kernel trace of the same command, PMC passes (separate runs
# per counter group, only --kernel-trace next to --pmc) for the two big kernels of the default step, kernel stats of the
# other configs, the caller benches and the short-sequence probe.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_end; mkdir -p $O
cd $R
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json; cp gpurun_out/bench_full.json $O/bench_full.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sweep > $O/stats.log 2>&1
pmc() { arg=$1; name=$2; shift 2; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -o p -- python $R/benchmarks/prof_step_kernels.py $arg > $O/$name.log 2>&1; }
pmc fwd c1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
pmc fwd c2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU
pmc fwd c3 FETCH_SIZE
pmc fwd c4 WRITE_SIZE
pmc bwd b1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
pmc bwd b2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU
pmc bwd b3 FETCH_SIZE
pmc bwd b4 WRITE_SIZE
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg4 -o s -- python $R/benchmarks/prof_one.py 4194304 1 16 1048576 both > $O/cfg4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg3 -o s -- python $R/benchmarks/prof_one.py 16384 8 1024 8192 both gated > $O/cfg3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/f1k -o s -- python $R/benchmarks/prof_one.py 2048 16 768 1024 both > $O/f1k.log 2>&1
cd $R
python benchmarks/hyena_dna_fwd.py --train tiny-16k small-32k hyena-pile-4k > $O/hyena_train.jsonl 2> $O/hyena_train.err
python benchmarks/m2_bert_fwd.py > $O/m2_bert_fwd.jsonl 2> $O/m2_bert_fwd.err
python benchmarks/short_probe.py > $O/short_probe.txt 2>&1
find $O -name "*_kernel_stats.csv" | head; find $O -name "*counter_collection.csv" | head -3
# the full GPU suite on the same (final) code: the driver's round-end command
cd $R
( time python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
# the reference's two test files in FULL, unmodified
( time FFC_REF_TESTS_FULL=1 python -m pytest tests/test_reference_verbatim_gpu.py -m gpu -x -q -s ) > $O/reference_verbatim.log 2>&1; tail -6 $O/reference_verbatim.log
