# A/B of the persistent double-buffered outer pass (BigBody::run_pipe; FFC_FLAGS bit 16 = pipelined; default = one block per workgroup as in round 3),
# module level fwd / bwd ms of HBM-level sizes.  Run through gpurun.
O=${1:-gpurun_out/bigpipe}; mkdir -p $O
for spec in "262144 16 768 131072 384" "1048576 16 768 524288 96" "2097152 16 768 1048576 48" "4194304 1 16 1048576" "1048576 8 48 1048576 48"; do
  for fl in 0 16 0 16; do
    FFC_FLAGS=$fl python benchmarks/sweep.py row $spec | sed "s/FFC_MULTIPASS=default/FFC_FLAGS=$fl/"
  done
done > $O/bigpipe.jsonl 2> $O/bigpipe.err
python - <<'PY' $O/bigpipe.jsonl
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        r = json.loads(l); print(f"{r['row']:60s} fwd {r['fwd_ms']:9.3f} bwd {r['bwd_ms']:9.3f} infer {r['fwd_infer_ms']:9.3f}")
PY
