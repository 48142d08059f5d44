# Routing A/B of fft 65536 / 131072: R passes of the fused 32768 kernel (default) against one HBM level around
# a fused inner size (FFC_MULTIPASS without the size), module level incl. saved spectra, fwd / bwd ms at B16 H768.  Run through gpurun.
O=${1:-gpurun_out/route}; mkdir -p $O
for spec in "131072 16 768 65536" "131072 16 384 131072" "65536 16 768 32768" "65536 16 768 65536" "131072 16 768 32768"; do
  set -- $spec
  for mp in "2048,65536,131072" "2048,65536" "2048,131072" "2048"; do
    FFC_MULTIPASS=$mp python benchmarks/sweep.py row $1 $2 $3 $4
  done
done > $O/route.jsonl 2> $O/route.err
python - <<'PY' $O/route.jsonl
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        r = json.loads(l); print(f"{r['row']:70s} fwd {r['fwd_ms']:8.3f} bwd {r['bwd_ms']:8.3f} sum {r['fwd_bwd_ms']:8.3f} infer {r['fwd_infer_ms']:8.3f}  peak {r['peak_fwd_bwd']/1e9:.2f} GB")
PY
