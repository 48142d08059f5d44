# quick GPU check used while tuning: FFT-conv parity + determinism + robustness tests, then the sweep rows
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/quick; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log | cut -c1-200
python benchmarks/sweep.py ${1:-all} 2>&1 | grep -v amdgpu.ids > $O/sweep.jsonl
python - <<PY
import json
for l in open("$O/sweep.jsonl"):
    r = json.loads(l); print(r["row"], r.get("fwd_ms"), r.get("bwd_ms"))
PY
