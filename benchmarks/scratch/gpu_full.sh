# full -m gpu suite (timed) + smoke + the default bench line
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/full; mkdir -p $O
cd $R
( time timeout 1700 python -m pytest tests -m gpu -q -x -p no:cacheprovider ) > $O/pytest.log 2>&1
tail -6 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
( time python bench.py ) > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["traffic_profiled"])
PY
tail -3 $O/bench.err
