"""bench.py's contract line: the driver parses the LAST stdout line; round 4's 24 KB line came back as
`parsed: null`.  The emitter is run here on the complete object of a real run (profiles/r04_bench.json, the 24 KB one) and on
a synthetic N = 2 object: the last line must round-trip through json.loads, stay under the limit and carry the contract keys
with `roofline` and `cpu_baseline`; the tables must come out before it, one small line per row."""
import io, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config")


def _full():
    return json.load(open(os.path.join(ROOT, "profiles", "r04_bench.json")))


def test_last_line_is_small_and_complete(tmp_path):
    out = _full()
    assert len(json.dumps(out)) > 20000          # the object that broke the driver's parser
    buf = io.StringIO()
    bench.emit(out, str(tmp_path / "x" / "bench_full.json"), stream=buf)
    lines = buf.getvalue().splitlines()
    last = lines[-1]
    assert len(last) < bench.LINE_LIMIT <= 4096
    d = json.loads(last)
    for k in CONTRACT:
        assert k in d, k
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-3
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(d["cpu_baseline"])
    assert d["value"] == round(out["value"], 3) and abs(d["ms_per_step"] - out["ms_per_step"]) < 1e-3
    assert len(d["sweep_fwd_bwd_ms"]) == len(out["sweep"]) and len(d["readme_x_h100"]) == len(out["readme_table"])
    # the tables: one line per row ahead of the contract line, each small, none of them with the contract's keys
    rows = [json.loads(l) for l in lines[:-1]]
    assert len(rows) == len(out["configs"]) + len(out["sweep"]) + len(out["readme_table"])
    assert all("table" in r and "metric" not in r for r in rows) and max(len(l) for l in lines[:-1]) < 1024
    # and the complete object went to the side file
    assert json.load(open(tmp_path / "x" / "bench_full.json"))["sweep"] == out["sweep"]


def test_multi_gpu_line_carries_strong_and_weak():
    out = _full()
    for k in ("sweep", "configs", "readme_table", "cpu_baseline"):
        out.pop(k)
    out.update(n_gpus=8, scaling="strong",
               strong={"value": 5e7, "unit": "seq/s", "ms_per_step": 0.25, "scaling": "strong", "heads_per_rank": 96, "workload": "x" * 200},
               weak={"value": 8e7, "unit": "seq/s", "ms_per_step": 1.2, "scaling": "weak", "heads_per_rank": 768})
    buf = io.StringIO()
    d = json.loads(bench.emit(out, None, stream=buf))
    assert buf.getvalue().count("\n") == 1 and d["strong"]["heads_per_rank"] == 96 and d["weak"]["value"] == 8e7
    assert "workload" not in d["strong"]


def test_line_carries_tflops_traffic_source_and_the_strong_rows(tmp_path):
    """round 6 (VERDICT r05 missing #4 / #5, weak #11): BASELINE's metric is "TFLOP/s & seq/s" -- both TFLOP/s figures ride on the contract
    line; `roofline.traffic` comes out of a committed rocprofv3 summary and the line names it (file # sha16); an N > 1 run carries the
    strong-scaled rows of the metric's grid as [fwd+bwd ms, heads per rank]"""
    out = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_full.json")))
    out.update(n_gpus=8, scaling="strong", strong_rows=[
        {"row": n, "fft": 2 * L, "fft_run": 2 * L, "B": 16, "H": 768, "L": L, "n_gpus": 8, "heads_per_rank": 96, "H_run": 96, "rescaled": False,
         "step_ms": 0.1234567 * (i + 1), "seq_per_s": 1e6, "scaling": "strong", "timing": "fwd+bwd, 10 steps between barriers, max over 8 ranks"}
        for i, (n, L) in enumerate((("sweep L=1024", 1024), ("sweep L=16384", 16384), ("sweep L=131072", 131072), ("sweep L=1048576", 1048576)))] +
        [{"row": "cfg4 H-sharded", "fft": 4194304, "fft_run": 2097152, "B": 1, "H": 16, "L": 1048576, "n_gpus": 8, "heads_per_rank": 2, "H_run": 2,
          "rescaled": False, "step_ms": 0.2, "seq_per_s": 8e4, "scaling": "strong", "timing": "x"}])
    buf = io.StringIO()
    last = bench.emit(out, str(tmp_path / "bench_full.json"), stream=buf)
    d = json.loads(last)
    assert len(last) < bench.LINE_LIMIT
    assert d["tflops_fft_equiv"] == round(out["tflops_fft_equiv"], 3) and d["tflops_dense_monarch"] > d["tflops_fft_equiv"]
    src = d["roofline"]["traffic_source"]
    assert src.startswith("profiles/r0") and "#" in src and len(src.split("#")[1]) == 16
    assert d["strong_rows_step_ms"]["cfg4 H-sharded"] == [0.2, 2] and len(d["strong_rows_step_ms"]) == 5
    rows = [json.loads(l) for l in buf.getvalue().splitlines()[:-1]]
    assert sum(r["table"] == "strong_rows" for r in rows) == 5


def test_rows_that_ran_a_fitted_fft_size_are_flagged_on_the_line(tmp_path):
    """round 5 (FlashFFTConv._fit_seqlen): a config row whose module ran a smaller fft size than it was built for says so on the contract
    line, with the seqlen-point timing of the same module next to it; the final round-5 object stays under the limit with them"""
    out = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_full.json")))
    for r in out["configs"]:
        if r["row"].startswith("cfg4"):
            r.update(fft_run=2097152, fwd_ms_seqlen_points=0.6864, bwd_ms_seqlen_points=0.7577)
    buf = io.StringIO()
    last = bench.emit(out, str(tmp_path / "bench_full.json"), stream=buf)
    d = json.loads(last)
    assert len(last) < bench.LINE_LIMIT
    assert d["configs_fft_run"] == {"cfg4": 2097152} and d["configs_seqlen_points_fwd_bwd_ms"] == {"cfg4": [0.6864, 0.7577]}
    assert d["configs_fwd_bwd_ms"]["cfg4"][0] < 0.6
