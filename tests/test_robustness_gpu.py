"""Robustness of the HIP path against state it does not own (run as scripts so that torch.empty can be patched
process-wide): results must not depend on (a) what the previous workgroup left in a CU's LDS / vector / accumulation
registers (benchmarks/gpu_poison.py, ffc_debug_poison) or (b) the previous contents of any buffer the Python layer
allocates (benchmarks/gpu_uninit.py pre-fills every torch.empty with NaN bit patterns)."""
import os, subprocess, sys
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "benchmarks")


@pytest.mark.parametrize("script,done", [("gpu_poison.py", "poison stress done, mismatching tensors: 0"),
                                         ("gpu_uninit.py", "uninit hunt done, mismatching tensors: 0")])
def test_script(script, done):
    r = subprocess.run([sys.executable, os.path.join(HERE, script)], capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert done in r.stdout, tail


@pytest.mark.parametrize("env,sizes", [({"FFC_MULTIPASS": ""}, ["2048", "65536", "131072"]), ({"FFC_BIG_2LEVEL": "1"}, ["2097152"]),
                                       ({"FFC_BIG_1LEVEL": "1"}, ["4194304"]),
                                       # round 3: the recomputing backward (reference memory footprint) at one size of every kind, and
                                       # the two-level 4M / 2-pass-inner 2M forms (L = N/2 runs them anyway at 4M; N/4 cases are forced)
                                       ({"FFC_SAVE_SPECTRUM": "0"}, ["4096", "32768", "65536", "262144", "2097152"]),
                                       ({"FFC_BIG_ONE128": "0"}, ["2097152", "4194304"]),
                                       ({"FFC_BIG_ONE_LAUNCH": "0"}, ["2097152", "4194304"])])
def test_alternative_factorisations_stay_correct(env, sizes):
    """the round-1 paths and the measured-slower factorisations stay reachable through environment switches (A/B runs):
    forward + every gradient against the torch.fft oracle (benchmarks/alt_paths_check.py)"""
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, os.path.join(HERE, "alt_paths_check.py")] + sizes, capture_output=True, text=True, timeout=900, env=e)
    assert r.returncode == 0 and "alt paths ok" in r.stdout, (r.stdout + r.stderr)[-3000:]
