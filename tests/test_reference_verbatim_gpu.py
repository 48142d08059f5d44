"""The reference's OWN test files, unmodified, against this package (`north_star`: "correctness passing
tests/test_flashfftconv.py"; VERDICT r03 missing #4).  oracle/fetch_reference_tests.py stages them byte for byte from
/root/reference/tests into oracle/_ref/reference_tests/ (git-ignored, travels to the GPU box); here their SHA-256 is checked
against the pinned values and pytest runs them in a subprocess whose `flashfftconv` is flash-fft-conv_amd/flashfftconv.

reference tests/test_flashfftconv.py:48-324: 4 tests x B {1,2,4,8,64} x H {768,111} x {fp16,bf16} x 14 fft sizes (256 .. 4M);
reference tests/test_conv1d.py:8-220: BHL / BLH forward (7 dtype pairs) and backward (3 dtype pairs).
FFC_REF_TESTS_K="<pytest -k expression>" narrows a manual run."""
import os, subprocess, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.fetch_reference_tests import DEST, FILES, sha256

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(FILES))
def test_reference_test_file_passes_unmodified(name, tmp_path):
    path = os.path.join(DEST, name)
    if not os.path.exists(path):
        pytest.skip(f"{path} not staged (run __graft_entry__.build() where /root/reference exists)")
    assert sha256(path) == FILES[name], "staged file differs from the reference's"
    env = dict(os.environ)
    env["PYTHONPATH"] = os.path.join(ROOT, "flash-fft-conv_amd") + os.pathsep + env.get("PYTHONPATH", "")
    env["PYTHONBREAKPOINT"] = "0"          # the reference tests call breakpoint() before a failing assert
    cmd = [sys.executable, "-m", "pytest", path, "-x", "-q", "-p", "no:cacheprovider", "--rootdir", str(tmp_path)]
    try:
        import xdist  # noqa: F401  (one GPU, but half of a case's time is host work: the torch.fft reference, allocations)
        cmd += ["-n", os.environ.get("FFC_REF_TESTS_PROCS", "4")]
    except ImportError:
        pass
    if os.environ.get("FFC_REF_TESTS_K"):
        cmd += ["-k", os.environ["FFC_REF_TESTS_K"]]
    r = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=3000)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-25:])
    assert r.returncode == 0, f"reference {name} failed against the drop-in:\n{tail}"
    print(tail.splitlines()[-1] if tail else "")
