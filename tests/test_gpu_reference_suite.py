"""The reference's OWN tests for this path, unmodified, against the product on the GPU (round-2 review item 7).

/root/reference/tests/utils/test_processing_utils.py:15-35 and /root/reference/tests/loss/test_li_losses.py:15-147 run in a
subprocess whose `colpali_engine` package is tests/reference_suite/stub (scorers and loss classes = colpali_amd's).  See
tests/reference_suite/README.md for how the two files get here without being committed."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
SUITE = os.path.join(ROOT, "tests", "_reference_tests")
STUB = os.path.join(ROOT, "tests", "reference_suite", "stub")


def _run(files, device, select):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([STUB, ROOT]), REFSUITE_DEVICE=device, PYTHONDONTWRITEBYTECODE="1")
    cmd = [sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider", "-p", "refsuite_plugin", "--rootdir", SUITE, "-c", os.devnull]
    if select:
        cmd += ["-k", select]
    res = subprocess.run(cmd + files, capture_output=True, text=True, env=env, cwd=SUITE, timeout=600)
    tail = (res.stdout + res.stderr)[-3000:]
    assert res.returncode == 0, tail
    return res.stdout


def _files():
    if not os.path.isdir(SUITE):
        pytest.skip("tests/_reference_tests/ is absent: run __graft_entry__.build() where /root/reference exists "
                    "(oracle/fetch_reference_tests.py); the directory travels with the working tree")
    return os.path.join(SUITE, "test_processing_utils.py"), os.path.join(SUITE, "test_li_losses.py")


def test_reference_scorer_tests_pass_on_the_product():
    scorer, _ = _files()
    out = _run([scorer], "cuda:0", None)
    assert "2 passed" in out, out


def test_reference_loss_class_tests_pass_on_the_product():
    _, losses = _files()
    out = _run([losses], "cuda:0", "not TestColbertModule")
    assert "7 passed" in out, out          # ColbertLoss 2, NegativeCE 2, PairwiseCE 1, PairwiseNegativeCE 2


def test_reference_helper_tests_pass_on_the_product():
    _, losses = _files()
    out = _run([losses], "cpu", "TestColbertModule")
    assert "6 passed" in out, out
