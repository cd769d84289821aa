"""The reference's own interface tests (tests/test_interface.py, 278 cases: constructor, every setter's values, error
types and messages) run UNMODIFIED against this repository's ``Simulator`` by aliasing the package name ``VGsim`` to
``vgsim_amd``.  Only possible where the reference checkout is mounted (the development container); skipped elsewhere —
nothing of the reference is copied, the file is executed where it lies."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_TESTS = os.environ.get("VGSIM_REFERENCE_TESTS", "/root/reference/tests/test_interface.py")


@pytest.mark.skipif(not os.path.exists(REF_TESTS), reason="reference checkout not mounted")
def test_reference_interface_tests_pass_against_this_simulator(tmp_path):
    shim = tmp_path / "VGsim"
    shim.mkdir()
    (shim / "__init__.py").write_text("from vgsim_amd import *  # noqa\nfrom vgsim_amd import Simulator, IO  # noqa\n")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([str(tmp_path), ROOT]))
    out = subprocess.run([sys.executable, "-m", "pytest", REF_TESTS, "-q", "-p", "no:cacheprovider", "--rootdir", str(tmp_path)],
                         cwd=str(tmp_path), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600).stdout.decode()
    tail = out.strip().splitlines()[-1]
    assert " passed" in tail and "failed" not in tail and "error" not in tail, out[-2000:]
    assert int(tail.split(" passed")[0].split()[-1]) >= 278
