"""`make -C oracle asan` (race / memory checking): the oracle's C++ restatement rebuilt under AddressSanitizer +
UndefinedBehaviorSanitizer and its own CPU tests (tests/test_oracle.py) run against that build in a child process.  Round 4's first
run of it found the signed overflow the reference itself commits on a NaN pose (PointToVoxel(NaN) + shift): spelled out as a
wrap in the restatement since."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


def test_oracle_restatement_is_clean_under_asan_and_ubsan():
    gxx = shutil.which("g++")
    if gxx is None or not os.path.isabs(subprocess.run([gxx, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()):
        pytest.skip("g++ / libasan not available")
    p = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert " passed" in p.stdout and "failed" not in p.stdout
