"""Host check of csrc/rl_crmath.hpp -- the correctly rounded heading / normal directions of the sweep's reference-order
mode (yaw = RN(atan2(y', x')), RN(cos / sin(fl(yaw +- fl(pi/2)))), models/trajectory.py:87-92, 250) -- against libquadmath.
The header is plain IEEE double arithmetic, so the host build returns the bits the device returns
(tests/test_hip_parity.py::test_cr_heading_on_the_device compares the device with the oracle's libquadmath build)."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_cr_heading_matches_libquadmath(tmp_path):
    exe = str(tmp_path / "crmath_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", os.path.join(HERE, "crmath_check.cpp"),
                           "-o", exe, "-lquadmath"])
    out = subprocess.run([exe, "300000"], capture_output=True, text=True)
    print(out.stdout)
    print(out.stderr[-2000:])
    assert out.returncode == 0, out.stdout
    assert "TOTAL 0" in out.stdout
    # every family ran, and the exact-axis constants are right
    for fam in ("uniform", "ratios", "breaks", "axes"):
        line = [l for l in out.stdout.splitlines() if l.startswith(fam)][0]
        assert "yaw=0 cl=0 sl=0 cr=0 sr=0" in line, line
