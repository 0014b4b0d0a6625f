"""GPU: the two arithmetic identities the default extend kernel rests on, checked on the hardware
itself by the exhaustive tools under tests/tools/ (full runs: profiles/r01/r01_div3_exhaustive.log,
profiles/r01/r01_rcp_exhaustive.log).  Here: every 512th divisor significand against all 2^23 dividend
significands (1.4e11 pairs), and the reciprocal for all 2^31 values of its range."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-w"]


def build(name, tmp_path):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available on this box")
    exe = tmp_path / name
    subprocess.check_call([HIPCC] + FLAGS + [os.path.join(ROOT, "tests", "tools", name + ".hip"), "-o", str(exe)])
    return str(exe)


def test_packed_division_is_the_ieee_quotient_on_this_gpu(tmp_path):
    out = subprocess.run([build("div3_exhaustive", tmp_path), "512"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-400:] + out.stderr[-400:]
    assert "mismatches 0" in out.stdout.splitlines()[-1]


def test_rcp_plus_newton_is_the_ieee_reciprocal_on_this_gpu(tmp_path):
    out = subprocess.run([build("rcp_exhaustive", tmp_path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-400:] + out.stderr[-400:]
    assert "mismatches: 5-op 0, 3-op 0" in out.stdout
