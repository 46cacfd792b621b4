"""CPU simulation of the primer prescan (specimux_amd/csrc/smx_prescan_core.h: the host/device code the gfx950 kernel
smx_prescan.hip runs) against a plain O(mn) dynamic program: every flag word, and the consumer-side decode (distance,
first optimal end, optimal-end mask), for several window lengths and symbol-table shapes.  No GPU needed."""
import os
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sim(tmp_path_factory):
    exe = os.fspath(tmp_path_factory.mktemp("prescan") / "prescan_sim")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(REPO, "specimux_amd", "csrc"), "-o", exe,
                           os.path.join(REPO, "tests", "cpu", "prescan_sim.cpp")])
    return exe


@pytest.mark.parametrize("args", [("80", "1"), ("160", "2"), ("16", "3"), ("256", "4"), ("80", "7", "one-degenerate-letter"),
                                  ("48", "9")], ids=lambda a: "S%s-seed%s%s" % (a[0], a[1], "-nsym5" if len(a) > 2 else ""))
def test_prescan_equals_dp(sim, args):
    out = subprocess.run([sim, *args], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert " 0 mismatches" in out.stdout
