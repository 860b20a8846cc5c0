"""examples/phase_from_strings.c: the C-ABI used from plain C (no Python, no torch): it must compile against include/ and
link against the in-tree library; on a GPU it must run the whole chain and tag reads with their haplotypes."""
import os
import subprocess

import pytest

from margin_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "phase_from_strings")
    libdir = os.path.dirname(capi.LIB_PATH)
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "phase_from_strings.c"),
                           "-L" + libdir, "-lmargin_rphmm", "-lm", "-Wl,-rpath," + libdir, "-o", exe])
    return exe


def test_c_example_builds_and_refuses_to_run_without_a_device(tmp_path):
    exe = _build(tmp_path)
    if capi.load().mrp_device_count() > 0:
        pytest.skip("a GPU is visible here")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_c_example_phases_reads(tmp_path):
    r = subprocess.run([_build(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "resident=1" in r.stdout and "agree with their haplotype" in r.stdout
