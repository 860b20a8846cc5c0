"""The host side of the device-resident pipeline without a device (tools/hostbench): rphmm_host.c compiled with the engine
replaced by stubs that only hand out segment numbers.  Tiling paths, overlap components, the merged column boundaries of every
cross product (r_cross_build), the final shadows and their expansion into per-column read lists (r_expand checks the column
depths against the reads' intervals) run for whole chunks; any inconsistency makes the call fail."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HB = os.path.join(ROOT, "tools", "hostbench")


@pytest.mark.parametrize("threads", [1, 4])
def test_host_structure_of_whole_chunks_is_consistent(tmp_path, threads):
    subprocess.check_call(["make", "-C", HB, "hostbench"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    chunks = str(tmp_path / "chunks.bin")
    subprocess.check_call([sys.executable, os.path.join(HB, "dump_chunks.py"), "3", chunks, "--sites", "300"])
    out = subprocess.check_output([os.path.join(HB, "hostbench"), chunks, "2", str(threads)], text=True)
    lines = [l for l in out.splitlines() if l.startswith("run ")]
    assert len(lines) == 2
    for l in lines:
        assert "rc 0 (ok)" in l and "levels" in l, l
    # the same chunks give the same structure in every run and for any number of host threads
    cols = {l.split("columns ")[1].split(",")[0] for l in lines}
    assert len(cols) == 1 and int(cols.pop()) > 1000
    # ... and everything the engine is told (intervals, column boundaries, read offsets, parents, static bounds) hashes alike
    hashes = {l.split("structure hash ")[1].strip() for l in lines}
    assert len(hashes) == 1
    _HASHES.setdefault("h", set()).update(hashes)
    assert len(_HASHES["h"]) == 1, "the level descriptions depend on the number of host threads"


_HASHES = {}


def test_tiling_paths_in_one_pass_equal_the_reference_walk():
    """getTilingPaths (coordination.c:19-55, 186-222) builds path after path; the product deals the sorted hmms out in one first-fit
    pass.  hostbench --selftest compares the two on 400 random interval sets (nested, touching, equal intervals; up to 2 500 hmms),
    and checks the radix / insertion sort of stRPHmm_cmpFn on the way."""
    subprocess.check_call(["make", "-C", HB, "hostbench"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = subprocess.check_output([os.path.join(HB, "hostbench"), "--selftest"], text=True)
    assert out.strip() == "selftest ok", out
