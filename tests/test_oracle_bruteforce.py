"""Oracle vs an independent naive Python evaluation of the same recursion (tests/bruteforce.py)."""
import numpy as np
import pytest

from margin_amd import synth
from tests import bruteforce
from tests.helpers import assert_job_equal


def _jobs(orc, chunk, pd, limit_cells=4000):
    oc = orc.OracleChunk(chunk)
    res = oc.phase(pd, capture_jobs=True)
    oc.close()
    return [j for j in res["jobs"] if len(j["partition"]) <= limit_cells]


def test_max_mode_bit_exact_vs_bruteforce(orc):
    chunk = synth.make_ont_chunk(seed=8, region_bp=20_000, n_sites=40, coverage=14)
    pd = synth.shipped_phase_params()
    pd["maxPartitionsInAColumn"] = pd["minPartitionsInAColumn"] = 12
    jobs = _jobs(orc, chunk, pd)
    assert len(jobs) >= 5 and any(j["flags"] & 2 for j in jobs)
    for j in jobs:
        assert_job_equal(j, bruteforce.forward_backward(chunk, j, j["flags"]), exact=True)


def test_sum_mode_multiallelic_vs_bruteforce(orc):
    chunk = synth.make_unit_test_chunk(seed=31, ref_length=40, coverage=8, min_read=5, max_read=25, error_rate=0.05)
    pd = synth.unit_test_params(max_partitions=10, max_not_sum=0)
    pd["includeAncestorSubProb"] = 1
    pd["roundsOfIterativeRefinement"] = 2
    jobs = _jobs(orc, chunk, pd)
    assert len(jobs) >= 3
    for j in jobs:
        assert_job_equal(j, bruteforce.forward_backward(chunk, j, j["flags"]), exact=False, atol=1e-9)
