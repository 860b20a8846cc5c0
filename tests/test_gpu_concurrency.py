"""Several host threads (one context each, as the reference's OpenMP chunk loop phase.c:276 would use
the library) sweep the same jobs concurrently; every result must be bit-identical to the oracle.
Regression test for a prefetch-ring register hazard that only showed up under concurrent load."""
import threading

import numpy as np
import pytest

from margin_amd import capi, synth
from tests.helpers import assert_job_equal

pytestmark = pytest.mark.gpu


def test_concurrent_contexts_bit_exact(orc):
    chunk = synth.make_ont_chunk(seed=11, region_bp=150_000, n_sites=300, coverage=30)
    oc = orc.OracleChunk(chunk)
    flats = oc.phase(synth.shipped_phase_params(), capture_jobs=True)["jobs"]
    oc.close()
    errors = []

    def worker():
        try:
            ctx = capi.Context(0)
            dchunk = capi.DeviceChunk.from_chunk(ctx, chunk)
            for _ in range(6):
                jobs = [capi.Job(dchunk, f, int(f["flags"])) for f in flats]
                capi.fb_run(ctx, jobs)
                for f, j in zip(flats, jobs):
                    assert_job_equal(f, j.results(), exact=True)
            dchunk.close()
            ctx.close()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker) for _ in range(5)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]
