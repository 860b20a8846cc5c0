"""Committed fixtures (tests/golden/*.npz; oracle-generated, see make_golden.py for provenance):
the oracle must still reproduce them (CPU) and so must the HIP path through the C-ABI (GPU)."""
import os

import numpy as np
import pytest

from tests import bruteforce
from tests.helpers import assert_job_equal

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ["ont_max_mode", "unit_sum_mode"]
KEYS = ["col_ref_start", "col_length", "col_depth", "col_cell_off", "col_read_off", "read_byte_off", "partition",
        "mask_from", "mask_to", "mcol_cell_off", "merge_from", "merge_to", "cell_next", "cell_prev", "cell_forward",
        "cell_backward", "merge_forward", "merge_backward", "col_total"]


class _Chunk:
    def __init__(self, z):
        self.allele_number, self.sub, self.prior, self.pool = z["allele_number"], z["sub"], z["prior"], z["pool"]
        self.allele_offset = np.concatenate([[0], np.cumsum(self.allele_number)]).astype(np.int64)


def load(name):
    z = np.load(os.path.join(HERE, name + ".npz"))
    jobs = []
    for i in range(int(z["n_jobs"])):
        j = {k: z[f"j{i}_{k}"] for k in KEYS}
        j["n_columns"], j["flags"] = (int(x) for x in z[f"j{i}_scalars"])
        j["hmm_forward"], j["hmm_backward"] = (float(x) for x in z[f"j{i}_hmm_fb"])
        jobs.append(j)
    return _Chunk(z), jobs


@pytest.mark.parametrize("name", NAMES)
def test_golden_vs_independent_bruteforce(name):
    chunk, jobs = load(name)
    exact = name == "ont_max_mode"
    for j in jobs:
        if len(j["partition"]) <= 3000:
            assert_job_equal(j, bruteforce.forward_backward(chunk, j, j["flags"]), exact=exact, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_golden_on_gpu(gpu_ctx, name):
    from margin_amd import capi
    from tests.helpers import run_jobs_on_gpu
    chunk, jobs = load(name)
    dchunk = capi.DeviceChunk(gpu_ctx, chunk.allele_number, chunk.sub, chunk.prior, chunk.pool)
    out = run_jobs_on_gpu(gpu_ctx, dchunk, jobs)
    for j, r in zip(jobs, out):
        assert_job_equal(j, r, exact=(name == "ont_max_mode"), atol=1e-9)
    dchunk.close()


# ---- whole phasing runs -----------------------------------------------------------------------
PHASE_NAMES = ["phase_ont", "phase_unit_max"]
PHASE_OUT = ["hap1", "hap2", "genotype", "ancestor", "genotype_probs", "hap_probs1", "hap_probs2", "support1", "support2"]


def load_phase(name):
    import json
    from margin_amd import synth
    z = np.load(os.path.join(HERE, name + ".npz"))
    off = np.concatenate([[0], np.cumsum(z["allele_number"])]).astype(np.int64)
    reads = []
    for i in range(len(z["read_ref_start"])):
        s, n = int(z["read_ref_start"][i]), int(z["read_length"][i])
        reads.append(synth.Read(name=str(z["read_names"][i]), ref_start=s, length=n, strand=int(z["read_strand"][i]), hap=0,
                                pool_off=int(z["read_pool_off"][i]), nbytes=int(off[s + n] - off[s])))
    chunk = synth.Chunk(allele_number=z["allele_number"], allele_offset=off, sub=z["sub"], prior=z["prior"], pool=z["pool"], reads=reads)
    return chunk, json.loads(str(z["params"])), z


def assert_phase_equal(got, z):
    assert int(got["ref_start"]) == int(z["ref_start"]) and int(got["length"]) == int(z["length"])
    for k in PHASE_OUT:
        assert (np.asarray(got[k]) == z[k]).all(), k
    assert list(got["reads1"]) == z["reads1"].tolist() and list(got["reads2"]) == z["reads2"].tolist()


@pytest.mark.parametrize("name", PHASE_NAMES)
def test_golden_phase_oracle(orc, name):
    chunk, pd, z = load_phase(name)
    oc = orc.OracleChunk(chunk)
    res = oc.phase(pd)
    oc.close()
    assert_phase_equal(res, z)
    assert res["fb_calls"] == int(z["fb_calls"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", PHASE_NAMES)
def test_golden_phase_on_gpu(gpu_ctx, name):
    """both product paths against the committed vectors: the per-chunk path and the device-resident pipeline"""
    from margin_amd import capi
    chunk, pd, z = load_phase(name)
    params = capi.Params.from_reference_names(pd)
    dchunk = capi.DeviceChunk.from_chunk(gpu_ctx, chunk)
    host = capi.phase_reads(gpu_ctx, dchunk, chunk, params)
    assert_phase_equal(host, z)
    (res,), st = capi.phase_reads_many(gpu_ctx, [dchunk], [chunk], params)
    assert st.resident == 1
    assert_phase_equal(res, z)
    assert res["n_sweeps"] == int(z["fb_calls"]) == host["n_sweeps"]
    dchunk.close()
