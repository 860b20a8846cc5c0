"""GPU parity of the device-resident merge (cross product -> sweep -> prune without leaving HBM,
SURVEY.md 8 f-1) against the oracle and against the hashing host path."""
import numpy as np
import pytest

from margin_amd import capi, synth
from tests.test_gpu_pipeline import assert_same_hmm

pytestmark = pytest.mark.gpu

PHASE_KEYS = ("hap1", "hap2", "genotype", "ancestor", "support1", "support2", "genotype_probs", "hap_probs1", "hap_probs2")


def _params(**over):
    pd = synth.shipped_phase_params()
    pd.update(over)
    return pd


@pytest.mark.parametrize("seed,n_sites,cov,over", [
    (3, 200, 30, {}),
    (4, 120, 45, {}),
    (9, 60, 12, {}),
    (6, 150, 30, dict(minPartitionsInAColumn=0, maxPartitionsInAColumn=50)),                       # unit-test style trimming
    (7, 150, 30, dict(minPartitionsInAColumn=10, maxPartitionsInAColumn=64, minPosteriorProbabilityForPartition=1e-6)),
    (8, 100, 25, dict(includeInvertedPartitions=0, minPartitionsInAColumn=0, maxPartitionsInAColumn=40)),  # row-major cross product
])
def test_resident_get_rp_hmms_matches_oracle(gpu_ctx, orc, seed, n_sites, cov, over):
    """The pruned hmms the resident merge returns are the oracle's, array for array: every level's
    closed-form cross product order, the integer posterior ranking and the stable tie order."""
    chunk = synth.make_ont_chunk(seed=seed, region_bp=n_sites * 500, n_sites=n_sites, coverage=cov)
    pd = _params(includeAncestorSubProb=0, **over)
    oc = orc.OracleChunk(chunk)
    dchunk = capi.DeviceChunk.from_chunk(gpu_ctx, chunk)
    params = capi.Params.from_reference_names(pd)
    for strand in (1, 0):
        idx = [i for i, r in enumerate(chunk.reads) if r.strand == strand]
        ref = oc.get_rp_hmms(orc.make_params(pd), idx)
        got = capi.get_rp_hmms_resident(gpu_ctx, dchunk, chunk, params, idx)
        assert len(ref) == len(got)
        for hr, hg in zip(ref, got):
            assert_same_hmm(orc.flatten(hr, oc.pool_off), capi.hmm_to_flat(hg), values=False)
            capi.hmm_destroy(hg)
    dchunk.close()
    oc.close()


def test_resident_phase_many_matches_oracle_and_host_path(gpu_ctx, orc):
    """Several chunks of different shapes phased in one call (their merge levels share launches): HP
    partition, haplotype strings and genotype calls equal the oracle's and the per-chunk host path's."""
    specs = [(3, 200, 30, 64), (5, 150, 40, 12), (11, 80, 20, 64), (12, 30, 8, 64)]
    chunks = [synth.make_ont_chunk(seed=s, region_bp=n * 500, n_sites=n, coverage=c) for s, n, c, _ in specs]
    pd = _params(maxCoverageDepth=64)
    params = capi.Params.from_reference_names(pd)
    dchunks = [capi.DeviceChunk.from_chunk(gpu_ctx, c) for c in chunks]
    got, st = capi.phase_reads_many(gpu_ctx, dchunks, chunks, params)
    assert st.resident == 1 and st.levels > 0 and st.cells > 0
    for chunk, dchunk, g in zip(chunks, dchunks, got):
        oc = orc.OracleChunk(chunk)
        ref = oc.phase(pd)
        oc.close()
        host = capi.phase_reads(gpu_ctx, dchunk, chunk, params)
        for other, name in ((ref, "oracle"), (host, "host path")):
            assert g["ref_start"] == other["ref_start"] and g["length"] == other["length"], name
            for k in PHASE_KEYS:
                assert (np.asarray(g[k]) == np.asarray(other[k])).all(), (name, k)
            assert g["reads1"] == other["reads1"] and g["reads2"] == other["reads2"], name
        assert g["hmm_forward"] == host["hmm_forward"] and g["hmm_backward"] == host["hmm_backward"]
        assert g["n_sweeps"] == ref["fb_calls"] == host["n_sweeps"]  # one sweep per overlap component + the final one
    for d in dchunks:
        d.close()


def test_resident_coverage_filter_and_empty_chunk(gpu_ctx, orc):
    """maxCoverageDepth lowered (reads discarded and re-added, bubbleGraph.c:2772) and a chunk without reads."""
    chunk = synth.make_ont_chunk(seed=5, region_bp=150 * 500, n_sites=150, coverage=40)
    empty = synth.make_ont_chunk(seed=6, region_bp=20 * 500, n_sites=20, coverage=5)
    empty.reads = []
    pd = _params(maxCoverageDepth=12)
    params = capi.Params.from_reference_names(pd)
    dchunks = [capi.DeviceChunk.from_chunk(gpu_ctx, c) for c in (chunk, empty)]
    got, st = capi.phase_reads_many(gpu_ctx, dchunks, [chunk, empty], params)
    assert st.resident == 1
    oc = orc.OracleChunk(chunk)
    ref = oc.phase(pd)
    oc.close()
    for k in PHASE_KEYS:
        assert (np.asarray(got[0][k]) == np.asarray(ref[k])).all(), k
    assert got[0]["reads1"] == ref["reads1"] and got[0]["reads2"] == ref["reads2"]
    assert got[1]["length"] == 0 and got[1]["reads1"] == [] and got[1]["reads2"] == []
    for d in dchunks:
        d.close()


def test_resident_falls_back_outside_its_range(gpu_ctx, orc):
    """Log-sum-exp mode is not handled by the resident merge: mrp_get_rp_hmms_resident says so, and
    mrp_phase_reads_many takes the per-chunk path (stats.resident == 0) with the same result."""
    chunk = synth.make_unit_test_chunk(seed=21, ref_length=100, coverage=10, min_read=10, max_read=50, error_rate=0.05)
    pd = synth.unit_test_params(max_partitions=50, max_not_sum=0)
    params = capi.Params.from_reference_names(pd)
    dchunk = capi.DeviceChunk.from_chunk(gpu_ctx, chunk)
    with pytest.raises(capi.MrpError) as ei:
        capi.get_rp_hmms_resident(gpu_ctx, dchunk, chunk, params)
    assert ei.value.code == capi.MRP_ERR_UNSUPPORTED
    got, st = capi.phase_reads_many(gpu_ctx, [dchunk], [chunk], params)
    assert st.resident == 0
    host = capi.phase_reads(gpu_ctx, dchunk, chunk, params)
    for k in PHASE_KEYS:
        assert (np.asarray(got[0][k]) == np.asarray(host[k])).all(), k
    assert got[0]["reads1"] == host["reads1"] and got[0]["reads2"] == host["reads2"]
    dchunk.close()
