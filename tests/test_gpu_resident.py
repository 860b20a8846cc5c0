"""GPU parity of the device-resident merge (cross product -> sweep -> prune without leaving HBM,
SURVEY.md 8 f-1) against the oracle and against the hashing host path."""
import os

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
    # nine chunks as two concurrent batches on sibling contexts (the default would need 48 chunks for two)
    specs = [(3, 200, 30, 64), (5, 150, 40, 12), (11, 80, 20, 64), (12, 30, 8, 64), (13, 60, 25, 64), (14, 90, 15, 64),
             (15, 40, 30, 64), (16, 120, 20, 64), (17, 70, 35, 64)]
    chunks = [synth.make_ont_chunk(seed=s, region_bp=n * 500, n_sites=n, coverage=c) for s, n, c, _ in specs]
    pd = _params(maxCoverageDepth=64)
    params = capi.Params.from_reference_names(pd)
    dchunks = [capi.DeviceChunk.from_chunk(gpu_ctx, c) for c in chunks]
    gpu_ctx.set_phase_groups(2)
    try:
        got, st = capi.phase_reads_many(gpu_ctx, dchunks, chunks, params)
    finally:
        gpu_ctx.set_phase_groups(0)
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
    assert st.resident == 0 and b"maxNotSumTransitions" in st.note  # the reason is reported, not only resident = 0
    host = capi.phase_reads(gpu_ctx, dchunk, chunk, params)
    for k in PHASE_KEYS:
        assert (np.asarray(got[0][k]) == np.asarray(host[k])).all(), k
    assert got[0]["reads1"] == host["reads1"] and got[0]["reads2"] == host["reads2"]
    dchunk.close()


def test_more_partitions_than_the_resident_kernels_keep_take_the_hashing_path_in_parallel(gpu_ctx, orc):
    """maxPartitionsInAColumn = 200 is the reference's code default (parser.c:22-23); the resident kernels keep at most 116
    per column.  The call does not fail: its chunks go through the hashing path, several host threads with a context each,
    and equal the oracle; the statistics say why."""
    pd = _params()
    pd["minPartitionsInAColumn"], pd["maxPartitionsInAColumn"] = 50, 200
    params = capi.Params.from_reference_names(pd)
    chunks = [synth.make_ont_chunk(seed=71 + i, region_bp=30_000, n_sites=60, coverage=16 + 2 * i) for i in range(5)]
    dchunks = [capi.DeviceChunk.from_chunk(gpu_ctx, c) for c in chunks]
    got, st = capi.phase_reads_many(gpu_ctx, dchunks, chunks, params)
    assert st.resident == 0 and b"partitions per column" in st.note
    for chunk, g in zip(chunks, got):
        _assert_equals_oracle(orc, chunk, g, pd)
    for d in dchunks:
        d.close()


def test_bubbles_to_haplotype_tags_end_to_end(gpu_ctx, orc):
    """The whole chain a chunk goes through around and on the path: bubble graph -> profile sequences and site tables
    (rphmm_frame.c) -> device-resident phasing -> read-to-haplotype assignment with phred scores, against the same chain
    made of the oracles (frame_oracle.py builder -> rphmm_oracle phasing -> frame_oracle assignment)."""
    from oracle import frame_oracle as fo
    rng = np.random.default_rng(31)
    n_sites, n_reads = 120, 90
    truth = rng.integers(0, 2, size=n_sites)
    spans, haps, strands = [], rng.integers(0, 2, size=n_reads), rng.integers(0, 2, size=n_reads)
    for r in range(n_reads):
        a = int(rng.integers(0, n_sites - 5))
        spans.append((a, int(min(n_sites - 1, a + rng.integers(4, 40)))))
    an, br, sup = [], [], []
    for i in range(n_sites):
        rs = [r for r, (a, b) in enumerate(spans) if a <= i <= b and (i in (a, b) or rng.random() < 0.9)]
        s = np.zeros((2, len(rs)), dtype=np.float32)
        for j, r in enumerate(rs):
            allele = truth[i] if haps[r] == 0 else 1 - truth[i]
            if rng.random() < 0.08:
                allele = 1 - allele
            s[allele, j] = np.float32(-rng.uniform(0.0, 0.3))
            s[1 - allele, j] = np.float32(-rng.uniform(1.0, 9.0))
        an.append(2); br.append(rs); sup.append(s)
    # product chain
    seqs, pool = capi.profile_seqs_from_bubbles(an, br, sup, n_reads)
    a_num, sub, prior = capi.reference_from_bubbles(an, br, sup, 0.0)
    off = np.concatenate([[0], np.cumsum(a_num)]).astype(np.int64)
    reads = [synth.Read(name=f"r{q['read']:04d}", ref_start=q["ref_start"], length=q["length"], strand=int(strands[q["read"]]),
                        hap=int(haps[q["read"]]), pool_off=q["pool_offset"], nbytes=int(off[q["ref_start"] + q["length"]] - off[q["ref_start"]]))
             for q in seqs]
    chunk = synth.Chunk(allele_number=a_num, allele_offset=off, sub=sub, prior=prior, pool=pool, reads=reads)
    pd = _params()
    params = capi.Params.from_reference_names(pd)
    dchunk = capi.DeviceChunk.from_chunk(gpu_ctx, chunk)
    (got,), st = capi.phase_reads_many(gpu_ctx, [dchunk], [chunk], params)
    assert st.resident == 1
    recs, _ = capi.read_records(chunk)
    hap, phred = capi.assign_reads_to_haplotypes(a_num, pool, recs, len(reads), got, min_phred=0)
    # oracle chain
    ref_seqs = fo.get_profile_seqs([fo.Bubble(2, rs, np.asarray(s).reshape(-1).tolist()) for rs, s in zip(br, sup)])
    assert list(ref_seqs.keys()) == [q["read"] for q in seqs]
    ref_pool = np.array([b for p in ref_seqs.values() for b in p["probs"]], dtype=np.uint8)
    assert (ref_pool == pool).all()
    oc = orc.OracleChunk(chunk)
    ref = oc.phase(pd)
    oc.close()
    for k in PHASE_KEYS:
        assert (np.asarray(got[k]) == np.asarray(ref[k])).all(), k
    assert got["reads1"] == ref["reads1"] and got["reads2"] == ref["reads2"]
    pseqs = {i: dict(refStart=p["refStart"], length=p["length"], probs=p["probs"]) for i, p in enumerate(ref_seqs.values())}
    ogf = dict(refStart=ref["ref_start"], length=ref["length"], hap1=ref["hap1"], hap2=ref["hap2"], reads1=set(ref["reads1"]),
               reads2=set(ref["reads2"]))
    h1, h2, ph = fo.phase_bam_chunk_reads(ogf, pseqs, off.tolist(), 0)
    assert {i for i in range(len(reads)) if hap[i] == 1} == h1 and {i for i in range(len(reads)) if hap[i] == 2} == h2
    assert all(phred[i] == ph[i] for i in ph)
    # the phasing is meaningful: reads agree with their true haplotype up to the global label
    agree = sum(1 for i, r in enumerate(reads) if hap[i] in (1, 2) and (hap[i] - 1) == r.hap)
    tagged = int(((hap == 1) | (hap == 2)).sum())
    assert max(agree, tagged - agree) >= 0.85 * tagged
    dchunk.close()


@pytest.mark.parametrize("seed,maxp", [(21, 50), (22, 20), (23, 116)])
def test_resident_unit_test_shape_max_mode(gpu_ctx, orc, seed, maxp):
    """tests/stRPHmmTest.c-shaped input (1..9 alleles per site, so columns mix allele counts and take the general
    emission path) in max-plus mode with the unit tests' trimming (min 0, max N): resident merge == oracle."""
    chunk = synth.make_unit_test_chunk(seed=seed, ref_length=150, coverage=12, min_read=10, max_read=60, error_rate=0.05)
    pd = synth.unit_test_params(max_partitions=maxp, max_not_sum=1)
    oc = orc.OracleChunk(chunk)
    dchunk = capi.DeviceChunk.from_chunk(gpu_ctx, chunk)
    params = capi.Params.from_reference_names(pd)
    ref = oc.get_rp_hmms(orc.make_params(pd))
    got = capi.get_rp_hmms_resident(gpu_ctx, dchunk, chunk, params)
    assert len(ref) == len(got)
    for hr, hg in zip(ref, got):
        assert_same_hmm(orc.flatten(hr, oc.pool_off), capi.hmm_to_flat(hg), values=False)
        capi.hmm_destroy(hg)
    # and the whole driver (ancestor model in the final sweep: sites with up to 9 alleles)
    pd2 = dict(pd, includeAncestorSubProb=1, roundsOfIterativeRefinement=3)
    refp = oc.phase(pd2)
    (gotp,), st = capi.phase_reads_many(gpu_ctx, [dchunk], [chunk], capi.Params.from_reference_names(pd2))
    assert st.resident == 1
    for k in PHASE_KEYS:
        assert (np.asarray(gotp[k]) == np.asarray(refp[k])).all(), k
    assert gotp["reads1"] == refp["reads1"] and gotp["reads2"] == refp["reads2"]
    dchunk.close()
    oc.close()


def test_resident_full_size_config2_chunk(gpu_ctx, orc):
    """BASELINE.json configs[1] at full size (1 Mb, 2 000 het sites, 30x): the resident pipeline's haplotypes, genotype
    calls and read bipartition against the oracle, next to a second full-size chunk sharing the launches."""
    chunks = [synth.make_ont_chunk(seed=s, region_bp=1_000_000, n_sites=2000, coverage=30.0) for s in (1, 2)]
    pd = _params()
    params = capi.Params.from_reference_names(pd)
    dchunks = [capi.DeviceChunk.from_chunk(gpu_ctx, c) for c in chunks]
    got, st = capi.phase_reads_many(gpu_ctx, dchunks, chunks, params)
    assert st.resident == 1 and st.cells > 10_000_000
    oc = orc.OracleChunk(chunks[0])
    ref = oc.phase(pd)
    oc.close()
    for k in PHASE_KEYS:
        assert (np.asarray(got[0][k]) == np.asarray(ref[k])).all(), k
    assert got[0]["reads1"] == ref["reads1"] and got[0]["reads2"] == ref["reads2"]
    assert got[0]["n_sweeps"] == ref["fb_calls"]
    # size-independent property on the second chunk: hap2 is the complement call wherever the genotype is heterozygous
    g = got[1]
    het = np.asarray(g["hap1"]) != np.asarray(g["hap2"])
    assert het.mean() > 0.9 and len(set(g["reads1"]) & set(g["reads2"])) == 0
    assert len(g["reads1"]) + len(g["reads2"]) == len(chunks[1].reads)
    for d in dchunks:
        d.close()


def test_resident_hifi_shape_multiallelic(gpu_ctx, orc):
    """BASELINE.json configs[4] shape: 35x reads of ~18 kb, 1 % allele error, sites with 2, 3 or 4 alleles
    (indel-like variants of phase_vcf mode) -- columns mix allele counts, so merge levels take the general emission path
    and the final sweep the ancestor model over up to 4 alleles.  Resident pipeline == oracle."""
    chunk = synth.make_ont_chunk(seed=51, region_bp=150_000, n_sites=300, coverage=35.0, median_len=18_000.0, allele_error=0.01,
                                 allele_choices=(2, 3, 4), allele_probs=(0.85, 0.1, 0.05), length_model="normal", normal_sd=3000.0)
    assert set(np.unique(chunk.allele_number)) == {2, 3, 4}
    pd = _params()
    oc = orc.OracleChunk(chunk)
    ref = oc.phase(pd)
    oc.close()
    dchunk = capi.DeviceChunk.from_chunk(gpu_ctx, chunk)
    (got,), st = capi.phase_reads_many(gpu_ctx, [dchunk], [chunk], capi.Params.from_reference_names(pd))
    assert st.resident == 1
    for k in PHASE_KEYS:
        assert (np.asarray(got[k]) == np.asarray(ref[k])).all(), k
    assert got["reads1"] == ref["reads1"] and got["reads2"] == ref["reads2"]
    assert got["n_sweeps"] == ref["fb_calls"]
    # the phasing recovers the simulated haplotypes up to the global label
    h1, t1, t2 = np.asarray(got["hap1"]).astype(np.int64), chunk.hap1[got["ref_start"]:got["ref_start"] + got["length"]], \
        chunk.hap2[got["ref_start"]:got["ref_start"] + got["length"]]
    assert max((h1 == t1).mean(), (h1 == t2).mean()) > 0.95
    dchunk.close()


def test_resident_depth_64_columns(gpu_ctx, orc):
    """Coverage beyond maxCoverageDepth = 64: reads are filtered down to 64 tiling paths (coordination.c:443-488) and the
    top merge levels reach columns with 49..64 reads -- the widest emission variants (16 words per partition), 64-bit
    partitions with the top bit in use, accept masks of depth 64."""
    chunk = synth.make_ont_chunk(seed=61, region_bp=30_000, n_sites=60, coverage=95.0, median_len=9_000.0, sigma=0.3)
    pd = _params(maxCoverageDepth=64)
    oc = orc.OracleChunk(chunk)
    ref = oc.phase(pd, capture_jobs=True)
    oc.close()
    deepest = max(int(np.max(j["col_depth"])) for j in ref["jobs"])
    assert deepest > 48, deepest
    dchunk = capi.DeviceChunk.from_chunk(gpu_ctx, chunk)
    params = capi.Params.from_reference_names(pd)
    (got,), st = capi.phase_reads_many(gpu_ctx, [dchunk], [chunk], params)
    assert st.resident == 1
    for k in PHASE_KEYS:
        assert (np.asarray(got[k]) == np.asarray(ref[k])).all(), k
    assert got["reads1"] == ref["reads1"] and got["reads2"] == ref["reads2"]
    # and the same sweeps through the forward/backward seam (the oracle's own jobs)
    from tests.helpers import assert_job_equal, run_jobs_on_gpu
    deep = [j for j in ref["jobs"] if int(np.max(j["col_depth"])) > 48][:4]
    for f, r in zip(deep, run_jobs_on_gpu(gpu_ctx, dchunk, deep)):
        assert_job_equal(f, r, exact=True)
    dchunk.close()


def _subset_chunk(chunk, keep):
    """the same sites and pool with a subset of the reads (pool offsets stay valid)"""
    return synth.Chunk(allele_number=chunk.allele_number, allele_offset=chunk.allele_offset, sub=chunk.sub, prior=chunk.prior,
                       pool=chunk.pool, reads=[chunk.reads[i] for i in keep])


def test_resident_ragged_inputs(gpu_ctx, orc):
    """Edge shapes in one call: a single read, one strand only, two reads that do not overlap (gap columns of
    stRPHmm_fuse, hmm.c:335-359), reads of one site, and a dense chunk next to them."""
    base = synth.make_ont_chunk(seed=71, region_bp=60_000, n_sites=120, coverage=25)
    reads = base.reads
    fwd = [i for i, r in enumerate(reads) if r.strand == 1]
    far = sorted(range(len(reads)), key=lambda i: reads[i].ref_start)
    a = far[0]
    b = next(i for i in far if reads[i].ref_start >= reads[a].ref_start + reads[a].length + 3)
    one_site = [i for i, r in enumerate(reads) if r.length == 1][:3]
    cases = [_subset_chunk(base, [0]), _subset_chunk(base, fwd), _subset_chunk(base, [a, b]),
             _subset_chunk(base, one_site + [a]) if one_site else _subset_chunk(base, [a]), base]
    pd = _params()
    params = capi.Params.from_reference_names(pd)
    dchunks = [capi.DeviceChunk.from_chunk(gpu_ctx, c) for c in cases]
    got, st = capi.phase_reads_many(gpu_ctx, dchunks, cases, params)
    assert st.resident == 1
    for c, g in zip(cases, got):
        oc = orc.OracleChunk(c)
        ref = oc.phase(pd)
        oc.close()
        assert g["ref_start"] == ref["ref_start"] and g["length"] == ref["length"]
        for k in PHASE_KEYS:
            assert (np.asarray(g[k]) == np.asarray(ref[k])).all(), (len(c.reads), k)
        assert g["reads1"] == ref["reads1"] and g["reads2"] == ref["reads2"], len(c.reads)
        assert g["n_sweeps"] == ref["fb_calls"]
    for d in dchunks:
        d.close()


def test_resident_many_small_chunks_do_not_depend_on_their_batch(gpu_ctx, orc):
    """Config-3 shape (chr20 cut into 100 kb chunks of ~130 sites): 150 chunks in one call.  The merge trees have different
    heights, so the level schedule (every root at the last level) mixes leaf merges of deep problems with larger merges of
    shallow ones.  Size-independent property: a chunk's result does not depend on what else is in the call -- the whole
    batch against the same chunks phased in three smaller calls in another order; a sample against the oracle."""
    rng = np.random.default_rng(3)
    chunks = []
    for s in range(150):
        n = int(rng.integers(60, 200))
        chunks.append(synth.make_ont_chunk(seed=1000 + s, region_bp=100_000, n_sites=n, coverage=float(rng.integers(12, 45))))
    pd = _params()
    params = capi.Params.from_reference_names(pd)
    dchunks = [capi.DeviceChunk.from_chunk(gpu_ctx, c) for c in chunks]
    got, st = capi.phase_reads_many(gpu_ctx, dchunks, chunks, params)
    assert st.resident == 1 and len(got) == 150
    order = rng.permutation(150)
    again = [None] * 150
    for part in (order[:40], order[40:110], order[110:]):
        res, st2 = capi.phase_reads_many(gpu_ctx, [dchunks[i] for i in part], [chunks[i] for i in part], params)
        assert st2.resident == 1
        for i, r in zip(part, res):
            again[i] = r
    for a, b in zip(got, again):
        for k in PHASE_KEYS:
            assert (np.asarray(a[k]) == np.asarray(b[k])).all(), k
        assert a["reads1"] == b["reads1"] and a["reads2"] == b["reads2"] and a["n_sweeps"] == b["n_sweeps"]
    for i in order[:6]:
        oc = orc.OracleChunk(chunks[i])
        ref = oc.phase(pd)
        oc.close()
        for k in PHASE_KEYS:
            assert (np.asarray(got[i][k]) == np.asarray(ref[k])).all(), k
        assert got[i]["reads1"] == ref["reads1"] and got[i]["reads2"] == ref["reads2"]
    for d in dchunks:
        d.close()


def test_resident_pruning_rules_over_random_parameters(gpu_ctx, orc):
    """hmm.c:1049-1163 with thresholds that bite: random (min, max) partitions per column and posterior thresholds from 0 to
    0.2, on chunks of random shape, all in ONE call per parameter set.  The prune kernel flags every next merge cell of the
    kept cells before it knows their order (its second stage checks that hmm.c:1090-1100 would keep them all and discards the
    level otherwise): the call must stay on the resident path and equal the oracle, array for array."""
    rng = np.random.default_rng(123)
    for trial in range(6):
        max_p = int(rng.choice([16, 40, 64, 100, 116]))
        min_p = int(rng.integers(0, max_p + 1)) if trial % 2 else 0
        thr = float(rng.choice([0.0, 1e-9, 1e-3, 0.02, 0.2]))
        pd = _params(minPartitionsInAColumn=min_p, maxPartitionsInAColumn=max_p, minPosteriorProbabilityForPartition=thr)
        params = capi.Params.from_reference_names(pd)
        chunks = [synth.make_ont_chunk(seed=700 + 10 * trial + c, region_bp=40_000, n_sites=int(rng.integers(30, 110)),
                                       coverage=float(rng.integers(10, 40)), allele_error=float(rng.choice([0.02, 0.08, 0.15])))
                  for c in range(5)]
        dchunks = [capi.DeviceChunk.from_chunk(gpu_ctx, c) for c in chunks]
        got, st = capi.phase_reads_many(gpu_ctx, dchunks, chunks, params)
        assert st.resident == 1, (min_p, max_p, thr)
        for chunk, g in zip(chunks, got):
            oc = orc.OracleChunk(chunk)
            ref = oc.phase(pd)
            oc.close()
            for k in PHASE_KEYS:
                assert (np.asarray(g[k]) == np.asarray(ref[k])).all(), (k, min_p, max_p, thr)
            assert g["reads1"] == ref["reads1"] and g["reads2"] == ref["reads2"]
        for d in dchunks:
            d.close()


def _assert_equals_oracle(orc, chunk, got, pd):
    oc = orc.OracleChunk(chunk)
    ref = oc.phase(pd)
    oc.close()
    for k in PHASE_KEYS:
        assert (np.asarray(got[k]) == np.asarray(ref[k])).all(), k
    assert got["reads1"] == ref["reads1"] and got["reads2"] == ref["reads2"]


def test_resident_pair_order_violation_sends_only_that_chunk_to_the_hashing_path(gpu_ctx, orc):
    """The closed-form cross product needs its parents in complement-pair order.  An ODD maxPartitionsInAColumn makes the
    prune cut through a (partition, complement) pair wherever a column has more linked cells than that, so the next level's
    cross kernel raises MRP_ENGINE_ERR_STRUCTURE for the hmms of that chunk: exactly those chunks are redone on the hashing
    path (stats.fallback_chunks), the shallow chunk of the same call -- at most five reads deep, so no column reaches 51 cells -- stays
    resident, and every result equals the oracle's."""
    pd = _params(minPartitionsInAColumn=0, maxPartitionsInAColumn=51)
    params = capi.Params.from_reference_names(pd)
    chunks = [synth.make_ont_chunk(seed=81, region_bp=60_000, n_sites=120, coverage=30),
              synth.make_ont_chunk(seed=82, region_bp=40_000, n_sites=80, coverage=2),
              synth.make_ont_chunk(seed=83, region_bp=50_000, n_sites=100, coverage=25)]
    dchunks = [capi.DeviceChunk.from_chunk(gpu_ctx, c) for c in chunks]
    got, st = capi.phase_reads_many(gpu_ctx, dchunks, chunks, params)
    assert st.resident == 1 and st.fallback_chunks == 2
    for chunk, g in zip(chunks, got):
        _assert_equals_oracle(orc, chunk, g, pd)
    # a single chunk through the resident getRPHmms reports the same condition as a status
    with pytest.raises(capi.MrpError) as ei:
        capi.get_rp_hmms_resident(gpu_ctx, dchunks[0], chunks[0], capi.Params.from_reference_names(dict(pd, includeAncestorSubProb=0)),
                                  [i for i, r in enumerate(chunks[0].reads) if r.strand == 1])
    assert ei.value.code == capi.MRP_ERR_UNSUPPORTED
    for d in dchunks:
        d.close()


def test_resident_merge_check_failure_sends_only_that_chunk_to_the_hashing_path(gpu_ctx, orc):
    """MRP_ENGINE_ERR_MERGE -- "a merge cell the kept cells lead to would itself be pruned" (hmm.c:1090-1100) -- cannot be
    produced by an input in max-plus mode (a merge cell's posterior is at least that of every cell leading to it, so
    whenever more than minPartitionsInAColumn cells are kept they and their merge cells all pass the threshold); the
    kernel checks it all the same.  The check's way out is exercised by fault injection (mrp_context_set_test_hooks bit 0: one
    hmm of the second level reports the error): that hmm's chunk, and only it, is redone on the hashing path; results are
    the oracle's."""
    pd = _params()
    params = capi.Params.from_reference_names(pd)
    gpu_ctx.set_test_hooks(1)
    chunks = [synth.make_ont_chunk(seed=91 + i, region_bp=50_000, n_sites=100, coverage=20 + 4 * i) for i in range(4)]
    dchunks = [capi.DeviceChunk.from_chunk(gpu_ctx, c) for c in chunks]
    got, st = capi.phase_reads_many(gpu_ctx, dchunks, chunks, params)
    assert st.resident == 1 and st.fallback_chunks == 1
    for chunk, g in zip(chunks, got):
        _assert_equals_oracle(orc, chunk, g, pd)
    gpu_ctx.set_test_hooks(0)
    _, st0 = capi.phase_reads_many(gpu_ctx, dchunks, chunks, params)
    assert st0.resident == 1 and st0.fallback_chunks == 0
    params.reserved = 1  # a parameter struct with a stray reserved field is refused, not interpreted
    with pytest.raises(capi.MrpError):
        capi.phase_reads_many(gpu_ctx, dchunks, chunks, params)
    for d in dchunks:
        d.close()


@pytest.mark.parametrize("case", ["biallelic", "mixed_alleles_long_reads", "many_alleles"])
def test_one_pass_cross_emission_equals_the_two_kernel_path_and_the_oracle(gpu_ctx, orc, case):
    """Merge levels compute a cell's emission from per-parent-cell tables inside the cross product kernel and write no
    partition (emissions.c:125-154 is a sum over the reads of the partition, partitions.c:21-28 concatenates the two
    parents' reads).  Same chunks through the separate cross product + emission kernels (mrp_context_set_test_hooks bit 1): both
    equal the oracle.  The cases cover sites with different allele counts inside one column, columns whose sites do not
    fit one table fill (long reads: hundreds of allele slots per column at the low levels) and a site with more alleles
    than the tables hold for wide columns (that chunk takes the hashing path)."""
    pd = _params()
    if case == "biallelic":
        chunks = [synth.make_ont_chunk(seed=301 + i, region_bp=60_000, n_sites=130, coverage=22 + 3 * i) for i in range(3)]
    elif case == "mixed_alleles_long_reads":
        chunks = [synth.make_ont_chunk(seed=311 + i, region_bp=150_000, n_sites=700, coverage=18, median_len=60_000.0, sigma=0.3,
                                       allele_choices=(2, 3, 4, 6), allele_probs=(0.4, 0.3, 0.2, 0.1)) for i in range(2)]
    else:
        chunks = [synth.make_ont_chunk(seed=321, region_bp=60_000, n_sites=120, coverage=24, allele_choices=(2, 5, 12, 16),
                                       allele_probs=(0.55, 0.25, 0.15, 0.05))]
    dchunks = [capi.DeviceChunk.from_chunk(gpu_ctx, c) for c in chunks]
    params = capi.Params.from_reference_names(pd)
    got1, st1 = capi.phase_reads_many(gpu_ctx, dchunks, chunks, params)
    gpu_ctx.set_test_hooks(2)
    try:
        got2, st2 = capi.phase_reads_many(gpu_ctx, dchunks, chunks, params)
    finally:
        gpu_ctx.set_test_hooks(0)
    assert st1.resident == 1 and st2.resident == 1 and st2.fallback_chunks == 0
    if case != "many_alleles":
        assert st1.fallback_chunks == 0
    for chunk, a, b in zip(chunks, got1, got2):
        _assert_equals_oracle(orc, chunk, a, pd)
        _assert_equals_oracle(orc, chunk, b, pd)
    for d in dchunks:
        d.close()


@pytest.mark.parametrize("knob", [{"MRP_POOL_BUDGET_MB": "64"}, {"MRP_CALL_UNITS": "40000"}], ids=["pool_budget_64MB", "call_sliced_at_40000_units"])
def test_device_memory_budget_gives_cached_blocks_back_without_changing_results(gpu_ctx, orc, knob):
    """The device pools of a process share one budget per device (mrp_internal.h DevPoolRegistry; MRP_POOL_BUDGET_MB, read once
    per process, hence a child process): with a budget (64 MB) below what even the live arrays of the call need, every
    reclaim gives blocks back to the driver and the next level allocates afresh -- results must not move, across repeated calls
    with different chunk subsets (best-fit reuse of blocks of other sizes) and through the work queue.  Second knob: a call whose
    (read, site) units exceed what the device's budget holds runs as consecutive slices (MRP_CALL_UNITS: here ~8 slices)."""
    import json, subprocess, sys
    code = r'''
import json, sys
import numpy as np
from margin_amd import capi, synth
pd = synth.shipped_phase_params()
params = capi.Params.from_reference_names(pd)
chunks = [synth.make_ont_chunk(seed=700 + s, region_bp=100_000, n_sites=int(90 + 7 * s), coverage=20.0 + s) for s in range(24)]
def key(r):
    return [[int(x) for x in np.asarray(r[k]).tolist()] for k in ("hap1", "hap2", "genotype", "support1", "support2")] + [r["reads1"], r["reads2"]]
ctx = capi.Context(0)
ctx.set_phase_groups(4)
dch = [capi.DeviceChunk.from_chunk(ctx, c) for c in chunks]
out = []
for lo, hi in ((0, 24), (3, 19), (0, 24)):
    res, st = capi.phase_reads_many(ctx, dch[lo:hi], chunks[lo:hi], params)
    assert st.resident == 1
    out.append([key(r) for r in res])
assert out[0] == out[2] and out[0][3:19] == out[1]
q = capi.Queue([0])
qres, _ = q.phase(chunks, params, chunks_per_batch=0)
q.close()
assert [key(r) for r in qres] == out[0]
print(json.dumps(out[0][:4]))
'''
    env = dict(os.environ, MRP_QUIET="1", **knob)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    small_budget = json.loads(r.stdout.strip().splitlines()[-1])
    pd = _params()
    for s in range(4):
        c = synth.make_ont_chunk(seed=700 + s, region_bp=100_000, n_sites=int(90 + 7 * s), coverage=20.0 + s)
        oc = orc.OracleChunk(c)
        ref = oc.phase(pd)
        oc.close()
        assert small_budget[s] == [[int(x) for x in np.asarray(ref[k]).tolist()] for k in ("hap1", "hap2", "genotype", "support1", "support2")] + [ref["reads1"], ref["reads2"]]


def test_a_refused_device_allocation_redoes_the_call_in_two_halves(orc):
    """the pools' budget is an estimate (free memory at first use, ~3.4 KB per unit); when the driver refuses
    an allocation all the same, mrp_phase_reads_many gives its caches back and redoes the call as two slices instead of failing.
    The refusal is injected (mrp_context_set_test_hooks bit 2: the next allocation of at least 1 MB is refused once); the
    results are those of the undisturbed call and of the oracle."""
    pd = _params()
    params = capi.Params.from_reference_names(pd)
    chunks = [synth.make_ont_chunk(seed=900 + s, region_bp=150_000, n_sites=int(110 + 9 * s), coverage=22.0 + s) for s in range(6)]
    with capi.Context(0) as ctx:
        ctx.set_phase_groups(1)
        dch = [capi.DeviceChunk.from_chunk(ctx, c) for c in chunks]
        ref, st0 = capi.phase_reads_many(ctx, dch, chunks, params)
        assert st0.resident == 1
        ctx.set_test_hooks(4)
        got, st = capi.phase_reads_many(ctx, dch, chunks, params)
        assert st.resident == 1 and st.hmms == st0.hmms and st.cells == st0.cells  # (two slices: the same hmms, summed)
        for a, b in zip(ref, got):
            for k in PHASE_KEYS:
                assert (np.asarray(a[k]) == np.asarray(b[k])).all(), k
            assert a["reads1"] == b["reads1"] and a["reads2"] == b["reads2"]
        oc = orc.OracleChunk(chunks[2])
        o = oc.phase(pd)
        oc.close()
        for k in ("hap1", "hap2", "genotype", "support1", "support2"):
            assert (np.asarray(got[2][k]) == np.asarray(o[k])).all(), k
        for d in dch:
            d.close()


def test_bench_shape_configs2_640_small_chunks_sampled_against_the_oracle(orc):
    """bench.py's configs[2] leg at ITS size: 640 chunks of ~130 het sites (chr20 cut into 100 kb chunks) in one
    mrp_phase_reads_many call with eight concurrent batches, 16 of the chunks against the oracle (every 40th: each batch of
    the call is sampled twice), all of them by size-independent properties."""
    n = 640
    chunks = [synth.make_ont_chunk(seed=50_000 + s, region_bp=130 * 500, n_sites=130, coverage=30.0) for s in range(n)]
    pd = _params()
    params = capi.Params.from_reference_names(pd)
    with capi.Context(0) as ctx:
        dch = [capi.DeviceChunk.from_chunk(ctx, c) for c in chunks]
        got, st = capi.phase_reads_many(ctx, dch, chunks, params)
        assert st.resident == 1 and st.fallback_chunks == 0
        for i in range(0, n, 40):
            oc = orc.OracleChunk(chunks[i])
            ref = oc.phase(pd)
            oc.close()
            for k in PHASE_KEYS:
                assert (np.asarray(got[i][k]) == np.asarray(ref[k])).all(), (i, k)
            assert got[i]["reads1"] == ref["reads1"] and got[i]["reads2"] == ref["reads2"], i
        for c, g in zip(chunks, got):
            assert len(set(g["reads1"]) & set(g["reads2"])) == 0 and len(g["reads1"]) + len(g["reads2"]) == len(c.reads)
        for d in dch:
            d.close()


def test_bench_shape_configs4_full_size_hifi_chunk_equals_the_oracle(gpu_ctx, orc):
    """bench.py's configs[4] leg at ITS chunk size: ONE HiFi-like chunk of 2 000 sites (35x reads N(18 kb, 3 kb), 1 % allele error,
    2-4 alleles per site) against the oracle, array for array."""
    chunk = synth.make_ont_chunk(seed=60_000, region_bp=2000 * 500, n_sites=2000, coverage=35.0, median_len=18_000.0, allele_error=0.01,
                                 allele_choices=(2, 3, 4), allele_probs=(0.85, 0.1, 0.05), length_model="normal", normal_sd=3000.0)
    pd = _params()
    oc = orc.OracleChunk(chunk)
    ref = oc.phase(pd)
    oc.close()
    dchunk = capi.DeviceChunk.from_chunk(gpu_ctx, chunk)
    (got,), st = capi.phase_reads_many(gpu_ctx, [dchunk], [chunk], capi.Params.from_reference_names(pd))
    assert st.resident == 1 and st.fallback_chunks == 0
    for k in PHASE_KEYS:
        assert (np.asarray(got[k]) == np.asarray(ref[k])).all(), k
    assert got["reads1"] == ref["reads1"] and got["reads2"] == ref["reads2"]
    assert got["n_sweeps"] == ref["fb_calls"]
    dchunk.close()


@pytest.mark.parametrize("hooks,env", [(8, {}), (0, {"MRP_UNITS": "0"})], ids=["general_prune_chain", "pairs_chain_on_cell_arrays"])
def test_prune_variants_agree_with_the_default_path_and_the_oracle(orc, hooks, env):
    """The default resident path runs the prune chain on complement pairs over arrays that hold one entry per pair (MRP_XF_UNITS).
    Two variants stay in the product (odd column limits / plain mode take the first, the two-kernel cross product + emission
    path the second): the general chain with one entry per cell (test hook bit 3) and the pair chain over per-cell arrays
    (MRP_UNITS=0, read at every level).  Both give the default path's results, which are the oracle's."""
    pd = _params()
    params = capi.Params.from_reference_names(pd)
    chunks = [synth.make_ont_chunk(seed=820 + s, region_bp=120_000 + 40_000 * s, n_sites=240 + 80 * s, coverage=28.0 + 4 * s) for s in range(3)]
    old = {k: os.environ.get(k) for k in env}
    with capi.Context(0) as ctx:
        dch = [capi.DeviceChunk.from_chunk(ctx, c) for c in chunks]
        ref, st0 = capi.phase_reads_many(ctx, dch, chunks, params)
        try:
            os.environ.update(env)
            ctx.set_test_hooks(hooks)
            got, st = capi.phase_reads_many(ctx, dch, chunks, params)
        finally:
            ctx.set_test_hooks(0)
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        assert st.resident == 1 and st.fallback_chunks == 0 and st.cells == st0.cells and st.merge_cells == st0.merge_cells
        for a, b in zip(ref, got):
            for k in PHASE_KEYS:
                assert (np.asarray(a[k]) == np.asarray(b[k])).all(), k
            assert a["reads1"] == b["reads1"] and a["reads2"] == b["reads2"]
        oc = orc.OracleChunk(chunks[1])
        o = oc.phase(pd)
        oc.close()
        for k in PHASE_KEYS:
            assert (np.asarray(got[1][k]) == np.asarray(o[k])).all(), k
        for d in dch:
            d.close()


def test_odd_column_limits_take_the_general_prune_chain_and_equal_the_oracle(gpu_ctx, orc):
    """maxPartitionsInAColumn = 51 (odd): a selection may cut a complement pair in two, so the engine keeps the general prune chain
    (PruneParams.pairs = 0).  Where a pair IS cut, the next level's cross product finds a parent outside the pair order and that
    chunk takes the hashing path (stats.fallback_chunks); either way the results are the oracle's."""
    chunk = synth.make_ont_chunk(seed=877, region_bp=90_000, n_sites=180, coverage=30.0)
    pd = _params(minPartitionsInAColumn=7, maxPartitionsInAColumn=51)
    oc = orc.OracleChunk(chunk)
    ref = oc.phase(pd)
    oc.close()
    dchunk = capi.DeviceChunk.from_chunk(gpu_ctx, chunk)
    (got,), st = capi.phase_reads_many(gpu_ctx, [dchunk], [chunk], capi.Params.from_reference_names(pd))
    assert st.resident == 1 and st.fallback_chunks in (0, 1)
    for k in PHASE_KEYS:
        assert (np.asarray(got[k]) == np.asarray(ref[k])).all(), k
    assert got["reads1"] == ref["reads1"] and got["reads2"] == ref["reads2"]
    dchunk.close()


def test_results_do_not_depend_on_the_host_threads(gpu_ctx, orc):
    """The host side of a level runs on a pool whose threads start every loop with a range of their own, keep per-thread fronts of
    the block pool and give the parents' blocks back as they merge them.  Size-independent property: the same 64 chunks (eight
    concurrent batches) phased with 1, 3 and 16 pool threads give the same results, array for array; a sample against the oracle."""
    chunks = [synth.make_ont_chunk(seed=7000 + s, region_bp=150_000, n_sites=300, coverage=30.0) for s in range(64)]
    pd = _params()
    params = capi.Params.from_reference_names(pd)
    dchunks = [capi.DeviceChunk.from_chunk(gpu_ctx, c) for c in chunks]
    lib = capi.load()
    results = []
    try:
        for threads in (1, 3, 16):
            lib.mrp_set_host_threads(threads)
            for _ in range(2):  # (the second call of a setting runs on the blocks the first one left in the pool)
                got, st = capi.phase_reads_many(gpu_ctx, dchunks, chunks, params)
            assert st.resident == 1 and st.fallback_chunks == 0
            results.append(got)
    finally:
        lib.mrp_set_host_threads(16)
    for other in results[1:]:
        for a, b in zip(results[0], other):
            for k in PHASE_KEYS:
                assert (np.asarray(a[k]) == np.asarray(b[k])).all(), k
            assert a["reads1"] == b["reads1"] and a["reads2"] == b["reads2"] and a["n_sweeps"] == b["n_sweeps"]
    for i in (0, 31, 63):
        oc = orc.OracleChunk(chunks[i])
        ref = oc.phase(pd)
        oc.close()
        for k in PHASE_KEYS:
            assert (np.asarray(results[0][i][k]) == np.asarray(ref[k])).all(), k
        assert results[0][i]["reads1"] == ref["reads1"] and results[0][i]["reads2"] == ref["reads2"]
    for d in dchunks:
        d.close()


def test_results_do_not_depend_on_the_batches_shares(gpu_ctx, orc, monkeypatch):
    """A call of many large chunks deals them to its concurrent batches in graded shares (2 : 3 : 4 : 5 : 5 ..., the first batch the
    smallest, rphmm_host.c), small chunks in equal shares; MRP_GROUP_WEIGHTS sets the shares.  Size-independent property: 160 small
    chunks in eight batches give the same results, chunk for chunk, with equal shares, with the graded ones and with reversed ones;
    a sample against the oracle."""
    chunks = [synth.make_ont_chunk(seed=9100 + s, region_bp=40_000, n_sites=80, coverage=30.0) for s in range(160)]
    pd = _params()
    params = capi.Params.from_reference_names(pd)
    dchunks = [capi.DeviceChunk.from_chunk(gpu_ctx, c) for c in chunks]
    gpu_ctx.set_phase_groups(8)
    results = []
    try:
        for shares in (None, "2:3:4:5:5:5:5:5", "5:5:5:5:4:3:2:1"):
            if shares is None:
                monkeypatch.delenv("MRP_GROUP_WEIGHTS", raising=False)
            else:
                monkeypatch.setenv("MRP_GROUP_WEIGHTS", shares)
            got, st = capi.phase_reads_many(gpu_ctx, dchunks, chunks, params)
            assert st.resident == 1 and st.fallback_chunks == 0
            results.append(got)
    finally:
        gpu_ctx.set_phase_groups(0)
    for other in results[1:]:
        for a, b in zip(results[0], other):
            for k in PHASE_KEYS:
                assert (np.asarray(a[k]) == np.asarray(b[k])).all(), k
            assert a["reads1"] == b["reads1"] and a["reads2"] == b["reads2"] and a["n_sweeps"] == b["n_sweeps"]
    for i in (0, 1, 79, 158, 159):
        oc = orc.OracleChunk(chunks[i])
        ref = oc.phase(pd)
        oc.close()
        for k in PHASE_KEYS:
            assert (np.asarray(results[0][i][k]) == np.asarray(ref[k])).all(), k
        assert results[0][i]["reads1"] == ref["reads1"] and results[0][i]["reads2"] == ref["reads2"]
    for d in dchunks:
        d.close()
