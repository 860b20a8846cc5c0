"""GPU parity of the ancestor substitution model (impl/emissions.c:156-172 ancestorHapProbabilities, :209-218 the
includeAncestorSubProb branch of genotypeLogProbability) with NON-zero substitution and prior tables.

The shipped parameters have hetSubstitutionProbability = 0, which turns every substitutionLogProbs entry into 0
(bubbleGraph.c:2466-2468), so the indexing sub[i * A + k] / prior[i] of the HIP path is only exercised by the inputs of
this file: random uint16 tables at the emission seam, at the forward/backward seam (wide cross product columns
included) and through the whole resident pipeline with tables made by mrp_reference_from_bubbles from a positive
hetSubstitutionProbability."""
import ctypes as C

import numpy as np
import pytest

from margin_amd import capi, synth
from tests.helpers import assert_job_equal, run_jobs_on_gpu

pytestmark = pytest.mark.gpu

PHASE_KEYS = ("hap1", "hap2", "genotype", "ancestor", "support1", "support2", "genotype_probs", "hap_probs1", "hap_probs2")
ANC = capi.FLAG_INCLUDE_ANCESTOR_SUB_PROB


def _tables(rng, A, sub_hi=400, prior_hi=80, asymmetric=True):
    """random substitutionLogProbs ([from * A + to], zero diagonal as -log(1 - het) * 30 rounds to) and allelePriorLogProbs"""
    sub, prior = [], []
    for a in A:
        a = int(a)
        s = rng.integers(1, sub_hi, size=(a, a))
        if not asymmetric:
            s = np.minimum(s, s.T)
        s[np.arange(a), np.arange(a)] = rng.integers(0, 3, size=a)
        sub.append(s.reshape(-1))
        prior.append(rng.integers(0, prior_hi, size=a))
    return np.concatenate(sub).astype(np.uint16), np.concatenate(prior).astype(np.uint16)


def test_emissions_kat_with_ancestor_tables(gpu_ctx, orc):
    """mrp_emissions with MRP_FLAG_INCLUDE_ANCESTOR_SUB_PROB against the definition evaluated directly
    (min_i (min_k h1[k] + sub[i][k]) + (min_k h2[k] + sub[i][k]) + prior[i], emissions.c:156-172,209-218) and against
    the oracle, depth 0..64, 1..9 alleles per site, asymmetric tables (a transposed index would show)."""
    L = orc.lib()
    rng = np.random.default_rng(4242)
    for depth in list(range(0, 64, 5)) + [31, 32, 33, 63, 64]:
        n_sites = int(rng.integers(1, 8))
        A = rng.integers(1, 10, size=n_sites).astype(np.uint32)
        off = np.concatenate([[0], np.cumsum(A)]).astype(np.int64)
        total = int(off[-1])
        sub, prior = _tables(rng, A)
        rows = rng.integers(0, 256, size=(max(depth, 1), total)).astype(np.uint8)
        dchunk = capi.DeviceChunk(gpu_ctx, A, sub, prior, rows.reshape(-1))
        byte_off = (np.arange(depth) * total).astype(np.int64)
        mask = (1 << depth) - 1
        parts = np.array([int(rng.integers(0, 2**63)) & mask for _ in range(61)] + [0, mask, mask >> 1], dtype=np.uint64)
        got_anc = capi.emissions(gpu_ctx, dchunk, 0, n_sites, byte_off, ANC, parts)
        got_plain = capi.emissions(gpu_ctx, dchunk, 0, n_sites, byte_off, 0, parts)
        ref = L.orc_reference_create(b"ref", n_sites, A.ctypes.data, sub.ctypes.data, prior.ctypes.data)
        rws = [np.ascontiguousarray(rows[i]) for i in range(depth)]
        ptrs = (C.c_void_p * max(depth, 1))(*[r.ctypes.data for r in rws])
        for p, ga, gp in zip(parts, got_anc, got_plain):
            sel = np.array([(int(p) >> i) & 1 for i in range(depth)], dtype=bool)
            h1 = rows[:depth][sel].astype(np.int64).sum(axis=0) if depth else np.zeros(total, np.int64)
            h2 = rows[:depth][~sel].astype(np.int64).sum(axis=0) if depth else np.zeros(total, np.int64)
            anc, plain, so = 0, 0, 0
            for s in range(n_sites):
                a = int(A[s])
                x, y = h1[off[s]:off[s + 1]], h2[off[s]:off[s + 1]]
                S = sub[so:so + a * a].reshape(a, a).astype(np.int64)
                anc += int(((x[None, :] + S).min(axis=1) + (y[None, :] + S).min(axis=1) + prior[off[s]:off[s + 1]].astype(np.int64)).min())
                plain += int(x.min()) + int(y.min())
                so += a * a
            assert ga == -float(anc) and gp == -float(plain)
            assert ga == L.orc_emission_raw(ptrs, ref, 0, n_sites, depth, int(p), 1)
        L.orc_reference_destroy(ref)
        dchunk.close()


def _chunk_with_tables(seed, n_sites, coverage, **kw):
    chunk = synth.make_ont_chunk(seed=seed, region_bp=n_sites * 500, n_sites=n_sites, coverage=coverage, **kw)
    rng = np.random.default_rng([seed, 99])
    chunk.sub, chunk.prior = _tables(rng, chunk.allele_number, sub_hi=300, prior_hi=70)
    return chunk


def _oracle_jobs(orc, oc, pd, idx=None):
    """every forward/backward the oracle issues inside getRPHmms (coordination.c:312) with the parameters as given:
    includeAncestorSubProb = 1 puts the ancestor model on the wide cross product columns of every merge level"""
    L = orc.lib()
    jobs = []

    def _obs(hmm_ptr, _user):
        j = orc.flatten(hmm_ptr, oc.pool_off)
        pr = C.cast(hmm_ptr.contents.parameters, C.POINTER(orc.Params)).contents
        j["flags"] = (1 if pr.maxNotSumTransitions else 0) | (2 if pr.includeAncestorSubProb else 0)
        jobs.append(j)

    cb = orc.FB_OBSERVER(_obs)
    L.orc_set_fb_observer(cb, None)
    try:
        hmms = oc.get_rp_hmms(orc.make_params(pd), idx)
    finally:
        L.orc_set_fb_observer(C.cast(None, orc.FB_OBSERVER), None)
    return jobs, hmms


@pytest.mark.parametrize("seed,alleles,probs", [(71, (2,), (1.0,)), (72, (2, 3, 4), (0.6, 0.25, 0.15))])
def test_forward_backward_with_ancestor_tables_every_level(gpu_ctx, orc, seed, alleles, probs):
    """stRPHmm_forwardBackward (hmm.c:931) with includeAncestorSubProb = 1 and non-zero tables at EVERY merge level
    (columns of up to 10 000 cells), bit-exact against the oracle; both the pre-resolved-index and the key-resolved form."""
    chunk = _chunk_with_tables(seed, 160, 30, allele_choices=alleles, allele_probs=probs)
    pd = synth.shipped_phase_params()
    pd["includeAncestorSubProb"] = 1
    oc = orc.OracleChunk(chunk)
    jobs, _ = _oracle_jobs(orc, oc, pd, [i for i, r in enumerate(chunk.reads) if r.strand == 1])
    assert all(j["flags"] == 3 for j in jobs) and max(int(np.diff(j["col_cell_off"]).max()) for j in jobs) >= 2500
    dchunk = capi.DeviceChunk.from_chunk(gpu_ctx, chunk)
    out = run_jobs_on_gpu(gpu_ctx, dchunk, jobs)
    for f, r in zip(jobs, out):
        assert_job_equal(f, r, exact=True)
    # the tables matter on this input: the same jobs without the ancestor model give other values somewhere
    out0 = run_jobs_on_gpu(gpu_ctx, dchunk, jobs[-3:], flags_override=1)
    assert any((np.asarray(a["cell_forward"]) != np.asarray(b["cell_forward"])).any() for a, b in zip(out[-3:], out0))
    out_keys = run_jobs_on_gpu(gpu_ctx, dchunk, jobs[-3:], use_indices=False)
    for f, r in zip(jobs[-3:], out_keys):
        assert_job_equal(f, r, exact=True)
    dchunk.close()
    oc.close()


def test_host_pipeline_with_ancestor_tables(gpu_ctx, orc):
    """mrp_phase_reads and mrp_phase_reads_many (resident) on a chunk whose site tables are non-zero: the final sweep
    (bubbleGraph.c:2748-2749) and fillInPredictedGenome (emissions.c:323-343) read them; results equal the oracle's."""
    chunk = _chunk_with_tables(73, 180, 28, allele_choices=(2, 3), allele_probs=(0.7, 0.3))
    pd = synth.shipped_phase_params()
    params = capi.Params.from_reference_names(pd)
    oc = orc.OracleChunk(chunk)
    ref = oc.phase(pd, capture_jobs=True)
    oc.close()
    assert ref["jobs"][-1]["flags"] == 3
    dchunk = capi.DeviceChunk.from_chunk(gpu_ctx, chunk)
    (final,) = run_jobs_on_gpu(gpu_ctx, dchunk, ref["jobs"][-1:])
    assert_job_equal(ref["jobs"][-1], final, exact=True)
    host = capi.phase_reads(gpu_ctx, dchunk, chunk, params)
    (res,), st = capi.phase_reads_many(gpu_ctx, [dchunk], [chunk], params)
    assert st.resident == 1
    for got in (host, res):
        for k in PHASE_KEYS:
            assert (np.asarray(got[k]) == np.asarray(ref[k])).all(), k
        assert got["reads1"] == ref["reads1"] and got["reads2"] == ref["reads2"]
        assert got["hmm_forward"] == ref["jobs"][-1]["hmm_forward"]
    # the tables are not decoration: with all-zero tables the genotype probabilities differ
    zero = synth.Chunk(chunk.allele_number, chunk.allele_offset, np.zeros_like(chunk.sub), np.zeros_like(chunk.prior), chunk.pool, chunk.reads)
    dz = capi.DeviceChunk.from_chunk(gpu_ctx, zero)
    (rz,), _ = capi.phase_reads_many(gpu_ctx, [dz], [zero], params)
    assert (np.asarray(rz["genotype_probs"]) != np.asarray(res["genotype_probs"])).any()
    dz.close()
    dchunk.close()


@pytest.mark.parametrize("het", [1e-3, 0.05])
def test_bubbles_with_positive_het_substitution_probability(gpu_ctx, orc, het):
    """bubbleGraph_getReference (bubbleGraph.c:2443-2474) with hetSubstitutionProbability > 0 -> non-zero uint16 tables
    from mrp_reference_from_bubbles -> resident phasing with the ancestor model in the final sweep == oracle chain."""
    from oracle import frame_oracle as fo
    rng = np.random.default_rng(int(het * 1e6) + 5)
    n_sites, n_reads = 110, 80
    an = rng.choice([2, 3, 4], size=n_sites, p=[0.6, 0.3, 0.1])
    truth1 = (rng.random(n_sites) * an).astype(int)
    truth2 = (truth1 + 1 + (rng.random(n_sites) * (an - 1)).astype(int)) % an
    spans, haps, strands = [], rng.integers(0, 2, size=n_reads), rng.integers(0, 2, size=n_reads)
    for r in range(n_reads):
        a = int(rng.integers(0, n_sites - 5))
        spans.append((a, int(min(n_sites - 1, a + rng.integers(4, 40)))))
    br, sup = [], []
    for i in range(n_sites):
        rs = [r for r, (a, b) in enumerate(spans) if a <= i <= b]
        s = np.zeros((int(an[i]), len(rs)), dtype=np.float32)
        for j, r in enumerate(rs):
            allele = int(truth1[i] if haps[r] == 0 else truth2[i])
            if rng.random() < 0.08:
                allele = int((allele + 1) % an[i])
            s[:, j] = -rng.uniform(1.0, 9.0, size=int(an[i])).astype(np.float32)
            s[allele, j] = np.float32(-rng.uniform(0.0, 0.3))
        br.append(rs); sup.append(s)
    seqs, pool = capi.profile_seqs_from_bubbles(an, br, sup, n_reads)
    a_num, sub, prior = capi.reference_from_bubbles(an, br, sup, het)
    assert sub.max() > 0 and prior.max() == 0  # the reference's builder leaves the priors at 0 (calloc, :2461)
    o_an, o_sub, o_prior = fo.get_reference([fo.Bubble(int(a), rs, np.asarray(s).reshape(-1).tolist()) for a, rs, s in zip(an, br, sup)], het)
    assert (np.asarray(o_sub, dtype=np.uint16) == sub).all() and (np.asarray(o_prior, dtype=np.uint16) == prior).all()
    off = np.concatenate([[0], np.cumsum(a_num)]).astype(np.int64)
    reads = [synth.Read(name=f"r{q['read']:04d}", ref_start=q["ref_start"], length=q["length"], strand=int(strands[q["read"]]),
                        hap=int(haps[q["read"]]), pool_off=q["pool_offset"], nbytes=int(off[q["ref_start"] + q["length"]] - off[q["ref_start"]]))
             for q in seqs]
    chunk = synth.Chunk(allele_number=a_num, allele_offset=off, sub=sub, prior=prior, pool=pool, reads=reads)
    pd = synth.shipped_phase_params()
    dchunk = capi.DeviceChunk.from_chunk(gpu_ctx, chunk)
    (got,), st = capi.phase_reads_many(gpu_ctx, [dchunk], [chunk], capi.Params.from_reference_names(pd))
    assert st.resident == 1
    oc = orc.OracleChunk(chunk)
    ref = oc.phase(pd)
    oc.close()
    for k in PHASE_KEYS:
        assert (np.asarray(got[k]) == np.asarray(ref[k])).all(), k
    assert got["reads1"] == ref["reads1"] and got["reads2"] == ref["reads2"]
    dchunk.close()
