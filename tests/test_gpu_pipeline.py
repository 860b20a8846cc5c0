"""GPU parity of the product host pipeline (flat stRPHmm structural ops + device sweeps) against the
oracle's restatement of coordination.c / hmm.c / genomeFragment.c / bubbleGraph.c:2673."""
import numpy as np
import pytest

from margin_amd import capi, synth

pytestmark = pytest.mark.gpu

STRUCT_KEYS = ["col_ref_start", "col_length", "col_depth", "col_cell_off", "col_read_off", "read_byte_off", "read_ids",
               "partition", "mask_from", "mask_to", "mcol_cell_off", "merge_from", "merge_to", "cell_next", "cell_prev"]
VALUE_KEYS = ["cell_forward", "cell_backward", "merge_forward", "merge_backward", "col_total"]


def assert_same_hmm(a, b, values=True):
    assert a["n_columns"] == b["n_columns"] and a["ref_start"] == b["ref_start"] and a["ref_length"] == b["ref_length"]
    K = a["n_columns"]
    for k in STRUCT_KEYS:
        x, y = np.asarray(a[k]), np.asarray(b[k])
        if k in ("cell_next", "cell_prev"):
            # entries of the last column's next / first column's prev are meaningless
            off = a["col_cell_off"]
            lo, hi = (0, int(off[K - 1])) if k == "cell_next" else (int(off[1]) if K > 1 else len(x), len(x))
            x, y = x[lo:hi], y[lo:hi]
        assert x.shape == y.shape and (x.astype(np.int64) == y.astype(np.int64)).all() if x.dtype != np.uint64 else (x == y).all(), k
    if values:
        for k in VALUE_KEYS:
            x, y = np.asarray(a[k]), np.asarray(b[k])
            assert ((x == y) | (np.isneginf(x) & np.isneginf(y))).all(), k


@pytest.mark.parametrize("seed,n_sites,cov", [(3, 200, 30), (4, 120, 45), (9, 60, 12)])
def test_get_rp_hmms_matches_oracle_every_array(gpu_ctx, orc, seed, n_sites, cov):
    """getRPHmms per strand: the pruned hmms (structure, order, and the f/b values they carry) are
    identical array for array -- this exercises fuse/align/cross-product/sweep/prune at every level."""
    chunk = synth.make_ont_chunk(seed=seed, region_bp=n_sites * 500, n_sites=n_sites, coverage=cov)
    pd = synth.shipped_phase_params()
    pd["includeAncestorSubProb"] = 0
    oc = orc.OracleChunk(chunk)
    dchunk = capi.DeviceChunk.from_chunk(gpu_ctx, chunk)
    params = capi.Params.from_reference_names(pd)
    for strand in (1, 0):
        idx = [i for i, r in enumerate(chunk.reads) if r.strand == strand]
        ref = oc.get_rp_hmms(orc.make_params(pd), idx)
        got = capi.get_rp_hmms(gpu_ctx, dchunk, chunk, params, idx)
        assert len(ref) == len(got)
        for hr, hg in zip(ref, got):
            fr = orc.flatten(hr, oc.pool_off)
            fg = capi.hmm_to_flat(hg)
            assert_same_hmm(fr, fg)
            capi.hmm_destroy(hg)
    dchunk.close()
    oc.close()


@pytest.mark.parametrize("seed,n_sites,cov,maxdepth", [(3, 200, 30, 64), (5, 150, 40, 12)])
def test_phase_reads_matches_oracle(gpu_ctx, orc, seed, n_sites, cov, maxdepth):
    """The whole phasing driver: haplotype strings, genotype calls and the read bipartition (the HP
    tags) are identical, including the coverage-filter path (maxCoverageDepth lowered)."""
    chunk = synth.make_ont_chunk(seed=seed, region_bp=n_sites * 500, n_sites=n_sites, coverage=cov)
    pd = synth.shipped_phase_params()
    pd["maxCoverageDepth"] = maxdepth
    oc = orc.OracleChunk(chunk)
    ref = oc.phase(pd)
    dchunk = capi.DeviceChunk.from_chunk(gpu_ctx, chunk)
    got = capi.phase_reads(gpu_ctx, dchunk, chunk, capi.Params.from_reference_names(pd))
    assert got["ref_start"] == ref["ref_start"] and got["length"] == ref["length"]
    for k in ("hap1", "hap2", "genotype", "ancestor", "support1", "support2", "genotype_probs", "hap_probs1", "hap_probs2"):
        assert (np.asarray(got[k]) == np.asarray(ref[k])).all(), k
    assert sorted(got["reads1"]) == sorted(ref["reads1"]) and sorted(got["reads2"]) == sorted(ref["reads2"])
    assert got["reads1"] == ref["reads1"] and got["reads2"] == ref["reads2"]
    assert got["n_sweeps"] == ref["fb_calls"]
    dchunk.close()
    oc.close()


def test_unit_test_shape_sum_mode_pipeline(gpu_ctx, orc):
    """tests/stRPHmmTest.c-shaped input (1..9 alleles, sum mode): structure identical, values and
    therefore prune decisions agree (posterior ties are broken identically because sorting is stable
    on both sides and values agree to ~1e-12)."""
    chunk = synth.make_unit_test_chunk(seed=21, ref_length=150, coverage=12, min_read=10, max_read=60, error_rate=0.05)
    pd = synth.unit_test_params(max_partitions=50, max_not_sum=1)
    oc = orc.OracleChunk(chunk)
    dchunk = capi.DeviceChunk.from_chunk(gpu_ctx, chunk)
    ref = oc.get_rp_hmms(orc.make_params(pd))
    got = capi.get_rp_hmms(gpu_ctx, dchunk, chunk, capi.Params.from_reference_names(pd))
    assert len(ref) == len(got)
    for hr, hg in zip(ref, got):
        assert_same_hmm(orc.flatten(hr, oc.pool_off), capi.hmm_to_flat(hg))
        capi.hmm_destroy(hg)
    dchunk.close()
    oc.close()


def test_split_where_phasing_is_uncertain_matches_oracle(gpu_ctx, orc):
    """stRPHMM_splitWherePhasingIsUncertain (hmm.c:1322-1383; the reference's system test runs it with
    splitHmmsWherePhasingUncertain, tests/stRPHmmTest.c:254-263) and stRPHmm_split (hmm.c:1231-1300): the pieces of every hmm
    of getRPHmms -- intervals, read lists, the cut column on both sides, cells, transitions, merge cells -- equal the oracle's."""
    import ctypes as C
    chunk = synth.make_unit_test_chunk(77, 200, 10, 10, 40, 0.05)
    pd = synth.unit_test_params(max_partitions=50, max_not_sum=1, min_cov=15)
    oc = orc.OracleChunk(chunk)
    oparams = orc.make_params(pd)
    dchunk = capi.DeviceChunk.from_chunk(gpu_ctx, chunk)
    params = capi.Params.from_reference_names(pd)
    ref = oc.get_rp_hmms(oparams)
    got = capi.get_rp_hmms(gpu_ctx, dchunk, chunk, params)
    assert len(ref) == len(got)
    L = orc.lib()
    n_pieces = 0
    for hr, hg in zip(ref, got):
        n = C.c_int64(0)
        parts = L.orc_hmm_splitWherePhasingIsUncertain(hr, C.byref(n))
        orc.check_error()
        mine = capi.hmm_split_where_phasing_is_uncertain(gpu_ctx, dchunk, chunk, hg, params)
        assert len(mine) == n.value
        for i in range(n.value):
            fr, fg = orc.flatten(parts[i], oc.pool_off), capi.hmm_to_flat(mine[i])
            assert_same_hmm(fr, fg, values=False)
            n_pieces += 1
        # one more cut of the first piece, in the middle of one of its columns when it has one longer than a site
        f0 = capi.hmm_to_flat(mine[0])
        starts, lens = np.asarray(f0["col_ref_start"]), np.asarray(f0["col_length"])
        wide = np.nonzero(lens > 1)[0]
        if f0["ref_length"] > 1:
            sp = int(starts[wide[0]] + 1) if wide.size else int(f0["ref_start"] + 1)
            right_o = L.orc_hmm_split(parts[0], sp)
            orc.check_error()
            right_g = capi.hmm_split(dchunk, chunk, mine[0], sp)
            assert_same_hmm(orc.flatten(parts[0], oc.pool_off), capi.hmm_to_flat(mine[0]), values=False)
            assert_same_hmm(orc.flatten(right_o, oc.pool_off), capi.hmm_to_flat(right_g), values=False)
            L.orc_hmm_destruct(right_o, 1)
            capi.hmm_destroy(right_g)
        for i in range(n.value):
            L.orc_hmm_destruct(parts[i], 1)
            capi.hmm_destroy(mine[i])
        L.free(parts)
    assert n_pieces > len(ref)  # the parameters of the system test do cut
    with pytest.raises(capi.MrpError):
        h = capi.get_rp_hmms(gpu_ctx, dchunk, chunk, params)
        try:
            capi.hmm_split(dchunk, chunk, h[0], int(capi.hmm_to_flat(h[0])["ref_start"]))
        finally:
            for x in h:
                capi.hmm_destroy(x)
    dchunk.close()
    oc.close()
