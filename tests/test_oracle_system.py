"""The reference's randomised system test (tests/stRPHmmTest.c:162-759) with fixed seeds, run on the
oracle: structural checks and the forward/backward invariants it asserts.  CPU only."""
import numpy as np
import pytest

from margin_amd import synth
from tests.helpers import posteriors

# (seed, ref_length, coverage, min_read, max_read, error, max_not_sum) -- the four parameterisations of
# stRPHmmTest.c:761-851, reference lengths scaled down so the CPU suite stays in seconds
CASES = [
    (101, 120, 12, 120, 120, 0.05, 0),   # :761 full-length reads, sum mode
    (102, 300, 12, 40, 40, 0.05, 1),     # :784 fixed-length reads, max mode
    (103, 200, 12, 10, 40, 0.05, 0),     # :807 short reads, sum mode
    (104, 150, 10, 10, 100, 0.01, 0),    # :830 mixed reads, sum mode
]


@pytest.mark.parametrize("seed,ref_len,cov,rmin,rmax,err,max_mode", CASES)
def test_system_invariants(orc, seed, ref_len, cov, rmin, rmax, err, max_mode):
    chunk = synth.make_unit_test_chunk(seed, ref_len, cov, rmin, rmax, err)
    oc = orc.OracleChunk(chunk)
    pd = synth.unit_test_params(max_partitions=50, max_not_sum=max_mode)
    params = orc.make_params(pd)
    hmms = oc.get_rp_hmms(params)
    L = orc.lib()
    spans = []
    covered = set()
    for h in hmms:
        L.orc_hmm_forwardBackward(h)
        orc.check_error()
        f = orc.flatten(h, oc.pool_off)
        K = f["n_columns"]
        spans.append((f["ref_start"], f["ref_start"] + f["ref_length"]))
        # column coordinates tile the hmm interval (:344-352)
        assert f["col_ref_start"][0] == f["ref_start"]
        assert (f["col_ref_start"][1:] == f["col_ref_start"][:-1] + f["col_length"][:-1]).all()
        assert f["col_ref_start"][-1] + f["col_length"][-1] == f["ref_start"] + f["ref_length"]
        assert (f["col_length"] > 0).all()
        for k in range(K):
            d = int(f["col_depth"][k])
            cells = f["partition"][f["col_cell_off"][k]:f["col_cell_off"][k + 1]]
            assert len(cells) >= 1
            if d < 64:
                assert (cells >> np.uint64(d) == 0).all()      # :372
            rids = f["read_ids"][f["col_read_off"][k]:f["col_read_off"][k + 1]]
            covered.update(int(r) for r in rids)
            for j, r in enumerate(rids):
                rd = chunk.reads[int(r)]
                # column inside the read's span (:360-362) and seqs pointer == getProb(seq, refStart, 0) (:365)
                assert rd.ref_start <= f["col_ref_start"][k] and rd.ref_start + rd.length >= f["col_ref_start"][k] + f["col_length"][k]
                want = rd.pool_off + int(chunk.allele_offset[f["col_ref_start"][k]] - chunk.allele_offset[rd.ref_start])
                assert f["read_byte_off"][f["col_read_off"][k] + j] == want
                if k + 1 < K:  # mask bits vs read ends (:388-397)
                    ends_here = rd.ref_start + rd.length == f["col_ref_start"][k] + f["col_length"][k]
                    assert ((int(f["mask_from"][k]) >> j) & 1) == (0 if ends_here else 1)
                if k > 0:      # (:404-414)
                    starts_here = rd.ref_start == f["col_ref_start"][k]
                    assert ((int(f["mask_to"][k - 1]) >> j) & 1) == (0 if starts_here else 1)
            if k + 1 < K:
                m0, m1 = int(f["mcol_cell_off"][k]), int(f["mcol_cell_off"][k + 1])
                assert (f["merge_from"][m0:m1] & f["mask_from"][k] == f["merge_from"][m0:m1]).all()  # :431-433
                assert (f["merge_to"][m0:m1] & f["mask_to"][k] == f["merge_to"][m0:m1]).all()
                assert bin(int(f["mask_from"][k])).count("1") == bin(int(f["mask_to"][k])).count("1")
        # forward == backward, every column total == forward (:460-466)
        assert abs(f["hmm_forward"] - f["hmm_backward"]) < 0.1
        assert np.abs(f["col_total"] - f["hmm_forward"]).max() < 0.1
        post = posteriors(f, f["cell_forward"], f["cell_backward"], f["col_total"])
        assert (post >= 0).all() and (post <= 1.0).all()
        if not max_mode:  # posteriors sum to one per column and per merge column (:470-503)
            for k in range(K):
                assert abs(post[f["col_cell_off"][k]:f["col_cell_off"][k + 1]].sum() - 1.0) < 0.1
                if k + 1 < K:
                    m0, m1 = int(f["mcol_cell_off"][k]), int(f["mcol_cell_off"][k + 1])
                    mp = np.minimum(1.0, np.exp(f["merge_forward"][m0:m1] + f["merge_backward"][m0:m1] - f["col_total"][k + 1]))
                    assert abs(mp.sum() - 1.0) < 0.1
        # trace back (:515-550): one cell per column, consecutive cells share a merge cell
        import ctypes as C
        n = C.c_int64(0)
        path = L.orc_hmm_forwardTraceBack(h, C.byref(n))
        assert n.value == K
        L.free(path)
    # hmms do not overlap (:268-274) and every read is in exactly one hmm (:294-329)
    spans.sort()
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))
    assert covered == set(range(len(chunk.reads)))
    for h in hmms:
        L.orc_hmm_destruct(h, 1)
    oc.close()


def test_split_where_phasing_is_uncertain(orc):
    """hmm.c:1322-1383 as used by the system test with splitHmmsWherePhasingUncertain (:254-263)."""
    import ctypes as C
    chunk = synth.make_unit_test_chunk(77, 200, 10, 10, 40, 0.05)
    oc = orc.OracleChunk(chunk)
    params = orc.make_params(synth.unit_test_params(max_partitions=50, max_not_sum=1, min_cov=15))
    hmms = oc.get_rp_hmms(params)
    L = orc.lib()
    total_len, pieces = 0, 0
    for h in hmms:
        start, length = h.contents.refStart, h.contents.refLength
        n = C.c_int64(0)
        parts = L.orc_hmm_splitWherePhasingIsUncertain(h, C.byref(n))
        orc.check_error()
        pos = start
        for i in range(n.value):
            p = parts[i].contents
            assert p.refStart == pos and p.refLength > 0
            pos += p.refLength
            pieces += 1
            L.orc_hmm_destruct(parts[i], 1)
        assert pos == start + length
        total_len += length
        L.free(parts)
    assert pieces >= len(hmms)
    oc.close()


def test_phase_driver_recovers_haplotypes(orc):
    """bubbleGraph.c:2673 driver on an ONT-like chunk: the two haplotypes come back (up to swap)
    and nearly every read lands on its true haplotype."""
    chunk = synth.make_ont_chunk(seed=5, region_bp=60_000, n_sites=120, coverage=25)
    oc = orc.OracleChunk(chunk)
    res = oc.phase(synth.shipped_phase_params())
    s, n = res["ref_start"], res["length"]
    t1, t2 = chunk.hap1[s:s + n], chunk.hap2[s:s + n]
    agree = max((res["hap1"] == t1).mean(), (res["hap1"] == t2).mean())
    assert agree > 0.97
    truth = np.array([r.hap for r in chunk.reads])
    a = (truth[res["reads1"]] == 0).sum() + (truth[res["reads2"]] == 1).sum()
    assert max(a, len(truth) - a) / len(truth) > 0.95
    assert sorted(res["reads1"] + res["reads2"]) == list(range(len(chunk.reads)))
    oc.close()
