"""GPU parity of the pair-HMM forward probability (mrp_forward_probabilities / mrp_allele_read_supports) through the C-ABI:
bit-exact fp64 against the CPU oracle (the arithmetic is + and * only; neither side may fuse them)."""
import os

import numpy as np
import pytest

from margin_amd import capi, synth
from oracle import pairhmm as ph
from tests.test_pairhmm import GOLDEN, omodel, random_anchors

pytestmark = pytest.mark.gpu


def models3():
    t, tr, em = synth.margin_phase_pair_hmm_arrays()
    f = capi.PairHmm.from_margin_hmm(t, tr, em)
    return [f, f.reverse_complement(), capi.PairHmm.default_nucleotide()]


def pack(pairs):
    strings, xo, xl, yo, yl, pos = [], [], [], [], [], 0
    for a, b in pairs:
        strings += [a, b]
        xo.append(pos); xl.append(len(a)); pos += len(a)
        yo.append(pos); yl.append(len(b)); pos += len(b)
    pool = np.concatenate(strings) if pos else np.zeros(0, np.uint8)
    return pool, np.array(xo, np.int64), np.array(xl, np.int32), np.array(yo, np.int64), np.array(yl, np.int32)


def test_golden_vectors(gpu_ctx):
    g = np.load(GOLDEN)
    models = [capi.PairHmm.from_buffer_copy(g["models"][i].tobytes()) for i in range(len(g["models"]))]
    for tag in ("short", "long"):
        out, st = capi.forward_probabilities(gpu_ctx, models, g["pool"], g[f"{tag}_x_off"], g[f"{tag}_x_len"], g[f"{tag}_y_off"], g[f"{tag}_y_len"],
                                             g[f"{tag}_model"], g[f"{tag}_anchor_off"], g[f"{tag}_anchors"], int(g["expansion"]),
                                             bool(g[f"{tag}_ragged"][0]), bool(g[f"{tag}_ragged"][1]))
        assert (out == g[f"{tag}_out"]).all()
        assert (st.pairs_wave > 0) == (tag == "long")


@pytest.mark.parametrize("ragged", [(False, False), (True, False), (False, True), (True, True)])
def test_pair_per_lane_kernel_bit_exact(gpu_ctx, ragged):
    """every launch class of the pair-per-lane kernel (x up to 26 / 34 / 52 / 100 symbols with three models), empty strings, Ns, three models"""
    rng = np.random.default_rng(100 + 2 * ragged[0] + ragged[1])
    pairs = []
    for _ in range(700):
        lx = int(rng.choice([0, 1, 2, 25, 26, 27, 34, 35, 52, 53, 100])) if rng.random() < 0.4 else int(rng.integers(0, 101))
        a = synth.random_sequence(rng, lx, n_rate=0.03)
        b = synth.evolve_sequence(rng, a) if rng.random() < 0.8 else synth.random_sequence(rng, int(rng.integers(0, 160)), n_rate=0.03)
        pairs.append((a, b))
    pairs += [(np.zeros(0, np.uint8), np.zeros(0, np.uint8)), (np.zeros(0, np.uint8), synth.random_sequence(rng, 9)), (synth.random_sequence(rng, 9), np.zeros(0, np.uint8))]
    pool, xo, xl, yo, yl = pack(pairs)
    mi = rng.integers(0, 3, size=len(pairs)).astype(np.uint8)
    ms = models3()
    out, st = capi.forward_probabilities(gpu_ctx, ms, pool, xo, xl, yo, yl, mi, expansion=4, ragged_left=ragged[0], ragged_right=ragged[1])
    ref = ph.forward_batch([omodel(m) for m in ms], pool, xo, xl, yo, yl, mi, expansion=4, ragged_left=ragged[0], ragged_right=ragged[1])
    assert st.pairs_lane == len(pairs) and st.pairs_wave == 0
    assert st.cells == int(((xl.astype(np.int64) + 1) * (yl.astype(np.int64) + 1)).sum())
    assert (out == ref).all()
    assert out[len(pairs) - 3] == 0.0  # two empty strings: LOG_ONE, pairwiseAligner.c:860-862


def test_pair_per_wave_kernel_bit_exact(gpu_ctx):
    """long strings without anchors (whole matrix, diagonals wider than a wave) and anchored bands of several expansions"""
    rng = np.random.default_rng(7)
    ms = models3()
    oms = [omodel(m) for m in ms]
    for expansion in (0, 4, 20):
        pairs, aoff, anc = [], [0], []
        for i in range(40):
            lx = int(rng.integers(101, 700)) if i % 4 else int(rng.integers(0, 60))
            a = synth.random_sequence(rng, lx, n_rate=0.01)
            b = synth.evolve_sequence(rng, a)
            pairs.append((a, b))
            if i % 3:
                anc += random_anchors(rng, len(a), len(b))
            aoff.append(len(anc))
        pool, xo, xl, yo, yl = pack(pairs)
        mi = rng.integers(0, 3, size=len(pairs)).astype(np.uint8)
        an = np.array(anc, np.int64).reshape(-1, 2)
        out, st = capi.forward_probabilities(gpu_ctx, ms, pool, xo, xl, yo, yl, mi, np.array(aoff, np.int64), an, expansion=expansion, ragged_left=True)
        ref = ph.forward_batch(oms, pool, xo, xl, yo, yl, mi, np.array(aoff, np.int64), an, expansion=expansion, ragged_left=True)
        assert st.pairs_wave > 0 and np.isfinite(ref).all()
        assert (out == ref).all()


def test_allele_read_supports_like_the_bubble_loop(gpu_ctx):
    """mrp_allele_read_supports = the loop of bubbleGraph.c:1421-1464, duplicates and strands included, float32 results"""
    t, tr, em = synth.margin_phase_pair_hmm_arrays()
    f = capi.PairHmm.from_margin_hmm(t, tr, em)
    r = f.reverse_complement()
    bubbles = synth.make_bubble_strings(seed=11, n_sites=60, coverage=20, duplicate_rate=0.3)
    bubbles.append(([synth.random_sequence(np.random.default_rng(1), 25)], [], []))  # a bubble without reads
    got, st = capi.allele_read_supports(gpu_ctx, f, r, bubbles)
    n_dup = 0
    for (alleles, reads, fwd), sup in zip(bubbles, got):
        ref = ph.allele_read_supports(omodel(f), omodel(r), alleles, reads, fwd) if reads else np.zeros((len(alleles), 0), np.float32)
        assert sup.shape == ref.shape and (sup == ref).all()
        n_dup += len(reads) - len({bytes(x) for x in reads})
    assert n_dup > 50 and st.pairs_lane == sum(2 * len({bytes(x) for x in b[1]}) for b in bubbles)


def test_config2_chunk_of_alignments(gpu_ctx):
    """one 1 Mb chunk of config 2 (2 000 sites x ~30 reads x 2 alleles = ~1.2e5 pairs): a sample against the oracle, and the
    size-independent property that a pair's result does not depend on the batch it travels in"""
    ms = models3()[:2]
    bubbles = synth.make_bubble_strings(seed=2, n_sites=2000, coverage=30)
    pool, xo, xl, yo, yl, mi = synth.pairs_from_bubbles(bubbles)
    out, st = capi.forward_probabilities(gpu_ctx, ms, pool, xo, xl, yo, yl, mi)
    assert len(out) > 100_000 and np.isfinite(out).all() and (out <= 0).all()
    rng = np.random.default_rng(0)
    pick = np.sort(rng.choice(len(out), size=3000, replace=False))
    ref = ph.forward_batch([omodel(m) for m in ms], pool, xo[pick], xl[pick], yo[pick], yl[pick], mi[pick])
    assert (out[pick] == ref).all()
    sub, _ = capi.forward_probabilities(gpu_ctx, ms, pool, xo[pick], xl[pick], yo[pick], yl[pick], mi[pick])
    assert (sub == out[pick]).all()


def test_errors(gpu_ctx):
    ms = models3()
    a = np.zeros(10, np.uint8)
    with pytest.raises(capi.MrpError) as e:  # anchors not increasing
        capi.forward_probabilities(gpu_ctx, ms, a, [0], [5], [5], [5], None, [0, 2], [[2, 2], [2, 3]])
    assert e.value.code == capi.MRP_ERR_ARG
    with pytest.raises(capi.MrpError) as e:  # string outside the pool
        capi.forward_probabilities(gpu_ctx, ms, a, [0], [5], [6], [5])
    assert e.value.code == capi.MRP_ERR_ARG
    with pytest.raises(capi.MrpError) as e:  # model index
        capi.forward_probabilities(gpu_ctx, ms, a, [0], [5], [5], [5], [3])
    assert e.value.code == capi.MRP_ERR_ARG
    big = np.zeros(2 * 2100, np.uint8)
    with pytest.raises(capi.MrpError) as e:  # a 2 101-cell diagonal
        capi.forward_probabilities(gpu_ctx, ms, big, [0], [2100], [2100], [2100])
    assert e.value.code == capi.MRP_ERR_UNSUPPORTED
    out, _ = capi.forward_probabilities(gpu_ctx, ms, a, np.zeros(0, np.int64), np.zeros(0, np.int32), np.zeros(0, np.int64), np.zeros(0, np.int32))
    assert len(out) == 0


def test_strings_to_haplotype_tags(gpu_ctx, orc):
    """From the aligned strings to the HP tags with nothing but the product in the chain: read substrings x alleles ->
    mrp_allele_read_supports (pair-HMM, device) -> profile bytes and site tables (rphmm_frame.c) -> device-resident phasing ->
    read-to-haplotype assignment; against the same chain built from the oracles."""
    from oracle import frame_oracle as fo
    rng = np.random.default_rng(77)
    n_sites, n_reads = 150, 110
    t, tr, em = synth.margin_phase_pair_hmm_arrays()
    fwd = capi.PairHmm.from_margin_hmm(t, tr, em)
    rev = fwd.reverse_complement()
    truth = rng.integers(0, 2, size=n_sites)
    haps, strands = rng.integers(0, 2, size=n_reads), rng.integers(0, 2, size=n_reads)
    spans = []
    for r in range(n_reads):
        a = int(rng.integers(0, n_sites - 5))
        spans.append((a, int(min(n_sites - 1, a + rng.integers(4, 40)))))
    bubbles, br = [], []
    for i in range(n_sites):
        ref = synth.random_sequence(rng, 25)
        alt = ref.copy()
        alt[12] = (alt[12] + 1 + rng.integers(0, 3)) % 4
        rs = [r for r, (a, b) in enumerate(spans) if a <= i <= b]
        reads = []
        for r in rs:
            allele = truth[i] if haps[r] == 0 else 1 - truth[i]
            reads.append(synth.evolve_sequence(rng, alt if allele else ref, 0.04, 0.02, 0.02))
        bubbles.append(([ref, alt], reads, [bool(strands[r]) for r in rs]))
        br.append(rs)
    sup, st = capi.allele_read_supports(gpu_ctx, fwd, rev, bubbles)
    for (alleles, reads, fs), s_ in zip(bubbles, sup):
        assert (s_ == ph.allele_read_supports(omodel(fwd), omodel(rev), alleles, reads, fs)).all()
    an = [2] * n_sites
    seqs, pool = capi.profile_seqs_from_bubbles(an, br, sup, n_reads)
    ref_seqs = fo.get_profile_seqs([fo.Bubble(2, rs, np.asarray(s_).reshape(-1).tolist()) for rs, s_ in zip(br, sup)])
    assert (np.array([b for p in ref_seqs.values() for b in p["probs"]], dtype=np.uint8) == pool).all()
    a_num, sub, prior = capi.reference_from_bubbles(an, br, sup, 0.0)
    off = np.concatenate([[0], np.cumsum(a_num)]).astype(np.int64)
    reads = [synth.Read(name=f"r{q['read']:04d}", ref_start=q["ref_start"], length=q["length"], strand=int(strands[q["read"]]), hap=int(haps[q["read"]]),
                        pool_off=q["pool_offset"], nbytes=int(off[q["ref_start"] + q["length"]] - off[q["ref_start"]])) for q in seqs]
    chunk = synth.Chunk(allele_number=a_num, allele_offset=off, sub=sub, prior=prior, pool=pool, reads=reads)
    pd = synth.shipped_phase_params()
    dchunk = capi.DeviceChunk.from_chunk(gpu_ctx, chunk)
    (got,), pst = capi.phase_reads_many(gpu_ctx, [dchunk], [chunk], capi.Params.from_reference_names(pd))
    oc = orc.OracleChunk(chunk)
    ref = oc.phase(pd)
    oc.close()
    for k in ("hap1", "hap2", "genotype", "ancestor"):
        assert (np.asarray(got[k]) == np.asarray(ref[k])).all(), k
    assert got["reads1"] == ref["reads1"] and got["reads2"] == ref["reads2"]
    recs, _ = capi.read_records(chunk)
    hap, _phred = capi.assign_reads_to_haplotypes(a_num, pool, recs, len(reads), got, min_phred=0)
    agree = sum(1 for i, r in enumerate(reads) if hap[i] in (1, 2) and (hap[i] - 1) == r.hap)
    tagged = int(((hap == 1) | (hap == 2)).sum())
    assert tagged >= 0.9 * len(reads) and max(agree, tagged - agree) >= 0.9 * tagged
    dchunk.close()


def test_many_models(gpu_ctx):
    """the emission tables of the pair-per-lane kernel live in LDS next to the rows: 40 models shrink the rows it can hold,
    200 models leave no room at all (every pair then takes the pair-per-wave kernel); results do not change"""
    rng = np.random.default_rng(5)
    base = models3()
    pairs = [(synth.random_sequence(rng, int(rng.integers(0, 60))), synth.random_sequence(rng, int(rng.integers(0, 60)))) for _ in range(300)]
    pool, xo, xl, yo, yl = pack(pairs)
    for n_models in (40, 200):
        ms = []
        for i in range(n_models):
            m = base[i % 3].copy()
            m.gap_open_x -= 0.01 * i
            m.e_match[5] -= 0.003 * i
            ms.append(m)
        mi = rng.integers(0, n_models, size=len(pairs)).astype(np.uint8)
        out, st = capi.forward_probabilities(gpu_ctx, ms, pool, xo, xl, yo, yl, mi)
        ref = ph.forward_batch([omodel(m) for m in ms], pool, xo, xl, yo, yl, mi)
        assert (out == ref).all()
        assert (st.pairs_lane == 0) == (n_models == 200)


def test_structural_variant_alleles_are_anchored(gpu_ctx):
    """bubbleGraph.c:1448-1451: strings longer than referenceExpansionForStructuralVariants are banded around their shared
    20-mers.  Bubbles with an insertion allele of several hundred bases next to ordinary SNP bubbles, through
    mrp_allele_read_supports and through the oracle's loop."""
    rng = np.random.default_rng(21)
    t, tr, em = synth.margin_phase_pair_hmm_arrays()
    f = capi.PairHmm.from_margin_hmm(t, tr, em)
    r = f.reverse_complement()
    bubbles = synth.make_bubble_strings(seed=4, n_sites=6, coverage=12)
    for _ in range(4):
        flank = synth.random_sequence(rng, 560)
        ins = synth.random_sequence(rng, int(rng.integers(80, 400)))
        ref = flank
        alt = np.concatenate([flank[:280], ins, flank[280:]])
        reads, strands = [], []
        for _k in range(10):
            reads.append(synth.evolve_sequence(rng, alt if rng.random() < 0.5 else ref, 0.03, 0.01, 0.01))
            strands.append(bool(rng.random() < 0.5))
        reads.append(reads[0].copy())
        strands.append(not strands[0])
        bubbles.append(([ref, alt], reads, strands))
    got, st = capi.allele_read_supports(gpu_ctx, f, r, bubbles, expansion=4, sv_threshold=512)
    assert st.pairs_wave > 0 and st.pairs_lane > 0
    for (alleles, reads, fs), s_ in zip(bubbles, got):
        ref_s = ph.allele_read_supports(omodel(f), omodel(r), alleles, reads, fs, expansion=4, sv_threshold=512)
        assert np.isfinite(ref_s).all() and (s_ == ref_s).all()
    # without the threshold the same pairs take their whole matrix: other numbers (the band cuts probability mass away)
    full, _ = capi.allele_read_supports(gpu_ctx, f, r, bubbles[-1:], expansion=4, sv_threshold=10 ** 9)
    assert (full[0] >= got[-1]).all() and (full[0] != got[-1]).any()


def test_degenerate_models_with_zero_probabilities(gpu_ctx):
    """log(0) = -inf in transitions and emissions (an hmm file may hold exact zeros): no gaps allowed -> pairs of unequal
    length are impossible (LOG_ZERO), exact-match-only emissions, one-sided gaps.  -inf must come out as -inf, never NaN."""
    rng = np.random.default_rng(17)
    t, tr, em = synth.margin_phase_pair_hmm_arrays()
    no_gaps = capi.PairHmm.from_margin_hmm(3, [1.0, 0.0, 0.0, 1.0, 0.0, 0.0, 1.0, 0.0, 0.0], em)
    exact = capi.PairHmm.from_margin_hmm(2, tr, [1.0 if i % 5 == 0 else 0.0 for i in range(16)] + em[16:])
    one_sided = capi.PairHmm.from_margin_hmm(3, [0.8, 0.2, 0.0, 0.5, 0.5, 0.0, 0.0, 0.0, 0.0], em)
    ms = [no_gaps, exact, one_sided]
    pairs = []
    for _ in range(240):
        a = synth.random_sequence(rng, int(rng.integers(0, 40)), n_rate=0.05)
        b = a.copy() if rng.random() < 0.4 else synth.evolve_sequence(rng, a, 0.05, 0.05, 0.05)
        pairs.append((a, b))
    for _ in range(12):
        a = synth.random_sequence(rng, int(rng.integers(120, 300)))
        pairs.append((a, a.copy() if rng.random() < 0.5 else synth.evolve_sequence(rng, a, 0.02, 0.0, 0.02)))
    pool, xo, xl, yo, yl = pack(pairs)
    mi = rng.integers(0, 3, size=len(pairs)).astype(np.uint8)
    out, st = capi.forward_probabilities(gpu_ctx, ms, pool, xo, xl, yo, yl, mi)
    ref = ph.forward_batch([omodel(m) for m in ms], pool, xo, xl, yo, yl, mi)
    assert st.pairs_lane > 0 and st.pairs_wave > 0
    assert not np.isnan(ref).any() and not np.isnan(out).any()
    assert np.isneginf(ref).sum() > 20 and np.isfinite(ref).sum() > 20
    assert ((out == ref) | (np.isneginf(out) & np.isneginf(ref))).all()


def test_maximum_sizes(gpu_ctx):
    """the limits of both kernels: x of exactly 100 symbols against reads of up to 1 500 (pair per lane, long rows), and
    unanchored pairs whose widest diagonal has 1 024 < cells <= 2 048 (the largest launch class of the pair-per-wave kernel)"""
    rng = np.random.default_rng(29)
    ms = models3()
    pairs = [(synth.random_sequence(rng, 100), synth.random_sequence(rng, int(rng.integers(800, 1500)))) for _ in range(70)]
    pairs += [(synth.random_sequence(rng, int(rng.integers(1, 30))), synth.random_sequence(rng, 1400)) for _ in range(10)]
    big = []
    for n in (1030, 1500, 2047):
        a = synth.random_sequence(rng, n)
        big.append((a, synth.evolve_sequence(rng, a, 0.02, 0.01, 0.01)))
    pairs += big
    pool, xo, xl, yo, yl = pack(pairs)
    mi = rng.integers(0, 3, size=len(pairs)).astype(np.uint8)
    out, st = capi.forward_probabilities(gpu_ctx, ms, pool, xo, xl, yo, yl, mi)
    ref = ph.forward_batch([omodel(m) for m in ms], pool, xo, xl, yo, yl, mi)
    assert st.pairs_lane == 80 and st.pairs_wave == 3
    assert np.isfinite(ref).all() and (out == ref).all()
