"""Pair-HMM forward probability, CPU side: the oracle (oracle/pairhmm_oracle.c) against the pins the
reference's own tests hold (tests/pairwiseAlignerTest.c), the product's host functions against the oracle, and the
committed golden vectors."""
import ctypes as C
import math
import os

import numpy as np
import pytest

from margin_amd import capi, synth
from oracle import pairhmm as ph

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "pairhmm_forward.npz")


def omodel(m: capi.PairHmm) -> ph.Model:
    return ph.Model.from_buffer_copy(bytes(m))


def random_anchors(rng, lx, ly):
    """getRandomAnchorPairs, tests/pairwiseAlignerTest.c:342-358 (without the per-anchor expansion of the dynamic band)"""
    out, x, y = [], -1, -1
    while True:
        x += int(rng.integers(1, 21))
        y += int(rng.integers(1, 21))
        if x >= lx or y >= ly:
            return out
        out.append((x, y))


def test_bands_known_answer():
    """test_bands, tests/pairwiseAlignerTest.c:64-127: the exact diagonals for three anchors on a 6 x 5 matrix."""
    expect = [(0, 0), (-1, 1), (-2, 2), (-1, 3), (-2, 4), (-1, 3), (-2, 4), (-3, 3), (-2, 2), (-1, 3), (0, 2), (1, 1)]
    anchors = [(1, 0), (2, 1), (3, 3)]
    lo, hi = ph.band(anchors, 6, 5, 2)
    assert list(zip(lo.tolist(), hi.tolist())) == expect
    lo, hi = capi.band_diagonals(anchors, 6, 5, 2)
    assert list(zip(lo.tolist(), hi.tolist())) == expect


def test_band_closed_form_equals_sequential_construction():
    """mrp_band_diagonals (closed form per anchor segment) against the oracle's restatement of the reference loop."""
    rng = np.random.default_rng(5)
    for _ in range(300):
        lx, ly = int(rng.integers(0, 120)), int(rng.integers(0, 120))
        anchors = random_anchors(rng, lx, ly) if rng.random() < 0.8 else []
        e = 2 * int(rng.integers(0, 8))
        a, b = ph.band(anchors, lx, ly, e)
        c, d = capi.band_diagonals(anchors, lx, ly, e)
        assert (a == c).all() and (b == d).all()
    # a pair without anchors covers the whole matrix
    lo, hi = capi.band_diagonals([], 7, 4, 4)
    for d in range(12):
        xs = [x for x in range(8) if 0 <= d - x <= 4]
        assert lo[d] == 2 * xs[0] - d and hi[d] == 2 * xs[-1] - d


def test_band_rejects_what_the_reference_asserts():
    for anchors in ([(3, 3), (3, 4)], [(2, 2), (1, 5)], [(6, 0)], [(0, 5)]):
        with pytest.raises(capi.MrpError) as e:
            capi.band_diagonals(anchors, 6, 5, 2)
        assert e.value.code == capi.MRP_ERR_ARG
        with pytest.raises(ValueError):
            ph.band(anchors, 6, 5, 2)
    with pytest.raises(capi.MrpError):
        capi.band_diagonals([], 6, 5, 3)  # odd expansion, pairwiseAligner.c:179


def test_log_add_tolerance():
    """test_logAdd, tests/pairwiseAlignerTest.c:129-139"""
    rng = np.random.default_rng(1)
    for _ in range(20000):
        i, j = rng.random(), rng.random()
        assert abs(math.exp(ph.log_add(math.log(i), math.log(j))) - (i + j)) < 0.001
    ninf = -math.inf
    assert ph.log_add(ninf, ninf) == ninf and ph.log_add(ninf, -3.0) == -3.0 and ph.log_add(-3.0, ninf) == -3.0
    assert ph.log_add(0.0, -7.5) == 0.0 and ph.log_add(-7.5, 0.0) == 0.0


def test_cell_forward_equals_backward():
    """test_cell, tests/pairwiseAlignerTest.c:168-197"""
    tf, tb = ph.test_cell(omodel(capi.PairHmm.default_nucleotide()), 0, 3)
    assert abs(tf - tb) < 1e-5


def test_diagonal_dp_calculations():
    """test_diagonalDPCalculations, tests/pairwiseAlignerTest.c:257-340: AGCG against AGTTCG"""
    sx, sy = capi.symbols_from_chars("AGCG"), capi.symbols_from_chars("AGTTCG")
    tf, tb, diag, post = ph.full_matrices(omodel(capi.PairHmm.default_nucleotide()), sx, sy, 2)
    assert abs(tf - tb) < 0.001
    assert np.abs(diag - tf).max() < 0.01
    pairs = {(int(x), int(y)) for x, y in zip(*np.nonzero(post >= 0.2))}
    assert pairs == {(0, 0), (1, 1), (2, 4), (3, 5)}
    # computeForwardProbability on the same pair ends in the same cell with the same end probabilities
    assert ph.forward_probability(omodel(capi.PairHmm.default_nucleotide()), sx, sy, expansion=2) == tf


def test_forward_probability_is_a_log_probability():
    """test_computeForwardProbability, tests/pairwiseAlignerTest.c:1153-1189"""
    rng = np.random.default_rng(9)
    m = omodel(capi.PairHmm.default_nucleotide())
    for _ in range(200):
        sx = synth.random_sequence(rng, int(rng.integers(10, 100)))
        sy = synth.evolve_sequence(rng, sx)
        p = ph.forward_probability(m, sx, sy, expansion=20, ragged_left=rng.random() > 0.5, ragged_right=rng.random() > 0.5)
        assert -math.inf < p <= 0.0
    assert ph.forward_probability(m, np.zeros(0, np.uint8), np.zeros(0, np.uint8)) == 0.0


def test_margin_hmm_parameters():
    t, tr, em = synth.margin_phase_pair_hmm_arrays()
    m = capi.PairHmm.from_margin_hmm(t, tr, em)
    assert m.match_continue == math.log(0.8) and m.gap_open_x == math.log(0.1) and m.gap_switch_to_x == -math.inf
    assert m.e_gap_x[0] == 0.0 and m.e_gap_y[2] == math.log(0.25) and m.e_match[5] == math.log(0.973)
    r = m.reverse_complement()
    e = np.array(m.e_match).reshape(4, 4)
    assert (np.array(r.e_match).reshape(4, 4) == e[::-1, ::-1]).all()  # A<->T, C<->G on both strings
    asym = capi.PairHmm.from_margin_hmm(3, [0.8, 0.15, 0.05, 0.5, 0.4, 0.1, 0.6, 0.1, 0.3], em)
    assert asym.gap_open_x == math.log(0.15) and asym.gap_open_y == math.log(0.05) and asym.gap_switch_to_x == math.log(0.1)


def test_supports_cache_is_keyed_by_the_substring_alone():
    """bubbleGraph.c:1431-1441: a reverse strand read whose substring equals an earlier forward strand read's copies its scores."""
    rng = np.random.default_rng(3)
    t, tr, em = synth.margin_phase_pair_hmm_arrays()
    f = capi.PairHmm.from_margin_hmm(t, tr, em)
    fm, rm = omodel(f), omodel(f.reverse_complement())
    ref = synth.random_sequence(rng, 25)
    alt = ref.copy()
    alt[12] = (alt[12] + 1) % 4
    r0 = synth.evolve_sequence(rng, ref)
    reads = [r0, synth.evolve_sequence(rng, alt), r0.copy(), r0.copy()]
    sup = ph.allele_read_supports(fm, rm, [ref, alt], reads, [True, False, False, True])
    assert (sup[:, 2] == sup[:, 0]).all() and (sup[:, 3] == sup[:, 0]).all()
    alone = ph.allele_read_supports(fm, rm, [ref, alt], [r0], [False])
    assert (alone[:, 0] != sup[:, 2]).any()  # on its own the reverse strand machine gives other numbers
    assert sup[0, 1] == np.float32(ph.forward_probability(rm, ref, reads[1]))


def test_oracle_reproduces_golden_vectors():
    g = np.load(GOLDEN)
    models = [ph.Model.from_buffer_copy(g["models"][i].tobytes()) for i in range(len(g["models"]))]
    for tag in ("short", "long"):
        out = ph.forward_batch(models, g["pool"], g[f"{tag}_x_off"], g[f"{tag}_x_len"], g[f"{tag}_y_off"], g[f"{tag}_y_len"], g[f"{tag}_model"],
                               g[f"{tag}_anchor_off"], g[f"{tag}_anchors"], int(g["expansion"]), bool(g[f"{tag}_ragged"][0]), bool(g[f"{tag}_ragged"][1]))
        assert (out == g[f"{tag}_out"]).all()


def test_pairhmm_entry_points_fail_loudly_without_a_context():
    """no CPU fallback: without a device context the product returns MRP_ERR_NO_DEVICE, it does not compute"""
    L = capi.load()
    m = capi.PairHmm.default_nucleotide()
    out = np.zeros(1)
    z64, z32 = np.zeros(1, np.int64), np.array([3], np.int32)
    pool = np.zeros(6, np.uint8)
    y_off = np.array([3], np.int64)
    rc = L.mrp_forward_probabilities(None, C.byref(m), 1, 1, pool.ctypes.data, 6, z64.ctypes.data, z32.ctypes.data, y_off.ctypes.data, z32.ctypes.data,
                                     None, None, None, 4, 0, 0, out.ctypes.data, None)
    assert rc == capi.MRP_ERR_NO_DEVICE and b"no CPU fallback" in L.mrp_last_error()
    first = np.array([0, 1], np.int64)
    sup = np.zeros(1, np.float32)
    rc = L.mrp_allele_read_supports(None, C.byref(m), C.byref(m), 1, first.ctypes.data, first.ctypes.data, pool.ctypes.data, 6, z64.ctypes.data,
                                    z32.ctypes.data, y_off.ctypes.data, z32.ctypes.data, np.ones(1, np.uint8).ctypes.data, 4, 512, sup.ctypes.data, None)
    assert rc == capi.MRP_ERR_NO_DEVICE


def test_kmer_alignment_anchors():
    """test_getKmerAlignmentAnchors, tests/pairwiseAlignerTest.c:1191-1230 (anchors strictly increasing, inside both strings),
    and the product's function against the oracle's restatement (repeats included: only the first occurrence of a k-mer of
    x counts, pairwiseAligner.c:1547-1551)."""
    rng = np.random.default_rng(8)
    for t in range(200):
        a = synth.random_sequence(rng, int(rng.integers(1, 1000)), n_rate=0.005)
        b = synth.evolve_sequence(rng, a, 0.02, 0.01, 0.01)
        if t % 5 == 0 and len(a) > 120:
            b = np.concatenate([b[:len(b) // 2], a[40:100], b[len(b) // 2:]])
        ref = ph.kmer_anchors(a, b)
        got = capi.kmer_alignment_anchors(a, b)
        assert ref.shape == got.shape and (ref == got).all()
        px = py = -1
        for x, y in got:
            assert px < x < len(a) and py < y < len(b)
            px, py = x, y
        if len(got):
            capi.band_diagonals(got, len(a), len(b), 4)  # valid anchors for band_construct
    assert len(capi.kmer_alignment_anchors(synth.random_sequence(rng, 19), synth.random_sequence(rng, 300))) == 0
    same = synth.random_sequence(rng, 64)
    assert capi.kmer_alignment_anchors(same, same).tolist() == [[i + 10, i + 10] for i in range(45)]
