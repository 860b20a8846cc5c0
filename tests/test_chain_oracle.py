"""The oracle side of the f-4 chain test (tests/test_gpu_chain.py) on the CPU: adjacent chunks phased by the rphmm oracle, read
assignment, stitching and phase sets by frame_oracle.py.  No reference vector exists for this chain; what is
checked here are the properties the reference's integration test asserts of a stitched run (tests/marginTest.c:175-178,
238-241: both haplotypes populated; every phased genotype is the input genotype cis or trans) and that stitching makes the labels
of all chunks consistent."""
import pytest

from margin_amd import synth
from tests import chain_helpers as ch


@pytest.mark.parametrize("seed", [1, 6, 8])
def test_oracle_chain_labels_are_consistent_across_chunks(orc, seed):
    from oracle import frame_oracle as fo
    chunks, windows, ids, truth = ch.make_adjacent_chunks(seed, n_chunks=4)
    r = ch.oracle_chain(orc, fo, chunks, windows, ids, synth.shipped_phase_params(), 0, 40, (1, 0.0, 0.5))
    assert any(r["switched"][1:])
    votes = []
    for c, (chunk, (hap, phred)) in enumerate(zip(chunks, r["assigned"])):
        n1 = sum(1 for h in hap if h == 1)
        n2 = sum(1 for h in hap if h == 2)
        assert 3 * n1 > n2 and 3 * n2 > n1  # (marginTest.c:175-178 asks two thirds on real data; the simulated reads pick a haplotype at random)
        for i, h in enumerate(hap):
            if h in (1, 2) and phred[i] > 0:
                votes.append(((int(h) - 1) ^ int(r["switched"][c]), truth[ids[c][i]]))
    agree = sum(1 for h, t in votes if h == t)
    assert max(agree, len(votes) - agree) >= 0.9 * len(votes)
    # one record per site of the stretch; the sites are all het in truth and nearly all are called het (marginTest.c:238-241 asks
    # that a phased genotype be the input genotype cis or trans)
    assert len(r["variants"]) == windows[-1][1] - windows[0][0]
    het = sum(1 for v in r["variants"] if {v["gt1"], v["gt2"]} == {0, 1})
    assert het >= 0.9 * len(r["variants"])
