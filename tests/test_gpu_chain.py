"""SURVEY.md 8 f-4 end to end on device results: ADJACENT chunks that share reads go through
    mrp_phase_reads_many (GPU, resident merge levels) -> mrp_assign_reads_to_haplotypes -> mrp_stitch_chunk -> mrp_phase_sets
and through the same chain made of the oracles (rphmm_oracle phasing, frame_oracle assignment / Stitcher / phase_sets); haplotype
strings, read sets, phred scores, the cis / trans counts and the flip decision of every chunk
(chunkToStitch_phaseAdjacentChunks, stitching.c:345-403), the read sets carried on, and the phase set id and break reason of every
record (writePhasedVcf, vcf.c:869-953) must be identical."""
import numpy as np
import pytest

from margin_amd import capi, synth
from tests import chain_helpers as ch

pytestmark = pytest.mark.gpu

PHASE_KEYS = ("hap1", "hap2", "genotype", "ancestor", "support1", "support2", "genotype_probs", "hap_probs1", "hap_probs2")
REASONS = ["Same", "NoHet", "MissingConcordancy", "UnlikelyConcordancy", "Discordancy"]


@pytest.mark.parametrize("seed,n_chunks,min_phred,ps_params", [(1, 4, 0, (1, 0.0, 0.5)), (6, 4, 3, (14, 0.02, 0.08)), (8, 4, 0, (12, 0.05, 0.05))])
def test_adjacent_chunks_phase_stitch_and_phase_sets_equal_the_oracle_chain(gpu_ctx, orc, seed, n_chunks, min_phred, ps_params):
    from oracle import frame_oracle as fo
    overlap = 40
    chunks, windows, ids, truth = ch.make_adjacent_chunks(seed, n_chunks=n_chunks, overlap=overlap)
    pd = synth.shipped_phase_params()
    ref = ch.oracle_chain(orc, fo, chunks, windows, ids, pd, min_phred, overlap, ps_params)

    # ---- the product chain: ONE device call for all chunks, then the host functions of rphmm_frame.c ----
    params = capi.Params.from_reference_names(pd)
    dchunks = [capi.DeviceChunk.from_chunk(gpu_ctx, c) for c in chunks]
    got, st = capi.phase_reads_many(gpu_ctx, dchunks, chunks, params)
    assert st.resident == 1 and st.fallback_chunks == 0
    stitch = capi.Stitch()
    switched, hap_of = [], {}
    for c, (chunk, res, oref) in enumerate(zip(chunks, got, ref["results"])):
        assert (res["ref_start"], res["length"]) == (oref["ref_start"], oref["length"])
        for k in PHASE_KEYS:
            assert (np.asarray(res[k]) == np.asarray(oref[k])).all(), (c, k)
        assert res["reads1"] == oref["reads1"] and res["reads2"] == oref["reads2"], c
        recs, _ = capi.read_records(chunk)
        hap, phred = capi.assign_reads_to_haplotypes(chunk.allele_number, chunk.pool, recs, len(chunk.reads), res, min_phred)
        ohap, ophred = ref["assigned"][c]
        assert (hap == ohap).all(), c
        tagged = ohap >= 0
        assert (phred[tagged] == ophred[tagged]).all(), c  # same libm, same operation order: bit for bit
        l1, l2 = ch.partition_lines(chunk, res, hap, phred, min_phred)
        assert (l1, l2) == ref["lines"][c]
        sw, counts = stitch.chunk(l1, l2)
        assert (sw, counts) == (ref["switched"][c], ref["counts"][c]), c
        switched.append(sw)
        # the read sets stitching carries to the next chunk (addToHapReadsSeen, stitching.c:244-283)
        for i in range(len(chunk.reads)):
            if hap[i] in (1, 2) and phred[i] > min_phred:
                hap_of.setdefault(ids[c][i], []).append((hap[i] - 1) ^ int(sw))
    assert stitch.size(1) == len(ref["stitcher"].readsInHap1) and stitch.size(2) == len(ref["stitcher"].readsInHap2)
    for name, p in ref["stitcher"].readsInHap1.items():
        assert stitch.lookup(1, name) == p
    for name, p in ref["stitcher"].readsInHap2.items():
        assert stitch.lookup(2, name) == p
    stitch.close()
    assert any(switched[1:])  # the seeds were chosen (on the oracle chain, tests/test_chain_oracle.py) so that stitching does flip a chunk

    variants = ch.stitched_variants(chunks, windows, ids, got, switched, overlap)
    sets = capi.phase_sets(variants, *ps_params)
    assert [(ps, REASONS[r]) for ps, r in sets] == ref["phase_sets"]

    # and it means something: after the flips the tags of all chunks agree with ONE labelling of the true haplotypes
    votes = [(h, truth[g]) for g, hs in hap_of.items() for h in hs]
    agree = sum(1 for h, t in votes if h == t)
    assert max(agree, len(votes) - agree) >= 0.9 * len(votes)
    for d in dchunks:
        d.close()
