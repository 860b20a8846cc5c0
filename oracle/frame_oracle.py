"""TEST INFRASTRUCTURE, NOT PRODUCT CODE -- restatement of the reference code that frames the read-partitioning path
(SURVEY.md 8 f-2, f-4), written the way the reference is (objects, dicts and sets), each function citing the reference
file:line it follows.  Pure Python on purpose: these are small bookkeeping computations.  Parity is "unpinned" in the
sense of oracle/rphmm_oracle.h: the reference ships no fixtures for these functions and cannot be built here; the
arithmetic is fully determined by the cited source (and by x86-64 float->integer conversion where the source leaves it
undefined, stated at each place).  Only tests/ may import this module."""
import math
import struct


def _f32(x):
    """round a Python float to IEEE single precision (the reference computes with float there)"""
    if math.isnan(x) or math.isinf(x):
        return x
    try:
        return struct.unpack("f", struct.pack("f", x))[0]
    except OverflowError:
        return math.inf if x > 0 else -math.inf


def roundf(x):
    """C roundf: half away from zero, in single precision"""
    x = _f32(x)
    if math.isnan(x) or math.isinf(x):
        return x
    return float(math.floor(abs(x) + 0.5)) * (1.0 if x >= 0 else -1.0) if abs(x) < 2 ** 23 else x


def log_add_exact(x, y):  # sonLib stMath_logAddExact
    if x == -math.inf:
        return y
    if y == -math.inf:
        return x
    return x + math.log(1.0 + math.exp(y - x)) if x > y else y + math.log(1.0 + math.exp(x - y))


def _to_i64_x86(v):  # cvttss2si r64
    if math.isnan(v) or math.isinf(v) or not (-2.0 ** 63 <= v < 2.0 ** 63):
        return -2 ** 63
    return int(v)


def _to_u16_x86(v):  # cvttss2si r32, low 16 bits
    if math.isnan(v) or math.isinf(v) or not (-2.0 ** 31 <= v < 2.0 ** 31):
        return 0
    return int(v) & 0xFFFF


class Bubble:  # inc/margin.h Bubble, the fields bubbleGraph.c:2356-2474 read
    def __init__(self, allele_no, reads, supports):
        self.alleleNo, self.reads, self.alleleReadSupports = allele_no, list(reads), list(supports)
        self.readNo = len(self.reads)
        self.alleleOffset = 0


def get_reference(bubbles, het_substitution_probability):
    """bubbleGraph_getReference bubbleGraph.c:2443-2474 -> (alleleNumber[], substitutionLogProbs[][] flat, priors flat)"""
    allele_number, sub, prior = [], [], []
    off = 0
    for b in bubbles:
        b.alleleOffset = off
        off += b.alleleNo
        allele_number.append(b.alleleNo)
        prior += [0] * b.alleleNo
        lp = -math.log(het_substitution_probability) if het_substitution_probability > 0 else math.inf
        for j in range(b.alleleNo):
            for k in range(b.alleleNo):
                sub.append(0 if j == k else _to_u16_x86(roundf(lp * 30.0)))
    return allele_number, sub, prior


def get_profile_seqs(bubbles):
    """bubbleGraph_getProfileSeqs bubbleGraph.c:2356-2441 -> ordered dict read -> (refStart, length, profileProbs)"""
    off = 0
    for b in bubbles:
        b.alleleOffset = off
        off += b.alleleNo
    total_alleles = off
    read_ends = {}
    for i, b in enumerate(bubbles):
        for r in b.reads:
            read_ends[r] = i
    pseqs = {}
    for i, b in enumerate(bubbles):
        for j, r in enumerate(b.reads):
            if r not in pseqs:
                length = read_ends[r] - i + 1
                last = bubbles[i + length].alleleOffset if i + length < len(bubbles) else total_alleles
                pseqs[r] = dict(refStart=i, length=length, alleleOffset=b.alleleOffset, probs=[0] * (last - b.alleleOffset))
            p = pseqs[r]
            total = -math.inf
            for k in range(b.alleleNo):
                total = log_add_exact(total, float(_f32(b.alleleReadSupports[b.readNo * k + j])))
            ao = b.alleleOffset - p["alleleOffset"]
            for k in range(b.alleleNo):
                lp = float(_f32(b.alleleReadSupports[b.readNo * k + j]))
                with_inf = 30.0 * (total - lp) if not (math.isinf(total) and math.isinf(lp) and total == lp) else math.nan
                l = _to_i64_x86(roundf(with_inf))
                p["probs"][ao + k] = (255 if l > 255 else l) & 0xFF
    return pseqs


def log_prob_of_read_given_haplotype(hap, start, length, pseq, allele_offset):  # genomeFragment.c:71-89
    total = 0.0
    first = allele_offset[pseq["refStart"]]
    for i in range(pseq["length"]):
        j = i + pseq["refStart"] - start
        if 0 <= j < length:
            total -= pseq["probs"][allele_offset[i + pseq["refStart"]] - first + int(hap[j])]
    return total / 30.0


def log_prob_of_being_in_partition(pseq, hap1, hap2, start, length, allele_offset):  # genomeFragment.c:91-100
    i = log_prob_of_read_given_haplotype(hap1, start, length, pseq, allele_offset)
    j = log_prob_of_read_given_haplotype(hap2, start, length, pseq, allele_offset)
    return i - log_add_exact(i, j)


def phase_bam_chunk_reads(gf, pseqs, allele_offset, min_phred):
    """stGenomeFragment_phaseBamChunkReads genomeFragment.c:234-276 -> (set hap1, set hap2, {read: phred})"""
    h1, h2, phreds = set(), set(), {}
    for r, p in pseqs.items():
        if r not in gf["reads1"] and r not in gf["reads2"]:
            continue
        hap1 = r in gf["reads1"]
        lp = (log_prob_of_being_in_partition(p, gf["hap2"], gf["hap1"], gf["refStart"], gf["length"], allele_offset) if hap1 else
              log_prob_of_being_in_partition(p, gf["hap1"], gf["hap2"], gf["refStart"], gf["length"], allele_offset))
        phred = -10 * lp / 2.302585
        phreds[r] = phred
        if not phred < min_phred:
            (h1 if hap1 else h2).add(r)
    return h1, h2, phreds


class Stitcher:
    """chunkToStitch_phaseAdjacentChunks stitching.c:345-403 with addToHapReadsSeen :244-283"""

    def __init__(self):
        self.readsInHap1, self.readsInHap2 = {}, {}

    @staticmethod
    def _intersection(pset, nset, primary_only):  # :306-343
        c = 0
        for name, nl in nset.items():
            if primary_only and nl < 0:
                continue
            if name in pset:
                if primary_only and pset[name] < 0:
                    continue
                c += 1
        return c

    @staticmethod
    def _add(hap, other, to_add):
        for name, prob in to_add.items():
            if name in other:
                if prob > other[name]:
                    del other[name]
                else:
                    continue
            if name not in hap or prob > hap[name]:
                hap[name] = prob

    def chunk(self, hap1_reads, hap2_reads, primary_only=False, do_not_switch=False):
        c1, c2 = dict(hap1_reads), dict(hap2_reads)
        cisH1 = self._intersection(self.readsInHap1, c1, primary_only)
        cisH2 = self._intersection(self.readsInHap2, c2, primary_only)
        transH1 = self._intersection(self.readsInHap2, c1, primary_only)
        transH2 = self._intersection(self.readsInHap1, c2, primary_only)
        switched = False
        if cisH1 + cisH2 < transH2 + transH1 and not do_not_switch:
            c1, c2 = c2, c1
            switched = True
        self._add(self.readsInHap1, self.readsInHap2, c1)
        self._add(self.readsInHap2, self.readsInHap1, c2)
        return switched, (cisH1, cisH2, transH1, transH2)


def binomial_coefficient(n, k):  # bubbleGraph.c:2860-2874, unsigned 128-bit
    M = (1 << 128) - 1
    ans = 1
    k = n - k if k > n - k else k
    j = 1
    while j <= k:
        if n % j == 0:
            ans = (ans * (n // j)) & M
        elif ans % j == 0:
            ans = ((ans // j) * n) & M
        else:
            ans = ((ans * n) & M) // j
        j += 1
        n -= 1
    return ans


def binomial_p_value(n, k):  # bubbleGraph.c:2876-2883 (n / 2 is C integer division)
    k = n - k if k < int(n / 2) else k
    j = 0
    for i in range(k, n + 1):
        j = (j + binomial_coefficient(n, i)) & ((1 << 128) - 1)
    return float(j) / math.pow(2.0, n)


def phase_sets(variants, min_spanning, min_binomial, max_discordant):
    """the phase set rules of writePhasedVcf vcf.c:869-953; variants = dicts(pos, gt1, gt2, alleleIdxToReads=[set,...])"""
    out = []
    prev_het, curr, phase_set = None, None, -1
    for nxt in variants:
        if curr is not None and curr["gt1"] != curr["gt2"]:
            prev_het = curr
        curr = nxt
        gt1, gt2 = curr["gt1"], curr["gt2"]
        determined = False
        if prev_het is not None and gt1 != gt2 and prev_het["gt1"] >= 0 and curr["gt1"] >= 0:
            pH1, pH2 = prev_het["alleleIdxToReads"][prev_het["gt1"]], prev_het["alleleIdxToReads"][prev_het["gt2"]]
            cH1, cH2 = curr["alleleIdxToReads"][gt1], curr["alleleIdxToReads"][gt2]
            hcpv1, hcpv2, hdpv1, hdpv2 = len(pH1 & cH1), len(pH2 & cH2), len(pH2 & cH1), len(pH1 & cH2)
            determined = True
        reason = "Same"
        if gt1 != gt2 and prev_het is None:
            reason = "NoHet"
        elif determined:
            if hcpv1 + hcpv2 < min_spanning:
                reason = "MissingConcordancy"
            elif binomial_p_value(hcpv1 + hcpv2, hcpv1) < min_binomial:
                reason = "UnlikelyConcordancy"
            elif 1.0 * (hdpv1 + hdpv2) / (hcpv1 + hcpv2 + hdpv1 + hdpv2) > max_discordant:
                reason = "Discordancy"
        if reason != "Same":
            phase_set = curr["pos"]
        out.append((phase_set if gt1 != gt2 else -1, reason))
    return out
