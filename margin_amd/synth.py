"""Seeded synthetic chunks for the stRPHmm hot path (inputs only -- no algorithm lives here).

Two generators:

* :func:`make_ont_chunk` -- BASELINE.json config 2 as restated in SURVEY.md section 8(d): a region
  with biallelic het sites at uniform positions, 30x log-normal reads, 50/50 strand and
  haplotype, 8 % allele error, profile bytes ``min(255, round(30*delta))`` with the supported
  allele at 0 (the encoding of bubbleGraph.c:2423-2435).
* :func:`make_unit_test_chunk` -- the shape used by the reference's own randomised system tests
  (tests/stRPHmmTest.c:13-160): 1..9 alleles per site, chosen allele 0 and every other allele 100.

A chunk is plain numpy: the site table, one packed uint8 profile pool and a per-read table.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np


@dataclass
class Read:
    name: str
    ref_start: int      # first site (stProfileSeq.refStart)
    length: int         # number of sites (stProfileSeq.length)
    strand: int         # 1 = forward
    hap: int            # true haplotype (0/1), for scoring only
    pool_off: int       # offset of profileProbs in Chunk.pool
    nbytes: int


@dataclass
class Chunk:
    allele_number: np.ndarray            # uint32[n_sites]
    allele_offset: np.ndarray            # int64[n_sites + 1]
    sub: np.ndarray                      # uint16[sum A^2]  (site-major, [from*A+to])
    prior: np.ndarray                    # uint16[sum A]
    pool: np.ndarray                     # uint8[pool_bytes]
    reads: List[Read] = field(default_factory=list)
    hap1: Optional[np.ndarray] = None
    hap2: Optional[np.ndarray] = None

    @property
    def n_sites(self) -> int:
        return int(self.allele_number.shape[0])

    @property
    def units(self) -> int:
        """het-sites x reads: sum of stProfileSeq.length (the metric's unit of work)."""
        return int(sum(r.length for r in self.reads))


def _finish(allele_number, reads_raw, hap1, hap2, sub=None) -> Chunk:
    allele_number = np.asarray(allele_number, dtype=np.uint32)
    off = np.zeros(len(allele_number) + 1, dtype=np.int64)
    np.cumsum(allele_number, out=off[1:])
    n_sub = int((allele_number.astype(np.int64) ** 2).sum())
    sub_arr = np.zeros(n_sub, dtype=np.uint16) if sub is None else np.asarray(sub, dtype=np.uint16)
    prior = np.zeros(int(off[-1]), dtype=np.uint16)
    pool_parts, reads, pos = [], [], 0
    for name, start, length, strand, hap, probs in reads_raw:
        probs = np.ascontiguousarray(probs, dtype=np.uint8)
        assert probs.shape[0] == off[start + length] - off[start]
        reads.append(Read(name, int(start), int(length), int(strand), int(hap), pos, int(probs.shape[0])))
        pool_parts.append(probs)
        pos += int(probs.shape[0])
    pool = np.concatenate(pool_parts) if pool_parts else np.zeros(0, dtype=np.uint8)
    return Chunk(allele_number, off, sub_arr, prior, pool, reads, hap1, hap2)


def make_ont_chunk(seed: int = 1, region_bp: int = 1_000_000, n_sites: int = 2000, coverage: float = 30.0,
                   median_len: float = 15_000.0, sigma: float = 0.6, min_len: int = 1000,
                   max_len: int = 100_000, allele_error: float = 0.08,
                   allele_choices: Sequence[int] = (2,), allele_probs: Sequence[float] = (1.0,),
                   length_model: str = "lognormal", normal_sd: float = 3000.0) -> Chunk:
    """SURVEY.md 8(d) config 2 generator (and configs 3-5 by changing the arguments)."""
    r_sites = np.random.default_rng([seed, 1])
    r_reads = np.random.default_rng([seed, 2])
    r_err = np.random.default_rng([seed, 3])
    r_prof = np.random.default_rng([seed, 4])
    pos = np.sort(r_sites.integers(0, region_bp, size=n_sites))
    A = r_sites.choice(np.asarray(allele_choices), size=n_sites, p=np.asarray(allele_probs)).astype(np.uint32)
    hap1 = (r_sites.random(n_sites) * A).astype(np.int64)
    shift = 1 + (r_sites.random(n_sites) * (A - 1)).astype(np.int64)
    hap2 = (hap1 + shift) % A
    off = np.zeros(n_sites + 1, dtype=np.int64)
    np.cumsum(A, out=off[1:])
    target = coverage * region_bp
    total, reads_raw, idx = 0.0, [], 0
    while total < target:
        if length_model == "lognormal":
            ln = float(np.exp(r_reads.normal(np.log(median_len), sigma)))
        else:
            ln = float(r_reads.normal(median_len, normal_sd))
        ln = int(min(max(ln, min_len), max_len))
        start_bp = int(r_reads.integers(-ln + 1, region_bp))
        strand = int(r_reads.random() < 0.5)
        hap = int(r_reads.random() < 0.5)
        lo, hi = max(start_bp, 0), min(start_bp + ln, region_bp)
        total += hi - lo
        s0, s1 = int(np.searchsorted(pos, lo, "left")), int(np.searchsorted(pos, hi, "left"))
        if s1 <= s0:
            continue
        truth = (hap1 if hap == 0 else hap2)[s0:s1]
        a_loc = A[s0:s1].astype(np.int64)
        wrong = r_err.random(s1 - s0) < allele_error
        alt = (truth + 1 + (r_err.random(s1 - s0) * (a_loc - 1)).astype(np.int64)) % a_loc
        observed = np.where(wrong, alt, truth)
        nb = int(off[s1] - off[s0])
        delta = np.abs(r_prof.normal(4.0, 2.0, size=nb))
        probs = np.minimum(255, np.rint(30.0 * delta)).astype(np.uint8)
        probs[(off[s0:s1] - off[s0]) + observed] = 0
        reads_raw.append((f"read_{idx:06d}", s0, s1 - s0, strand, hap, probs))
        idx += 1
    return _finish(A, reads_raw, hap1, hap2)


def make_unit_test_chunk(seed: int, ref_length: int, coverage: int, min_read: int, max_read: int,
                         error_rate: float, max_alleles: int = 9) -> Chunk:
    """Shape of tests/stRPHmmTest.c simulateReads (:106-160) with a fixed seed."""
    rng = np.random.default_rng([seed, 7])
    A = rng.integers(1, max_alleles + 1, size=ref_length).astype(np.uint32)
    hap1 = (rng.random(ref_length) * A).astype(np.int64)
    hap2 = (rng.random(ref_length) * A).astype(np.int64)
    off = np.zeros(ref_length + 1, dtype=np.int64)
    np.cumsum(A, out=off[1:])
    remaining, reads_raw, idx = coverage * ref_length, [], 0
    while remaining > 0:
        hap = int(rng.random() > 0.5)
        ln = int(rng.integers(min_read, max_read + 1))
        start = int(rng.integers(0, ref_length - ln + 1))
        truth = (hap1 if hap == 0 else hap2)[start:start + ln]
        a_loc = A[start:start + ln].astype(np.int64)
        err = rng.random(ln) < error_rate
        observed = np.where(err, (rng.random(ln) * a_loc).astype(np.int64), truth)
        nb = int(off[start + ln] - off[start])
        probs = np.full(nb, 100, dtype=np.uint8)
        probs[(off[start:start + ln] - off[start]) + observed] = 0
        reads_raw.append((f"read_{idx:06d}", start, ln, int(rng.random() < 0.5), hap, probs))
        idx += 1
        remaining -= ln
    return _finish(A, reads_raw, hap1, hap2)


def shipped_phase_params() -> dict:
    """params/base_params.json 'phase' block: the values every BASELINE config runs with."""
    return dict(maxNotSumTransitions=1, minPartitionsInAColumn=100, maxPartitionsInAColumn=100,
                minPosteriorProbabilityForPartition=0.0, maxCoverageDepth=64,
                minReadCoverageToSupportPhasingBetweenHeterozygousSites=2, includeInvertedPartitions=1,
                roundsOfIterativeRefinement=10, includeAncestorSubProb=1)


def unit_test_params(max_partitions: int = 50, max_not_sum: int = 0, min_cov: int = 0) -> dict:
    """tests/stRPHmmTest.c:91-104 getHmmParams (calloc'd, so every other field is 0)."""
    return dict(maxNotSumTransitions=max_not_sum, minPartitionsInAColumn=0, maxPartitionsInAColumn=max_partitions,
                minPosteriorProbabilityForPartition=0.0, maxCoverageDepth=64,
                minReadCoverageToSupportPhasingBetweenHeterozygousSites=min_cov, includeInvertedPartitions=1,
                roundsOfIterativeRefinement=0, includeAncestorSubProb=0)


# ---- read x allele alignment pairs (pair-HMM forward probability) ----

def random_sequence(rng, length: int, n_rate: float = 0.0) -> np.ndarray:
    """uint8 symbols 0..3 (ACGT), 4 = N with probability n_rate"""
    s = rng.integers(0, 4, size=length).astype(np.uint8)
    if n_rate > 0:
        s[rng.random(length) < n_rate] = 4
    return s


def evolve_sequence(rng, s: np.ndarray, sub: float = 0.05, ins: float = 0.03, dele: float = 0.03) -> np.ndarray:
    """substitutions, insertions and deletions at ONT-like rates (the role of evolveSequence in tests/pairwiseAlignerTest.c)"""
    out = []
    for c in s:
        r = rng.random()
        if r < dele:
            continue
        out.append(int(rng.integers(0, 4)) if r < dele + sub else int(c))
        while rng.random() < ins:
            out.append(int(rng.integers(0, 4)))
    return np.array(out, dtype=np.uint8)


def margin_phase_pair_hmm_arrays():
    """"hmmForwardStrandReadGivenReference" of the reference's params/base_params.json (type, transitions, emissions)"""
    transitions = [0.8, 0.1, 0.1, 0.5, 0.5, 0.0, 0.5, 0.0, 0.5]
    emissions = [0.969, 0.005, 0.017, 0.009, 0.008, 0.973, 0.007, 0.012, 0.021, 0.007, 0.967, 0.006, 0.008, 0.008, 0.004, 0.98,
                 1.0, 1.0, 1.0, 1.0, 0.25, 0.25, 0.25, 0.25]
    return 2, transitions, emissions


def make_bubble_strings(seed: int = 1, n_sites: int = 2000, coverage: int = 30, expansion: int = 12, allele_error=(0.05, 0.03, 0.03),
                        duplicate_rate: float = 0.0):
    """The strings margin phase aligns for one chunk of config 2: per het SNP site two alleles (the reference window of
    referenceExpansionForSmallVariants = 12 either side, the site substituted) and ~coverage read substrings, each a noisy
    copy of one allele, strand Bernoulli(0.5).  Returns a list of (alleles, reads, forward_strand) bubbles."""
    rng = np.random.default_rng(seed)
    bubbles = []
    for _ in range(n_sites):
        ref = random_sequence(rng, 2 * expansion + 1)
        alt = ref.copy()
        alt[expansion] = (alt[expansion] + 1 + rng.integers(0, 3)) % 4
        n = max(1, int(rng.poisson(coverage)))
        reads, strands = [], []
        for _k in range(n):
            if reads and rng.random() < duplicate_rate:
                reads.append(reads[int(rng.integers(0, len(reads)))].copy())
            else:
                reads.append(evolve_sequence(rng, ref if rng.random() < 0.5 else alt, *allele_error))
            strands.append(bool(rng.random() < 0.5))
        bubbles.append(([ref, alt], reads, strands))
    return bubbles


def pairs_from_bubbles(bubbles):
    """flatten bubbles into (pool, x_off, x_len, y_off, y_len, model_index): every allele x read pair, model 0 = forward strand"""
    strings, xo, xl, yo, yl, mi = [], [], [], [], [], []
    pos = 0
    for alleles, reads, fwd in bubbles:
        a_at = []
        for a in alleles:
            strings.append(a); a_at.append((pos, len(a))); pos += len(a)
        for r, f in zip(reads, fwd):
            strings.append(r)
            for (ao, al) in a_at:
                xo.append(ao); xl.append(al); yo.append(pos); yl.append(len(r)); mi.append(0 if f else 1)
            pos += len(r)
    pool = np.concatenate(strings) if strings else np.zeros(0, dtype=np.uint8)
    return (pool, np.array(xo, dtype=np.int64), np.array(xl, dtype=np.int32), np.array(yo, dtype=np.int64), np.array(yl, dtype=np.int32),
            np.array(mi, dtype=np.uint8))
