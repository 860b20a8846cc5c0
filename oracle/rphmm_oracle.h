/*
 * rphmm_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of margin's read-partitioning HMM layer ("L1" in SURVEY.md):
 * impl/partitions.c, profileSeq.c, emissions.c, column.c, mergeColumn.c, hmm.c, coordination.c,
 * genomeFragment.c and the phasing driver bubbleGraph.c:2673-2801.  Each function cites the
 * reference file:line it follows.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (margin_amd/, libmargin_rphmm.so) never
 * links, imports or calls it.
 *
 * PARITY PINNING.  The reference cannot be built in this environment (externalTools/sonLib and
 * externalTools/htslib are empty submodules, inc/margin.h:23-30 includes both) and its hot-path
 * tests are randomised without a seed (tests/stRPHmmTest.c), so it ships no golden vectors for
 * forward/backward values.  This oracle is pinned by every exact check the reference's own
 * tests hold for the path: the popcount and flipAReadsPartition known answers
 * (stRPHmmTest.c:853-862,1116-1127), the bit-count-vector identity (stRPHmmTest.c:864-928) and
 * the system-test invariants (stRPHmmTest.c:268-550).  Beyond those, forward/backward values are
 * "parity unpinned": max-mode values are integer arithmetic fully determined by the cited source;
 * sum-mode values depend on sonLib's stMath_logAddExact (absent; restated below).
 *
 * Behaviour the reference leaves to sonLib containers and that this oracle fixes
 * deterministically (documented in DESIGN.md): stList_sort2 is taken to be stable; the
 * iteration order of stHash / stSet (pointer-hashed in the reference, hence address dependent)
 * is insertion order here.
 */
#ifndef RPHMM_ORACLE_H_
#define RPHMM_ORACLE_H_

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_ALLELE_LOG_PROB_BITS 8          /* inc/margin.h:135 */
#define ORC_MAX_READ_PARTITIONING_DEPTH 64  /* inc/margin.h:142 */
#define ORC_PROFILE_PROB_SCALAR 30.0        /* inc/margin.h:189 */

typedef struct orc_site {          /* inc/margin.h:164-171 */
    uint64_t alleleNumber;
    uint64_t alleleOffset;
    uint16_t *substitutionLogProbs;
    uint16_t *allelePriorLogProbs;
} orc_site;

typedef struct orc_reference {     /* inc/margin.h:175-180 */
    char name[64];
    uint64_t length;
    uint64_t totalAlleles;
    orc_site *sites;
} orc_reference;

typedef struct orc_profile_seq {   /* inc/margin.h:191-203 */
    orc_reference *ref;
    char readId[64];
    int64_t id;                    /* caller's read index (not in the reference) */
    uint64_t refStart;
    uint64_t length;
    uint64_t alleleOffset;
    uint8_t *profileProbs;
} orc_profile_seq;

typedef struct orc_params {        /* the stRPHmmParameters fields the L1 code reads, inc/margin.h:239-322 */
    int32_t maxNotSumTransitions;
    int64_t minPartitionsInAColumn;
    int64_t maxPartitionsInAColumn;
    double minPosteriorProbabilityForPartition;
    int64_t maxCoverageDepth;
    int64_t minReadCoverageToSupportPhasingBetweenHeterozygousSites;
    int32_t includeInvertedPartitions;
    int64_t roundsOfIterativeRefinement;
    int32_t includeAncestorSubProb;
} orc_params;

typedef struct orc_cell {          /* inc/margin.h:421-425 */
    uint64_t partition;
    double forwardLogProb, backwardLogProb;
    struct orc_cell *nCell;
} orc_cell;

typedef struct orc_merge_cell {    /* inc/margin.h:463-467 */
    uint64_t fromPartition, toPartition;
    double forwardLogProb, backwardLogProb;
} orc_merge_cell;

struct orc_column;
typedef struct orc_merge_column orc_merge_column; /* inc/margin.h:439-445; opaque (two hashes) */

typedef struct orc_column {        /* inc/margin.h:393-402 */
    int64_t refStart, length, depth;
    orc_profile_seq **seqHeaders;
    uint8_t **seqs;
    orc_cell *head;
    orc_merge_column *nColumn, *pColumn;
    double totalLogProb;
} orc_column;

typedef struct orc_hmm {           /* inc/margin.h:340-353 */
    orc_reference *ref;
    int64_t refStart, refLength;
    orc_profile_seq **profileSeqs;
    int64_t nProfileSeqs;
    int64_t columnNumber;
    int64_t maxDepth;
    orc_column *firstColumn, *lastColumn;
    const orc_params *parameters;
    double forwardLogProb, backwardLogProb;
} orc_hmm;

typedef struct orc_genome_fragment { /* inc/margin.h:482-520 */
    orc_reference *reference;
    uint64_t refStart, length;
    int64_t *reads1, *reads2;        /* read ids (orc_profile_seq.id), in insertion order */
    int64_t nReads1, nReads2;
    uint64_t *genotypeString, *haplotypeString1, *haplotypeString2, *ancestorString;
    uint64_t *readsSupportingHaplotype1, *readsSupportingHaplotype2;
    float *genotypeProbs, *haplotypeProbs1, *haplotypeProbs2;
} orc_genome_fragment;

/* error handling: the reference calls st_errAbort; the oracle records the message instead */
const char *orc_last_error(void);
void orc_clear_error(void);

/* partitions.c */
uint64_t orc_makeAcceptMask(uint64_t depth);
uint64_t orc_mergePartitionsOrMasks(uint64_t p1, uint64_t p2, uint64_t d1, uint64_t d2);
uint64_t orc_maskPartition(uint64_t partition, uint64_t mask);
uint64_t orc_invertPartition(uint64_t partition, uint64_t depth);
int orc_seqInHap1(uint64_t partition, int64_t seqIndex);
uint64_t orc_flipAReadsPartition(uint64_t partition, uint64_t readIndex);
int orc_popcount64(uint64_t x);

/* stMath_logAddExact restatement and hmm.c:15-20 */
double orc_logAddExact(double x, double y);
double orc_logAddP(double a, double b, int maxNotSum);

/* reference / profile sequences */
orc_reference *orc_reference_create(const char *name, int64_t nSites, const uint32_t *alleleNumber,
                                    const uint16_t *sub, const uint16_t *prior);
void orc_reference_destroy(orc_reference *ref);
orc_profile_seq *orc_profile_seq_create(orc_reference *ref, const char *readId, int64_t id,
                                        int64_t refStart, int64_t length, const uint8_t *probs);
void orc_profile_seq_destroy(orc_profile_seq *seq);
uint8_t *orc_profile_seq_getProb(orc_profile_seq *seq, uint64_t site, uint64_t allele);

/* emissions.c */
uint64_t *orc_calculateCountBitVectors(uint8_t **seqs, orc_reference *ref, uint64_t firstSite,
                                       uint64_t length, uint64_t depth);
uint64_t orc_getLogProbOfAllele(uint64_t *bitCountVectors, uint64_t depth, uint64_t partition,
                                uint64_t siteOffset, uint64_t allele);
double orc_emissionLogProbability(orc_column *column, orc_cell *cell, uint64_t *bitCountVectors,
                                  orc_reference *ref, const orc_params *params);
/* convenience for tests: emission of one partition over a column given raw row pointers */
double orc_emission_raw(uint8_t **seqs, orc_reference *ref, int64_t firstSite, int64_t length,
                        int64_t depth, uint64_t partition, int includeAncestorSubProb);

/* hmm.c / column.c / mergeColumn.c */
orc_hmm *orc_hmm_construct(orc_profile_seq *seq, const orc_params *params);
void orc_hmm_destruct(orc_hmm *hmm, int destructColumns);
int orc_hmm_overlapOnReference(orc_hmm *a, orc_hmm *b);
int orc_hmm_cmp(const orc_hmm *a, const orc_hmm *b);
orc_hmm *orc_hmm_fuse(orc_hmm *left, orc_hmm *right);
void orc_hmm_alignColumns(orc_hmm *a, orc_hmm *b);
orc_hmm *orc_hmm_createCrossProductOfTwoAlignedHmm(orc_hmm *a, orc_hmm *b);
void orc_hmm_forwardBackward(orc_hmm *hmm);
void orc_hmm_prune(orc_hmm *hmm);
/* returns path as malloc'd array of cell pointers, one per column */
orc_cell **orc_hmm_forwardTraceBack(orc_hmm *hmm, int64_t *pathLength);
orc_hmm *orc_hmm_split(orc_hmm *hmm, int64_t splitPoint);
/* returns malloc'd array of hmms */
orc_hmm **orc_hmm_splitWherePhasingIsUncertain(orc_hmm *hmm, int64_t *nOut);
double orc_cell_posteriorProb(orc_cell *cell, orc_column *column);
double orc_merge_cell_posteriorProb(orc_merge_cell *mCell, orc_merge_column *mColumn);

/* merge column accessors (stRPMergeColumn is opaque) */
uint64_t orc_mcol_maskFrom(orc_merge_column *m);
uint64_t orc_mcol_maskTo(orc_merge_column *m);
int64_t orc_mcol_size(orc_merge_column *m);
orc_merge_cell *orc_mcol_cell(orc_merge_column *m, int64_t i); /* insertion order */
orc_column *orc_mcol_next(orc_merge_column *m);
orc_column *orc_mcol_prev(orc_merge_column *m);
orc_merge_cell *orc_mcol_getNextMergeCell(orc_cell *cell, orc_merge_column *m);     /* mergeColumn.c:63 */
orc_merge_cell *orc_mcol_getPreviousMergeCell(orc_cell *cell, orc_merge_column *m); /* mergeColumn.c:72 */

/* coordination.c */
orc_hmm **orc_getRPHmms(orc_profile_seq **seqs, int64_t n, const orc_params *params, int64_t *nOut);
void orc_filterReadsByCoverageDepth(orc_profile_seq **seqs, int64_t n, const orc_params *params,
                                    orc_profile_seq **filtered, int64_t *nFiltered,
                                    orc_profile_seq **discarded, int64_t *nDiscarded);
/* tiling path count for a read set (coordination.c:224) */
int64_t orc_tilingPathCount(orc_profile_seq **seqs, int64_t n, const orc_params *params);

/* genomeFragment.c + emissions.c:246-343 */
orc_genome_fragment *orc_genome_fragment_construct(orc_hmm *hmm, orc_cell **path, int64_t pathLength);
void orc_genome_fragment_refine(orc_genome_fragment *gF, orc_hmm *hmm, orc_cell **path,
                                int64_t pathLength, int64_t maxIterations);
void orc_genome_fragment_destroy(orc_genome_fragment *gF);
double orc_getLogProbOfReadGivenHaplotype(const uint64_t *hap, int64_t start, int64_t length,
                                          orc_profile_seq *seq, orc_reference *ref);

/* bubbleGraph.c:2673-2801 (phasing driver), given profile sequences and strands instead of a
 * BubbleGraph.  strands[i] != 0 = forward.  Returns the genome fragment; *finalHmm (optional)
 * receives the root hmm after the final sweep (caller destroys). */
orc_genome_fragment *orc_phase_profile_seqs(orc_profile_seq **seqs, const uint8_t *strands, int64_t n,
                                            const orc_params *params, orc_hmm **finalHmm);

/* Observer invoked right after every stRPHmm_forwardBackward issued by the driver/coordination
 * code (coordination.c:312, bubbleGraph.c:2749, hmm.c:1332), before prune mutates the hmm. */
typedef void (*orc_fb_observer)(orc_hmm *hmm, void *user);
void orc_set_fb_observer(orc_fb_observer fn, void *user);
/* replaces the driver's stRPHmm_forwardBackward calls (one) / the sweep loop of one mergeTwoTilingPaths call (many); NULL restores */
void orc_set_fb_override(void (*one)(orc_hmm *), void (*many)(orc_hmm **, int64_t));

/* Wall-clock seconds spent inside orc_hmm_forwardBackward since the last reset, and calls. */
void orc_fb_timer_reset(void);
double orc_fb_timer_seconds(void);
int64_t orc_fb_timer_calls(void);

/*
 * Flattening an hmm into the arrays of mrp_hmm_job (include/margin_rphmm.h).
 * sizes[0..3] = K, sum C, sum M, sum depth.  readByteOff[j] is the offset of column->seqs[i]
 * inside that read's own profileProbs array and readIds[j] the read's id; the caller adds the
 * read's base offset in its profile pool.
 */
void orc_hmm_flat_sizes(orc_hmm *hmm, int64_t sizes[4]);
void orc_hmm_flatten(orc_hmm *hmm, int32_t *colRefStart,
                     int32_t *colLength, int32_t *colDepth, int64_t *colCellOff, int64_t *colReadOff,
                     int64_t *readByteOff, int64_t *readIds, uint64_t *partition, uint64_t *maskFrom,
                     uint64_t *maskTo, int64_t *mcolCellOff, uint64_t *mergeFrom, uint64_t *mergeTo,
                     uint32_t *cellNext, uint32_t *cellPrev, double *cellF, double *cellB,
                     double *mergeF, double *mergeB, double *colTotal);

#ifdef __cplusplus
}
#endif
#endif
