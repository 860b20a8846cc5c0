/*
 * rphmm_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See rphmm_oracle.h for scope,
 * pinning status ("parity unpinned" beyond the reference's own exact tests) and the container
 * order conventions.  Paths cited are relative to the upstream margin tree.
 */
#define _GNU_SOURCE
#include "rphmm_oracle.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------------------------ */
/* errors (reference: st_errAbort -> exit; here: sticky message)                               */
/* ------------------------------------------------------------------------------------------ */
static __thread char g_err[512];

static void orc_fail(const char *fmt, ...) {
    if (g_err[0] != '\0') return; /* keep the first */
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char *orc_last_error(void) { return g_err[0] ? g_err : NULL; }
void orc_clear_error(void) { g_err[0] = '\0'; }

static void *xmalloc(size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
    return p;
}
static void *xcalloc(size_t n, size_t s) {
    void *p = calloc(n ? n : 1, s ? s : 1);
    if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
    return p;
}

/* ------------------------------------------------------------------------------------------ */
/* small containers standing in for sonLib stList / stHash                                     */
/* ------------------------------------------------------------------------------------------ */
typedef struct { void **a; int64_t n, cap; } pvec;

static void pvec_push(pvec *v, void *p) {
    if (v->n == v->cap) {
        v->cap = v->cap ? v->cap * 2 : 8;
        v->a = realloc(v->a, sizeof(void *) * (size_t) v->cap);
    }
    v->a[v->n++] = p;
}
static void *pvec_pop(pvec *v) { return v->a[--v->n]; }
static void pvec_free(pvec *v) { free(v->a); v->a = NULL; v->n = v->cap = 0; }

/* uint64 -> pointer open-addressing map, identity-style hash as in mergeColumn.c:13-19 (mixed) */
typedef struct { uint64_t *k; void **v; int64_t cap, n; } umap;

static inline uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
static void umap_init(umap *m, int64_t expect) {
    int64_t cap = 16;
    while (cap < expect * 2) cap *= 2;
    m->cap = cap; m->n = 0;
    m->k = xmalloc(sizeof(uint64_t) * (size_t) cap);
    m->v = xcalloc((size_t) cap, sizeof(void *));
}
static void umap_free(umap *m) { free(m->k); free(m->v); m->k = NULL; m->v = NULL; m->cap = m->n = 0; }
static void *umap_get(const umap *m, uint64_t key) {
    if (m->cap == 0) return NULL;
    uint64_t i = mix64(key) & (uint64_t) (m->cap - 1);
    while (m->v[i] != NULL) {
        if (m->k[i] == key) return m->v[i];
        i = (i + 1) & (uint64_t) (m->cap - 1);
    }
    return NULL;
}
static void umap_put(umap *m, uint64_t key, void *val);
static void umap_grow(umap *m) {
    umap o = *m;
    umap_init(m, o.cap);
    for (int64_t i = 0; i < o.cap; i++) if (o.v[i]) umap_put(m, o.k[i], o.v[i]);
    free(o.k); free(o.v);
}
static void umap_put(umap *m, uint64_t key, void *val) {
    if (m->cap == 0) umap_init(m, 8);
    if ((m->n + 1) * 2 > m->cap) umap_grow(m);
    uint64_t i = mix64(key) & (uint64_t) (m->cap - 1);
    while (m->v[i] != NULL) {
        if (m->k[i] == key) { m->v[i] = val; return; }
        i = (i + 1) & (uint64_t) (m->cap - 1);
    }
    m->k[i] = key; m->v[i] = val; m->n++;
}

/* stable merge sort of pointers by descending key (stList_sort2 with cellCmpFn, hmm.c:944-962;
 * stability is the stated assumption, see header) */
typedef struct { void *p; double key; } keyed;
static void keyed_sort_desc(keyed *a, int64_t n) {
    if (n < 2) return;
    keyed *tmp = xmalloc(sizeof(keyed) * (size_t) n);
    for (int64_t w = 1; w < n; w *= 2) {
        for (int64_t lo = 0; lo < n; lo += 2 * w) {
            int64_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            int64_t i = lo, j = mid, o = lo;
            while (i < mid && j < hi) {
                /* comparator returns -1 when p1 > p2: element with larger key first; ties keep order */
                if (a[j].key > a[i].key) tmp[o++] = a[j++]; else tmp[o++] = a[i++];
            }
            while (i < mid) tmp[o++] = a[i++];
            while (j < hi) tmp[o++] = a[j++];
        }
        memcpy(a, tmp, sizeof(keyed) * (size_t) n);
    }
    free(tmp);
}

/* ------------------------------------------------------------------------------------------ */
/* partitions.c                                                                                */
/* ------------------------------------------------------------------------------------------ */
uint64_t orc_makeAcceptMask(uint64_t depth) { /* partitions.c:13-19 */
    return depth < 64 ? ~(0xFFFFFFFFFFFFFFFFULL << depth) : 0xFFFFFFFFFFFFFFFFULL;
}
uint64_t orc_mergePartitionsOrMasks(uint64_t p1, uint64_t p2, uint64_t d1, uint64_t d2) { /* :21-28 */
    (void) d2;
    return d1 < 64 ? ((p2 << d1) | p1) : p1; /* shift by 64 is UB in C; d2 must be 0 then */
}
uint64_t orc_maskPartition(uint64_t partition, uint64_t mask) { return partition & mask; } /* :30-35 */
uint64_t orc_invertPartition(uint64_t partition, uint64_t depth) { /* :37-42 */
    return orc_makeAcceptMask(depth) & ~partition;
}
int orc_seqInHap1(uint64_t partition, int64_t seqIndex) { return (int) ((partition >> seqIndex) & 1); } /* :44-51 */
uint64_t orc_flipAReadsPartition(uint64_t partition, uint64_t readIndex) { /* :71-76 */
    return partition ^ ((uint64_t) 1 << readIndex);
}
int orc_popcount64(uint64_t x) { /* emissions.c:38-56; the non-builtin form is used as a cross-check in tests */
    return __builtin_popcountll(x);
}

/* ------------------------------------------------------------------------------------------ */
/* log-space addition                                                                          */
/* ------------------------------------------------------------------------------------------ */
double orc_logAddExact(double x, double y) {
    /* sonLib stMath_logAddExact (source absent from the tree, call site hmm.c:19): exact
     * log(exp(x)+exp(y)) with log-zero guards, evaluated from the larger argument. */
    if (x == -INFINITY) return y;
    if (y == -INFINITY) return x;
    if (x > y) return x + log(1.0 + exp(y - x));
    return y + log(1.0 + exp(x - y));
}
double orc_logAddP(double a, double b, int maxNotSum) { /* hmm.c:15-20 */
    return maxNotSum ? (a > b ? a : b) : orc_logAddExact(a, b);
}

/* ------------------------------------------------------------------------------------------ */
/* reference + profile sequences                                                               */
/* ------------------------------------------------------------------------------------------ */
orc_reference *orc_reference_create(const char *name, int64_t nSites, const uint32_t *alleleNumber,
                                    const uint16_t *sub, const uint16_t *prior) {
    /* bubbleGraph.c:2446-2474 builds the same structure from a BubbleGraph */
    orc_reference *ref = xcalloc(1, sizeof(*ref));
    snprintf(ref->name, sizeof(ref->name), "%s", name ? name : "ref");
    ref->length = (uint64_t) nSites;
    ref->sites = xcalloc((size_t) nSites, sizeof(orc_site));
    uint64_t off = 0; size_t subOff = 0;
    for (int64_t i = 0; i < nSites; i++) {
        orc_site *s = &ref->sites[i];
        uint64_t A = alleleNumber[i];
        s->alleleNumber = A;
        s->alleleOffset = off;
        s->allelePriorLogProbs = xcalloc(A, sizeof(uint16_t));
        s->substitutionLogProbs = xcalloc(A * A, sizeof(uint16_t));
        if (prior) memcpy(s->allelePriorLogProbs, prior + off, sizeof(uint16_t) * A);
        if (sub) memcpy(s->substitutionLogProbs, sub + subOff, sizeof(uint16_t) * A * A);
        off += A; subOff += A * A;
    }
    ref->totalAlleles = off;
    return ref;
}
void orc_reference_destroy(orc_reference *ref) { /* hmm.c:26-35 */
    if (!ref) return;
    for (uint64_t i = 0; i < ref->length; i++) {
        free(ref->sites[i].allelePriorLogProbs);
        free(ref->sites[i].substitutionLogProbs);
    }
    free(ref->sites);
    free(ref);
}
static uint16_t *site_sub(orc_site *site, int64_t from, int64_t to) { /* emissions.c:13-19 */
    return &site->substitutionLogProbs[(uint64_t) from * site->alleleNumber + (uint64_t) to];
}

orc_profile_seq *orc_profile_seq_create(orc_reference *ref, const char *readId, int64_t id,
                                        int64_t refStart, int64_t length, const uint8_t *probs) {
    /* profileSeq.c:13-29 (constructEmptyProfile) followed by a copy of the caller's bytes */
    orc_profile_seq *seq = xcalloc(1, sizeof(*seq));
    seq->ref = ref;
    snprintf(seq->readId, sizeof(seq->readId), "%s", readId ? readId : "");
    seq->id = id;
    seq->refStart = (uint64_t) refStart;
    seq->length = (uint64_t) length;
    seq->alleleOffset = ref->sites[refStart].alleleOffset;
    uint64_t lastAllele = (uint64_t) (refStart + length) < ref->length ? ref->sites[refStart + length].alleleOffset
                                                                       : ref->totalAlleles;
    seq->profileProbs = xcalloc(lastAllele - seq->alleleOffset, sizeof(uint8_t));
    if (probs) memcpy(seq->profileProbs, probs, lastAllele - seq->alleleOffset);
    return seq;
}
void orc_profile_seq_destroy(orc_profile_seq *seq) {
    if (!seq) return;
    free(seq->profileProbs);
    free(seq);
}
uint8_t *orc_profile_seq_getProb(orc_profile_seq *seq, uint64_t site, uint64_t allele) { /* profileSeq.c:41-47 */
    return &seq->profileProbs[seq->ref->sites[site].alleleOffset - seq->alleleOffset + allele];
}

/* ------------------------------------------------------------------------------------------ */
/* emissions.c                                                                                 */
/* ------------------------------------------------------------------------------------------ */
static inline uint64_t *bcv_at(uint64_t *bcv, uint64_t siteOffset, uint64_t allele, uint64_t bit) { /* :67-75 */
    return &bcv[siteOffset * ORC_ALLELE_LOG_PROB_BITS + allele * ORC_ALLELE_LOG_PROB_BITS + bit];
}
static uint64_t bcv_one(uint8_t **seqs, uint64_t depth, uint64_t siteOffset, uint64_t allele, uint64_t bit) { /* :77-89 */
    uint64_t v = 0;
    for (uint64_t i = 0; i < depth; i++) {
        uint8_t *p = &seqs[i][siteOffset];
        v |= ((((uint64_t) p[allele]) >> bit) & 1) << i;
    }
    return v;
}
uint64_t *orc_calculateCountBitVectors(uint8_t **seqs, orc_reference *ref, uint64_t firstSite,
                                       uint64_t length, uint64_t depth) { /* :91-123 */
    if (ref->length == 0) return xmalloc(0);
    uint64_t firstAllele = ref->sites[firstSite].alleleOffset;
    uint64_t lastAllele = firstSite + length < ref->length ? ref->sites[firstSite + length].alleleOffset
                                                           : ref->totalAlleles;
    uint64_t *bcv = xmalloc((lastAllele - firstAllele) * ORC_ALLELE_LOG_PROB_BITS * sizeof(uint64_t));
    for (uint64_t i = firstSite; i < firstSite + length; i++) {
        uint64_t siteOffset = ref->sites[i].alleleOffset - firstAllele;
        for (uint64_t j = 0; j < ref->sites[i].alleleNumber; j++)
            for (uint64_t k = 0; k < ORC_ALLELE_LOG_PROB_BITS; k++)
                *bcv_at(bcv, siteOffset, j, k) = bcv_one(seqs, depth, siteOffset, j, k);
    }
    return bcv;
}
uint64_t orc_getLogProbOfAllele(uint64_t *bcv, uint64_t depth, uint64_t partition, uint64_t siteOffset,
                                uint64_t allele) { /* :125-138 */
    (void) depth;
    uint64_t *j = bcv_at(bcv, siteOffset, allele, 0);
    uint64_t negLogProb = (uint64_t) orc_popcount64(j[0] & partition);
    for (uint64_t i = 1; i < ORC_ALLELE_LOG_PROB_BITS; i++)
        negLogProb += ((uint64_t) orc_popcount64(j[i] & partition)) << i;
    return negLogProb;
}
static inline uint64_t minu(uint64_t a, uint64_t b) { return a < b ? a : b; }

static void allele_hap_probs(orc_site *site, uint64_t depth, uint64_t siteOffset, uint64_t partition,
                             uint64_t *bcv, uint64_t *out) { /* :144-154 */
    for (uint64_t i = 0; i < site->alleleNumber; i++)
        out[i] = orc_getLogProbOfAllele(bcv, depth, partition, siteOffset, i);
}
static void ancestor_hap_probs(orc_site *site, uint64_t *alleleLogProbs, uint64_t *anc) { /* :156-172 */
    for (uint64_t i = 0; i < site->alleleNumber; i++) {
        uint16_t *j = site_sub(site, (int64_t) i, 0);
        anc[i] = alleleLogProbs[0] + j[0];
        for (uint64_t k = 1; k < site->alleleNumber; k++) anc[i] = minu(anc[i], alleleLogProbs[k] + j[k]);
    }
}
static uint64_t min_allele(orc_site *site, uint64_t *alleleLogProbs) { /* :174-185 */
    uint64_t p = alleleLogProbs[0];
    for (uint64_t i = 1; i < site->alleleNumber; i++) p = minu(p, alleleLogProbs[i]);
    return p;
}
static uint64_t genotype_cost(uint64_t depth, orc_site *site, uint64_t siteOffset, uint64_t partition,
                              uint64_t *bcv, int includeAncestorSubProb) { /* :187-219 */
    uint64_t A = site->alleleNumber;
    uint64_t h1[A], h2[A];
    allele_hap_probs(site, depth, siteOffset, partition, bcv, h1);
    allele_hap_probs(site, depth, siteOffset, ~partition, bcv, h2);
    if (!includeAncestorSubProb) return min_allele(site, h1) + min_allele(site, h2);
    uint64_t a1[A], a2[A];
    ancestor_hap_probs(site, h1, a1);
    ancestor_hap_probs(site, h2, a2);
    uint64_t g = a1[0] + a2[0] + site->allelePriorLogProbs[0];
    for (uint64_t i = 1; i < A; i++) g = minu(g, a1[i] + a2[i] + site->allelePriorLogProbs[i]);
    return g;
}
static double emission_core(uint64_t depth, int64_t refStart, int64_t length, uint64_t partition,
                            uint64_t *bcv, orc_reference *ref, int includeAncestorSubProb) { /* :221-240 */
    uint64_t cost = 0;
    uint64_t firstAllele = ref->sites[refStart].alleleOffset;
    for (int64_t i = refStart; i < refStart + length; i++) {
        orc_site *site = &ref->sites[i];
        cost += genotype_cost(depth, site, site->alleleOffset - firstAllele, partition, bcv, includeAncestorSubProb);
    }
    return -((double) cost);
}
double orc_emissionLogProbability(orc_column *column, orc_cell *cell, uint64_t *bcv, orc_reference *ref,
                                  const orc_params *params) {
    return emission_core((uint64_t) column->depth, column->refStart, column->length, cell->partition, bcv, ref,
                         params->includeAncestorSubProb);
}
double orc_emission_raw(uint8_t **seqs, orc_reference *ref, int64_t firstSite, int64_t length, int64_t depth,
                        uint64_t partition, int includeAncestorSubProb) {
    uint64_t *bcv = orc_calculateCountBitVectors(seqs, ref, (uint64_t) firstSite, (uint64_t) length, (uint64_t) depth);
    double e = emission_core((uint64_t) depth, firstSite, length, partition, bcv, ref, includeAncestorSubProb);
    free(bcv);
    return e;
}

/* ------------------------------------------------------------------------------------------ */
/* column.c / mergeColumn.c                                                                    */
/* ------------------------------------------------------------------------------------------ */
struct orc_merge_column {
    uint64_t maskFrom, maskTo;
    umap from, to;   /* mergeCellsFrom / mergeCellsTo (mergeColumn.c:27-31) */
    pvec cells;      /* insertion order: stands in for stHash iteration order */
    orc_column *nColumn, *pColumn;
};

static orc_cell *cell_new(uint64_t partition) { /* column.c:157-161 */
    orc_cell *c = xcalloc(1, sizeof(*c));
    c->partition = partition;
    return c;
}
static orc_column *column_new(int64_t refStart, int64_t length, int64_t depth, orc_profile_seq **hdr,
                              uint8_t **seqs) { /* column.c:12-30 */
    orc_column *c = xcalloc(1, sizeof(*c));
    c->refStart = refStart; c->length = length; c->depth = depth;
    c->seqHeaders = hdr; c->seqs = seqs; c->head = NULL;
    return c;
}
static void column_free(orc_column *c) { /* column.c:32-46 */
    orc_cell *cell = c->head;
    while (cell) { orc_cell *p = cell; cell = cell->nCell; free(p); }
    free(c->seqHeaders); free(c->seqs); free(c);
}
static orc_merge_column *mcol_new(uint64_t maskFrom, uint64_t maskTo) { /* mergeColumn.c:21-33 */
    orc_merge_column *m = xcalloc(1, sizeof(*m));
    m->maskFrom = maskFrom; m->maskTo = maskTo;
    return m;
}
static void mcol_free(orc_merge_column *m) { /* mergeColumn.c:35-39 */
    for (int64_t i = 0; i < m->cells.n; i++) free(m->cells.a[i]);
    pvec_free(&m->cells); umap_free(&m->from); umap_free(&m->to);
    free(m);
}
static orc_merge_cell *mcell_new(uint64_t from, uint64_t to, orc_merge_column *m) { /* mergeColumn.c:92-109 */
    orc_merge_cell *c = xcalloc(1, sizeof(*c));
    c->fromPartition = from; c->toPartition = to;
    umap_put(&m->from, from, c);
    umap_put(&m->to, to, c);
    pvec_push(&m->cells, c);
    return c;
}
orc_merge_cell *orc_mcol_getNextMergeCell(orc_cell *cell, orc_merge_column *m) { /* mergeColumn.c:63-70 */
    return umap_get(&m->from, orc_maskPartition(cell->partition, m->maskFrom));
}
orc_merge_cell *orc_mcol_getPreviousMergeCell(orc_cell *cell, orc_merge_column *m) { /* mergeColumn.c:72-79 */
    return umap_get(&m->to, orc_maskPartition(cell->partition, m->maskTo));
}
uint64_t orc_mcol_maskFrom(orc_merge_column *m) { return m->maskFrom; }
uint64_t orc_mcol_maskTo(orc_merge_column *m) { return m->maskTo; }
int64_t orc_mcol_size(orc_merge_column *m) { return m->cells.n; }
orc_merge_cell *orc_mcol_cell(orc_merge_column *m, int64_t i) { return m->cells.a[i]; }
orc_column *orc_mcol_next(orc_merge_column *m) { return m->nColumn; }
orc_column *orc_mcol_prev(orc_merge_column *m) { return m->pColumn; }

double orc_cell_posteriorProb(orc_cell *cell, orc_column *column) { /* column.c:177-193 */
    double p = exp(cell->forwardLogProb + cell->backwardLogProb - column->totalLogProb);
    if (p > 1.1) orc_fail("ERROR: invalid prob %f", p);
    if (p < 0.0) orc_fail("ERROR: invalid prob %f", p);
    return p > 1.0 ? 1.0 : p;
}
double orc_merge_cell_posteriorProb(orc_merge_cell *mCell, orc_merge_column *mColumn) { /* mergeColumn.c:129-146 */
    double p = exp(mCell->forwardLogProb + mCell->backwardLogProb - mColumn->nColumn->totalLogProb);
    if (p > 1.001) orc_fail("ERROR: invalid prob %f", p);
    if (p < 0) orc_fail("ERROR: invalid prob %f", p);
    return p > 1.0 ? 1.0 : p;
}

static void column_split(orc_column *column, int64_t firstHalfLength, orc_hmm *hmm) { /* column.c:70-130 */
    orc_profile_seq **hdr = xmalloc(sizeof(*hdr) * (size_t) column->depth);
    memcpy(hdr, column->seqHeaders, sizeof(*hdr) * (size_t) column->depth);
    uint8_t **seqs = xmalloc(sizeof(*seqs) * (size_t) column->depth);
    uint64_t firstAllele = hmm->ref->sites[column->refStart].alleleOffset;
    uint64_t lastAllele = hmm->ref->sites[column->refStart + firstHalfLength].alleleOffset;
    for (int64_t i = 0; i < column->depth; i++) seqs[i] = &column->seqs[i][lastAllele - firstAllele];
    orc_column *r = column_new(column->refStart + firstHalfLength, column->length - firstHalfLength, column->depth,
                               hdr, seqs);
    uint64_t accept = orc_makeAcceptMask((uint64_t) column->depth);
    orc_merge_column *m = mcol_new(accept, accept);
    orc_cell *cell = column->head;
    orc_cell **pCell = &r->head;
    do {
        *pCell = cell_new(cell->partition);
        mcell_new(cell->partition, cell->partition, m);
        pCell = &(*pCell)->nCell;
    } while ((cell = cell->nCell) != NULL);
    r->pColumn = m; m->nColumn = r;
    if (column->nColumn == NULL) {
        hmm->lastColumn = r;
    } else {
        column->nColumn->pColumn = r;
        r->nColumn = column->nColumn;
    }
    column->nColumn = m; m->pColumn = column;
    hmm->columnNumber++;
    column->length = firstHalfLength;
}

/* ------------------------------------------------------------------------------------------ */
/* hmm.c: construction, fuse, align, cross product                                             */
/* ------------------------------------------------------------------------------------------ */
static void hmm_add_seqs(orc_hmm *hmm, orc_profile_seq **a, int64_t n) {
    hmm->profileSeqs = realloc(hmm->profileSeqs, sizeof(*a) * (size_t) (hmm->nProfileSeqs + n + 1));
    memcpy(hmm->profileSeqs + hmm->nProfileSeqs, a, sizeof(*a) * (size_t) n);
    hmm->nProfileSeqs += n;
}

int orc_hmm_cmp(const orc_hmm *h1, const orc_hmm *h2) { /* hmm.c:67-95 */
    int i = strcmp(h1->ref->name, h2->ref->name);
    if (i == 0) {
        i = h1->refStart > h2->refStart ? 1 : h1->refStart < h2->refStart ? -1 : 0;
        if (i == 0) {
            i = h2->refLength > h1->refLength ? 1 : h2->refLength < h1->refLength ? -1 : 0;
            if (i == 0) {
                if (h1->nProfileSeqs > 0 && h2->nProfileSeqs > 0)
                    i = strcmp(h1->profileSeqs[0]->readId, h2->profileSeqs[0]->readId);
                if (i == 0) i = h1 > h2 ? 1 : (h1 < h2 ? -1 : 0);
            }
        }
    }
    return i;
}

orc_hmm *orc_hmm_construct(orc_profile_seq *seq, const orc_params *params) { /* hmm.c:97-133 */
    orc_hmm *hmm = xcalloc(1, sizeof(*hmm));
    hmm->ref = seq->ref;
    hmm->refStart = (int64_t) seq->refStart;
    hmm->refLength = (int64_t) seq->length;
    hmm_add_seqs(hmm, &seq, 1);
    hmm->parameters = params;
    hmm->columnNumber = 1;
    hmm->maxDepth = 1;
    orc_profile_seq **hdr = xmalloc(sizeof(*hdr));
    hdr[0] = seq;
    uint8_t **seqs = xmalloc(sizeof(*seqs));
    seqs[0] = seq->profileProbs;
    orc_column *column = column_new(hmm->refStart, hmm->refLength, 1, hdr, seqs);
    hmm->firstColumn = hmm->lastColumn = column;
    orc_cell *cell = cell_new(1);
    column->head = cell;
    cell->nCell = cell_new(0);
    return hmm;
}

void orc_hmm_destruct(orc_hmm *hmm, int destructColumns) { /* hmm.c:135-156 */
    if (!hmm) return;
    free(hmm->profileSeqs);
    if (destructColumns) {
        orc_column *column = hmm->firstColumn;
        while (1) {
            orc_merge_column *m = column->nColumn;
            column_free(column);
            if (m == NULL) break;
            column = m->nColumn;
            mcol_free(m);
        }
    }
    free(hmm);
}

int orc_hmm_overlapOnReference(orc_hmm *h1, orc_hmm *h2) { /* hmm.c:1165-1190 */
    if (h1->refLength <= 0 || h2->refLength <= 0) {
        orc_fail("Trying to compare HMMs with a zero length coordinate interval");
        return 0;
    }
    if (strcmp(h1->ref->name, h2->ref->name) != 0) return 0;
    if (h1->refStart > h2->refStart) return orc_hmm_overlapOnReference(h2, h1);
    return h1->refStart + h1->refLength > h2->refStart;
}

orc_hmm *orc_hmm_fuse(orc_hmm *left, orc_hmm *right) { /* hmm.c:283-372 */
    if (strcmp(left->ref->name, right->ref->name) != 0) {
        orc_fail("Attempting to fuse two hmms not on the same reference sequence"); return NULL;
    }
    if (orc_hmm_overlapOnReference(left, right)) {
        orc_fail("Attemping to fuse two hmms that overlap in reference coordinates"); return NULL;
    }
    if (left->refStart >= right->refStart) {
        orc_fail("Left hmm does not precede right hmm in reference coordinates for merge"); return NULL;
    }
    if (left->parameters != right->parameters) {
        orc_fail("HMM parameters differ in fuse function, panic."); return NULL;
    }
    orc_hmm *hmm = xcalloc(1, sizeof(*hmm));
    hmm->ref = left->ref;
    hmm->refStart = left->refStart;
    hmm->refLength = right->refStart + right->refLength - left->refStart;
    hmm_add_seqs(hmm, left->profileSeqs, left->nProfileSeqs);
    hmm_add_seqs(hmm, right->profileSeqs, right->nProfileSeqs);
    hmm->columnNumber = left->columnNumber + right->columnNumber;
    hmm->maxDepth = left->maxDepth > right->maxDepth ? left->maxDepth : right->maxDepth;
    hmm->parameters = left->parameters;

    orc_merge_column *m = mcol_new(0, 0);
    left->lastColumn->nColumn = m;
    m->pColumn = left->lastColumn;
    mcell_new(0, 0, m);
    int64_t gap = right->refStart - (left->refStart + left->refLength);
    if (gap > 0) {
        orc_column *column = column_new(left->refStart + left->refLength, gap, 0, NULL, NULL);
        m->nColumn = column; column->pColumn = m;
        column->head = cell_new(0);
        m = mcol_new(0, 0);
        mcell_new(0, 0, m);
        column->nColumn = m; m->pColumn = column;
        hmm->columnNumber += 1;
    }
    m->nColumn = right->firstColumn;
    right->firstColumn->pColumn = m;
    hmm->firstColumn = left->firstColumn;
    hmm->lastColumn = right->lastColumn;
    orc_hmm_destruct(left, 0);
    orc_hmm_destruct(right, 0);
    return hmm;
}

void orc_hmm_alignColumns(orc_hmm *h1, orc_hmm *h2) { /* hmm.c:374-507 */
    if (!orc_hmm_overlapOnReference(h1, h2)) {
        orc_fail("Attempting to align two HMMs that do not overlap in reference coordinate space"); return;
    }
    if (h1->refStart > h2->refStart) { orc_hmm_alignColumns(h2, h1); return; }
    if (h1->refStart < h2->refStart) { /* :396-424 empty prefix for h2 */
        orc_column *column = column_new(h1->refStart, h2->refStart - h1->refStart, 0, NULL, NULL);
        column->head = cell_new(0);
        orc_merge_column *m = mcol_new(0, 0);
        mcell_new(0, 0, m);
        h2->firstColumn->pColumn = m; m->nColumn = h2->firstColumn;
        m->pColumn = column; column->nColumn = m;
        h2->firstColumn = column;
        h2->refLength += h2->refStart - h1->refStart;
        h2->refStart = h1->refStart;
        h2->columnNumber++;
    }
    if (h1->refLength < h2->refLength) { orc_hmm_alignColumns(h2, h1); return; }
    if (h1->refLength > h2->refLength) { /* :435-462 empty suffix for h2 */
        orc_column *column = column_new(h2->lastColumn->refStart + h2->lastColumn->length,
                                        h1->refLength - h2->refLength, 0, NULL, NULL);
        column->head = cell_new(0);
        orc_merge_column *m = mcol_new(0, 0);
        mcell_new(0, 0, m);
        h2->lastColumn->nColumn = m; m->pColumn = h2->lastColumn;
        m->nColumn = column; column->pColumn = m;
        h2->lastColumn = column;
        h2->refLength = h1->refLength;
        h2->columnNumber++;
    }
    orc_column *c1 = h1->firstColumn, *c2 = h2->firstColumn;
    while (1) { /* :476-504 */
        if (c1->length > c2->length) column_split(c1, c2->length, h1);
        else if (c1->length < c2->length) column_split(c2, c1->length, h2);
        if (c1->nColumn == NULL) break;
        c1 = c1->nColumn->nColumn;
        c2 = c2->nColumn->nColumn;
    }
}

orc_hmm *orc_hmm_createCrossProductOfTwoAlignedHmm(orc_hmm *h1, orc_hmm *h2) { /* hmm.c:534-750 */
    if (strcmp(h1->ref->name, h2->ref->name) != 0) {
        orc_fail("Trying to create cross product of two HMMs on different reference sequences"); return NULL;
    }
    if (h1->refStart != h2->refStart) {
        orc_fail("Trying to create cross product of two HMMs with different reference interval starts"); return NULL;
    }
    if (h1->refLength != h2->refLength) {
        orc_fail("Trying to create cross product of two HMMs with different reference interval length"); return NULL;
    }
    if (h1->columnNumber != h2->columnNumber) {
        orc_fail("Trying to create cross product of two HMMs with different column numbers"); return NULL;
    }
    if (h1->parameters != h2->parameters) {
        orc_fail("Hmm parameters differ in fuse function, panic."); return NULL;
    }
    orc_hmm *hmm = xcalloc(1, sizeof(*hmm));
    hmm->ref = h1->ref; hmm->refStart = h1->refStart; hmm->refLength = h1->refLength;
    hmm_add_seqs(hmm, h1->profileSeqs, h1->nProfileSeqs);
    hmm_add_seqs(hmm, h2->profileSeqs, h2->nProfileSeqs);
    hmm->columnNumber = h1->columnNumber;
    hmm->parameters = h1->parameters;

    orc_column *c1 = h1->firstColumn, *c2 = h2->firstColumn;
    orc_merge_column *mColumn = NULL;
    while (1) {
        int64_t depth = c1->depth + c2->depth;
        if (depth > hmm->maxDepth) hmm->maxDepth = depth;
        orc_profile_seq **hdr = xmalloc(sizeof(*hdr) * (size_t) (depth ? depth : 1));
        uint8_t **seqs = xmalloc(sizeof(*seqs) * (size_t) (depth ? depth : 1));
        if (c1->depth) { memcpy(hdr, c1->seqHeaders, sizeof(*hdr) * (size_t) c1->depth); memcpy(seqs, c1->seqs, sizeof(*seqs) * (size_t) c1->depth); }
        if (c2->depth) { memcpy(hdr + c1->depth, c2->seqHeaders, sizeof(*hdr) * (size_t) c2->depth); memcpy(seqs + c1->depth, c2->seqs, sizeof(*seqs) * (size_t) c2->depth); }
        orc_column *column = column_new(c1->refStart, c1->length, depth, hdr, seqs);
        if (mColumn != NULL) { mColumn->nColumn = column; column->pColumn = mColumn; }
        else hmm->firstColumn = column;

        orc_cell **pCell = &column->head;
        orc_cell *cell1 = c1->head;
        if (hmm->parameters->includeInvertedPartitions) { /* :627-655 */
            umap seen; umap_init(&seen, 64);
            do {
                orc_cell *cell2 = c2->head;
                do {
                    uint64_t p = orc_mergePartitionsOrMasks(cell1->partition, cell2->partition, (uint64_t) c1->depth,
                                                            (uint64_t) c2->depth);
                    if (umap_get(&seen, p) == NULL) {
                        orc_cell *c = cell_new(p);
                        umap_put(&seen, p, c);
                        *pCell = c; pCell = &c->nCell;
                        if (depth > 0) {
                            uint64_t ip = orc_invertPartition(p, (uint64_t) depth);
                            orc_cell *ic = cell_new(ip);
                            umap_put(&seen, ip, ic);
                            *pCell = ic; pCell = &ic->nCell;
                        }
                    }
                } while ((cell2 = cell2->nCell) != NULL);
            } while ((cell1 = cell1->nCell) != NULL);
            umap_free(&seen);
        } else { /* :657-668 */
            do {
                orc_cell *cell2 = c2->head;
                do {
                    orc_cell *c = cell_new(orc_mergePartitionsOrMasks(cell1->partition, cell2->partition,
                                                                      (uint64_t) c1->depth, (uint64_t) c2->depth));
                    *pCell = c; pCell = &c->nCell;
                } while ((cell2 = cell2->nCell) != NULL);
            } while ((cell1 = cell1->nCell) != NULL);
        }

        orc_merge_column *m1 = c1->nColumn, *m2 = c2->nColumn;
        if (m1 == NULL) { hmm->lastColumn = column; break; }

        uint64_t fromMask = orc_mergePartitionsOrMasks(m1->maskFrom, m2->maskFrom, (uint64_t) m1->pColumn->depth,
                                                       (uint64_t) m2->pColumn->depth);
        uint64_t toMask = orc_mergePartitionsOrMasks(m1->maskTo, m2->maskTo, (uint64_t) m1->nColumn->depth,
                                                     (uint64_t) m2->nColumn->depth);
        mColumn = mcol_new(fromMask, toMask);
        mColumn->pColumn = column; column->nColumn = mColumn;
        for (int64_t i = 0; i < m1->cells.n; i++) { /* :699-740, hash iteration = insertion order here */
            orc_merge_cell *mc1 = m1->cells.a[i];
            for (int64_t j = 0; j < m2->cells.n; j++) {
                orc_merge_cell *mc2 = m2->cells.a[j];
                uint64_t from = orc_mergePartitionsOrMasks(mc1->fromPartition, mc2->fromPartition,
                                                           (uint64_t) m1->pColumn->depth, (uint64_t) m2->pColumn->depth);
                uint64_t to = orc_mergePartitionsOrMasks(mc1->toPartition, mc2->toPartition,
                                                         (uint64_t) m1->nColumn->depth, (uint64_t) m2->nColumn->depth);
                if (hmm->parameters->includeInvertedPartitions) {
                    if (umap_get(&mColumn->from, from) == NULL) {
                        mcell_new(from, to, mColumn);
                        if (orc_popcount64(fromMask) > 0) {
                            uint64_t ifrom = mColumn->maskFrom &
                                             orc_invertPartition(from, (uint64_t) (m1->pColumn->depth + m2->pColumn->depth));
                            uint64_t ito = mColumn->maskTo &
                                           orc_invertPartition(to, (uint64_t) (m1->nColumn->depth + m2->nColumn->depth));
                            mcell_new(ifrom, ito, mColumn);
                        }
                    }
                } else {
                    mcell_new(from, to, mColumn);
                }
            }
        }
        c1 = m1->nColumn; c2 = m2->nColumn;
    }
    return hmm;
}

/* ------------------------------------------------------------------------------------------ */
/* hmm.c: forward / backward (the hot path)                                                    */
/* ------------------------------------------------------------------------------------------ */
static orc_fb_observer g_observer; static void *g_observer_user;
static __thread double g_fb_seconds; static __thread int64_t g_fb_calls; /* per thread: one chunk per thread (phase.c:276) */
void orc_set_fb_observer(orc_fb_observer fn, void *user) { g_observer = fn; g_observer_user = user; }
void orc_fb_timer_reset(void) { g_fb_seconds = 0.0; g_fb_calls = 0; }
double orc_fb_timer_seconds(void) { return g_fb_seconds; }
int64_t orc_fb_timer_calls(void) { return g_fb_calls; }

static void initialise_probs(orc_hmm *hmm) { /* hmm.c:752-789 */
    hmm->forwardLogProb = -INFINITY;
    hmm->backwardLogProb = -INFINITY;
    orc_column *column = hmm->firstColumn;
    while (1) {
        column->totalLogProb = -INFINITY;
        orc_cell *cell = column->head;
        do { cell->forwardLogProb = -INFINITY; cell->backwardLogProb = -INFINITY; } while ((cell = cell->nCell) != NULL);
        if (column->nColumn == NULL) break;
        orc_merge_column *m = column->nColumn;
        for (int64_t i = 0; i < m->cells.n; i++) {
            orc_merge_cell *mc = m->cells.a[i];
            mc->forwardLogProb = -INFINITY; mc->backwardLogProb = -INFINITY;
        }
        column = m->nColumn;
    }
}
static void forward_pass(orc_hmm *hmm) { /* hmm.c:827-879 with forwardCellCalc1/2 :791-825 */
    const int maxNotSum = hmm->parameters->maxNotSumTransitions;
    orc_column *column = hmm->firstColumn;
    while (1) {
        uint64_t *bcv = orc_calculateCountBitVectors(column->seqs, hmm->ref, (uint64_t) column->refStart,
                                                     (uint64_t) column->length, (uint64_t) column->depth);
        orc_cell *cell = column->head;
        do {
            if (column->pColumn != NULL) {
                orc_merge_cell *mc = orc_mcol_getPreviousMergeCell(cell, column->pColumn);
                if (!mc) { orc_fail("forward: missing previous merge cell"); free(bcv); return; }
                cell->forwardLogProb = mc->forwardLogProb;
            } else {
                cell->forwardLogProb = 0.0;
            }
            double e = orc_emissionLogProbability(column, cell, bcv, hmm->ref, hmm->parameters);
            cell->forwardLogProb += e;
            cell->backwardLogProb = e; /* stash, hmm.c:809-811 */
            if (column->nColumn != NULL) {
                orc_merge_cell *mc = orc_mcol_getNextMergeCell(cell, column->nColumn);
                if (!mc) { orc_fail("forward: missing next merge cell"); free(bcv); return; }
                mc->forwardLogProb = orc_logAddP(mc->forwardLogProb, cell->forwardLogProb, maxNotSum);
            } else {
                hmm->forwardLogProb = orc_logAddP(hmm->forwardLogProb, cell->forwardLogProb, maxNotSum);
            }
        } while ((cell = cell->nCell) != NULL);
        free(bcv);
        if (column->nColumn == NULL) break;
        column = column->nColumn->nColumn;
    }
}
static void backward_pass(orc_hmm *hmm) { /* hmm.c:910-929 with backwardCellCalc :881-908 */
    const int maxNotSum = hmm->parameters->maxNotSumTransitions;
    orc_column *column = hmm->lastColumn;
    while (1) {
        orc_cell *cell = column->head;
        do {
            double p = cell->backwardLogProb;
            if (column->nColumn != NULL) {
                orc_merge_cell *mc = orc_mcol_getNextMergeCell(cell, column->nColumn);
                cell->backwardLogProb = mc->backwardLogProb;
                p += mc->backwardLogProb;
            } else {
                cell->backwardLogProb = 0.0;
            }
            if (column->pColumn != NULL) {
                orc_merge_cell *mc = orc_mcol_getPreviousMergeCell(cell, column->pColumn);
                mc->backwardLogProb = orc_logAddP(mc->backwardLogProb, p, maxNotSum);
            } else {
                hmm->backwardLogProb = orc_logAddP(hmm->backwardLogProb, p, maxNotSum);
            }
            column->totalLogProb = orc_logAddP(column->totalLogProb, cell->forwardLogProb + cell->backwardLogProb,
                                               maxNotSum);
        } while ((cell = cell->nCell) != NULL);
        if (column->pColumn == NULL) break;
        column = column->pColumn->pColumn;
    }
}
void orc_hmm_forwardBackward(orc_hmm *hmm) { /* hmm.c:931-942 */
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    initialise_probs(hmm);
    forward_pass(hmm);
    if (!g_err[0]) backward_pass(hmm);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    g_fb_seconds += (double) (t1.tv_sec - t0.tv_sec) + 1e-9 * (double) (t1.tv_nsec - t0.tv_nsec);
    g_fb_calls++;
}
/* The seam, replaceable (tools/adaptor_probe.py times the product's adaptor in it): `one` stands in for
 * stRPHmm_forwardBackward (hmm.c:931) wherever the driver calls it; `many`, if set, is given all cross products of one
 * mergeTwoTilingPaths call (the loop of coordination.c:285-328) at once, before any of them is pruned. */
static void (*g_fb_override)(orc_hmm *);
static void (*g_fb_many_override)(orc_hmm **, int64_t);
void orc_set_fb_override(void (*one)(orc_hmm *), void (*many)(orc_hmm **, int64_t)) { g_fb_override = one; g_fb_many_override = many; }
static void fb_and_notify(orc_hmm *hmm) {
    if (g_fb_override) {
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        g_fb_override(hmm);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        g_fb_seconds += (double) (t1.tv_sec - t0.tv_sec) + 1e-9 * (double) (t1.tv_nsec - t0.tv_nsec);
        g_fb_calls++;
    } else orc_hmm_forwardBackward(hmm);
    if (g_observer) g_observer(hmm, g_observer_user);
}

/* ------------------------------------------------------------------------------------------ */
/* hmm.c: prune                                                                                */
/* ------------------------------------------------------------------------------------------ */
static void filter_merge_cells(orc_merge_column *m, umap *chosen) { /* hmm.c:964-987 */
    pvec kept = {0};
    for (int64_t i = 0; i < m->cells.n; i++) {
        orc_merge_cell *mc = m->cells.a[i];
        if (umap_get(chosen, (uint64_t) (uintptr_t) mc) != NULL) pvec_push(&kept, mc);
        else free(mc);
    }
    pvec_free(&m->cells); umap_free(&m->from); umap_free(&m->to);
    m->cells = kept;
    umap_init(&m->from, kept.n); umap_init(&m->to, kept.n);
    for (int64_t i = 0; i < kept.n; i++) {
        orc_merge_cell *mc = kept.a[i];
        umap_put(&m->from, mc->fromPartition, mc);
        umap_put(&m->to, mc->toPartition, mc);
    }
}
typedef orc_merge_cell *(*mcell_getter)(orc_cell *, orc_merge_column *);

/* getLinkedCells hmm.c:1021-1047: keep cells still linked to mColumn, sorted by posterior desc */
static keyed *linked_cells(orc_column *column, mcell_getter getP, orc_merge_column *mColumn, int64_t *nOut) {
    int64_t cap = 64, n = 0;
    keyed *a = xmalloc(sizeof(keyed) * (size_t) cap);
    orc_cell *cell = column->head;
    do {
        if (mColumn == NULL || getP(cell, mColumn) != NULL) {
            if (n == cap) { cap *= 2; a = realloc(a, sizeof(keyed) * (size_t) cap); }
            a[n].p = cell; a[n].key = orc_cell_posteriorProb(cell, column); n++;
            cell = cell->nCell;
        } else {
            orc_cell *nCell = cell->nCell;
            free(cell);
            cell = nCell;
        }
    } while (cell != NULL);
    keyed_sort_desc(a, n);
    *nOut = n;
    return a;
}
static void relink_cells(orc_column *column, keyed *a, int64_t n) { /* hmm.c:1006-1019 */
    orc_cell **pCell = &column->head;
    for (int64_t i = 0; i < n; i++) { orc_cell *c = a[i].p; *pCell = c; pCell = &c->nCell; }
    *pCell = NULL;
}
static void prune_forwards(orc_hmm *hmm) { /* hmm.c:1049-1109 */
    const orc_params *P = hmm->parameters;
    orc_column *column = hmm->firstColumn;
    orc_merge_column *mColumn = NULL;
    while (1) {
        int64_t n;
        keyed *cells = linked_cells(column, orc_mcol_getPreviousMergeCell, mColumn, &n);
        while (n > P->minPartitionsInAColumn &&
               (n > P->maxPartitionsInAColumn || cells[n - 1].key < P->minPosteriorProbabilityForPartition)) {
            free(cells[--n].p);
        }
        relink_cells(column, cells, n);
        mColumn = column->nColumn;
        if (mColumn == NULL) { free(cells); break; }
        /* getLinkedMergeCells hmm.c:989-1004; stSet_getList order = first insertion (see header) */
        umap chosen; umap_init(&chosen, n);
        int64_t mn = 0;
        keyed *mcs = xmalloc(sizeof(keyed) * (size_t) (n ? n : 1));
        for (int64_t i = 0; i < n; i++) {
            orc_merge_cell *mc = orc_mcol_getNextMergeCell(cells[i].p, mColumn);
            if (!mc) { orc_fail("prune: missing next merge cell"); break; }
            if (umap_get(&chosen, (uint64_t) (uintptr_t) mc) == NULL) {
                umap_put(&chosen, (uint64_t) (uintptr_t) mc, mc);
                mcs[mn].p = mc; mcs[mn].key = orc_merge_cell_posteriorProb(mc, mColumn); mn++;
            }
        }
        keyed_sort_desc(mcs, mn);
        umap kept; umap_init(&kept, mn);
        while (mn > P->minPartitionsInAColumn &&
               (mn > P->maxPartitionsInAColumn || mcs[mn - 1].key < P->minPosteriorProbabilityForPartition)) {
            mn--;
        }
        for (int64_t i = 0; i < mn; i++) umap_put(&kept, (uint64_t) (uintptr_t) mcs[i].p, mcs[i].p);
        filter_merge_cells(mColumn, &kept);
        umap_free(&kept); umap_free(&chosen); free(mcs); free(cells);
        column = mColumn->nColumn;
    }
}
static void prune_backwards(orc_hmm *hmm) { /* hmm.c:1111-1158 */
    orc_column *column = hmm->lastColumn;
    orc_merge_column *mColumn = NULL;
    while (1) {
        int64_t n;
        keyed *cells = linked_cells(column, orc_mcol_getNextMergeCell, mColumn, &n);
        relink_cells(column, cells, n);
        mColumn = column->pColumn;
        if (mColumn == NULL) { free(cells); break; }
        umap chosen; umap_init(&chosen, n);
        for (int64_t i = 0; i < n; i++) {
            orc_merge_cell *mc = orc_mcol_getPreviousMergeCell(cells[i].p, mColumn);
            if (mc) umap_put(&chosen, (uint64_t) (uintptr_t) mc, mc);
        }
        filter_merge_cells(mColumn, &chosen);
        umap_free(&chosen); free(cells);
        column = mColumn->pColumn;
    }
}
void orc_hmm_prune(orc_hmm *hmm) { /* hmm.c:1160-1163 */
    prune_forwards(hmm);
    prune_backwards(hmm);
}

/* ------------------------------------------------------------------------------------------ */
/* hmm.c: trace back, split                                                                    */
/* ------------------------------------------------------------------------------------------ */
orc_cell **orc_hmm_forwardTraceBack(orc_hmm *hmm, int64_t *pathLength) { /* hmm.c:165-219 */
    pvec path = {0};
    orc_column *column = hmm->lastColumn;
    orc_cell *cell = column->head;
    double maxProb = cell->forwardLogProb;
    orc_cell *maxCell = cell;
    while ((cell = cell->nCell) != NULL) {
        if (cell->forwardLogProb > maxProb) { maxProb = cell->forwardLogProb; maxCell = cell; }
    }
    pvec_push(&path, maxCell);
    while (column->pColumn != NULL) {
        orc_merge_cell *mc = orc_mcol_getPreviousMergeCell(maxCell, column->pColumn);
        column = column->pColumn->pColumn;
        cell = column->head;
        maxCell = NULL;
        maxProb = -INFINITY;
        do {
            if (orc_mcol_getNextMergeCell(cell, column->nColumn) == mc && cell->forwardLogProb > maxProb) {
                maxProb = cell->forwardLogProb; maxCell = cell;
            }
        } while ((cell = cell->nCell) != NULL);
        if (maxCell == NULL) { orc_fail("traceback: no compatible cell"); break; }
        pvec_push(&path, maxCell);
    }
    /* reverse (hmm.c:216) */
    for (int64_t i = 0, j = path.n - 1; i < j; i++, j--) { void *t = path.a[i]; path.a[i] = path.a[j]; path.a[j] = t; }
    *pathLength = path.n;
    return (orc_cell **) path.a;
}

static orc_column *get_column(orc_column *column, int64_t site) { /* hmm.c:1192-1209 */
    while (1) {
        if (site < column->refStart + column->length) return column;
        if (column->nColumn == NULL) break;
        column = column->nColumn->nColumn;
    }
    orc_fail("Site: %lld not contained in hmm", (long long) site);
    return column;
}
static void reset_column_number_and_depth(orc_hmm *hmm) { /* hmm.c:1211-1229 */
    hmm->columnNumber = 0; hmm->maxDepth = 0;
    orc_column *column = hmm->firstColumn;
    while (1) {
        hmm->columnNumber++;
        if (hmm->maxDepth < column->depth) hmm->maxDepth = column->depth;
        if (column->nColumn == NULL) break;
        column = column->nColumn->nColumn;
    }
}
orc_hmm *orc_hmm_split(orc_hmm *hmm, int64_t splitPoint) { /* hmm.c:1231-1300 */
    if (splitPoint <= hmm->refStart) { orc_fail("The split point is at or before the start of the reference interval"); return NULL; }
    if (splitPoint >= hmm->refStart + hmm->refLength) { orc_fail("The split point is after the last position of the reference interval"); return NULL; }
    orc_hmm *suffix = xcalloc(1, sizeof(*suffix));
    suffix->ref = hmm->ref;
    suffix->refStart = splitPoint;
    suffix->refLength = hmm->refLength + hmm->refStart - splitPoint;
    hmm->refLength = splitPoint - hmm->refStart;
    suffix->parameters = hmm->parameters;
    orc_profile_seq **prefixSeqs = xmalloc(sizeof(*prefixSeqs) * (size_t) (hmm->nProfileSeqs + 1));
    int64_t nPrefix = 0;
    for (int64_t i = 0; i < hmm->nProfileSeqs; i++) {
        orc_profile_seq *s = hmm->profileSeqs[i];
        if ((int64_t) s->refStart < splitPoint) prefixSeqs[nPrefix++] = s;
        if ((int64_t) (s->refStart + s->length) > splitPoint) hmm_add_seqs(suffix, &s, 1);
    }
    free(hmm->profileSeqs);
    hmm->profileSeqs = prefixSeqs; hmm->nProfileSeqs = nPrefix;
    orc_column *splitColumn = get_column(hmm->firstColumn, splitPoint);
    if (splitPoint > splitColumn->refStart) {
        column_split(splitColumn, splitPoint - splitColumn->refStart, hmm);
        splitColumn = splitColumn->nColumn->nColumn;
    }
    suffix->firstColumn = splitColumn;
    suffix->lastColumn = hmm->lastColumn;
    hmm->lastColumn = splitColumn->pColumn->pColumn;
    hmm->lastColumn->nColumn = NULL;
    mcol_free(splitColumn->pColumn);
    splitColumn->pColumn = NULL;
    reset_column_number_and_depth(hmm);
    reset_column_number_and_depth(suffix);
    return suffix;
}
static int sites_linkage_well_supported(orc_hmm *hmm, int64_t leftSite, int64_t rightSite) { /* hmm.c:1302-1320 */
    orc_column *l = get_column(hmm->firstColumn, leftSite);
    orc_column *r = get_column(l, rightSite);
    int64_t common = 0;
    for (int64_t i = 0; i < l->depth; i++)
        for (int64_t j = 0; j < r->depth; j++)
            if (l->seqHeaders[i] == r->seqHeaders[j]) { common++; break; }
    return common >= hmm->parameters->minReadCoverageToSupportPhasingBetweenHeterozygousSites;
}
orc_hmm **orc_hmm_splitWherePhasingIsUncertain(orc_hmm *hmm, int64_t *nOut) { /* hmm.c:1322-1383 */
    fb_and_notify(hmm);
    int64_t pathLength;
    orc_cell **path = orc_hmm_forwardTraceBack(hmm, &pathLength);
    orc_genome_fragment *gF = orc_genome_fragment_construct(hmm, path, pathLength);
    pvec hets = {0};
    for (uint64_t i = 0; i < gF->length; i++)
        if (gF->haplotypeString1[i] != gF->haplotypeString2[i]) pvec_push(&hets, (void *) (intptr_t) (gF->refStart + i));
    pvec out = {0};
    for (int64_t i = 0; i + 1 < hets.n; i++) {
        int64_t j = (int64_t) (intptr_t) hets.a[i], k = (int64_t) (intptr_t) hets.a[i + 1];
        if (!sites_linkage_well_supported(hmm, j, k)) {
            int64_t splitPoint = j + (k - j + 1) / 2;
            orc_hmm *right = orc_hmm_split(hmm, splitPoint);
            pvec_push(&out, hmm);
            hmm = right;
        }
    }
    pvec_push(&out, hmm);
    pvec_free(&hets); free(path); orc_genome_fragment_destroy(gF);
    *nOut = out.n;
    return (orc_hmm **) out.a;
}

/* ------------------------------------------------------------------------------------------ */
/* coordination.c                                                                              */
/* ------------------------------------------------------------------------------------------ */
/* sorted "set" of hmms ordered by orc_hmm_cmp (stSortedSet with stRPHmm_cmpFn) */
static int hmm_cmp_qsort(const void *a, const void *b) {
    return orc_hmm_cmp(*(orc_hmm *const *) a, *(orc_hmm *const *) b);
}
/* getTilingPaths coordination.c:186-222 (+ getNextClosestNonoverlappingHmm :19-55).
 * in: array of hmms (consumed); out: pvec of pvec* tiling paths */
static pvec tiling_paths_from(orc_hmm **hmms, int64_t n) {
    qsort(hmms, (size_t) n, sizeof(*hmms), hmm_cmp_qsort);
    uint8_t *used = xcalloc((size_t) (n ? n : 1), 1);
    pvec paths = {0};
    int64_t remaining = n, first = 0;
    while (remaining > 0) {
        pvec *tp = xcalloc(1, sizeof(*tp));
        pvec_push(&paths, tp);
        while (used[first]) first++;
        int64_t cur = first;
        pvec_push(tp, hmms[cur]); used[cur] = 1; remaining--;
        while (1) {
            /* next closest non-overlapping hmm after cur in sort order among the unused */
            orc_hmm *h1 = hmms[cur];
            int64_t nxt = -1;
            for (int64_t j = cur + 1; j < n; j++) {
                if (used[j]) continue;
                orc_hmm *h2 = hmms[j];
                if (strcmp(h1->ref->name, h2->ref->name) != 0) { nxt = j; break; }
                if (h1->refStart + h1->refLength <= h2->refStart) { nxt = j; break; }
            }
            if (nxt < 0) break;
            pvec_push(tp, hmms[nxt]); used[nxt] = 1; remaining--;
            cur = nxt;
        }
    }
    free(used);
    return paths;
}
static pvec tiling_paths2(orc_profile_seq **seqs, int64_t n, const orc_params *params) { /* coordination.c:224-242 */
    orc_hmm **hmms = xmalloc(sizeof(*hmms) * (size_t) (n ? n : 1));
    for (int64_t i = 0; i < n; i++) hmms[i] = orc_hmm_construct(seqs[i], params);
    pvec paths = tiling_paths_from(hmms, n);
    free(hmms);
    return paths;
}
int64_t orc_tilingPathCount(orc_profile_seq **seqs, int64_t n, const orc_params *params) {
    pvec paths = tiling_paths2(seqs, n, params);
    int64_t c = paths.n;
    for (int64_t i = 0; i < paths.n; i++) {
        pvec *tp = paths.a[i];
        for (int64_t j = 0; j < tp->n; j++) orc_hmm_destruct(tp->a[j], 1);
        pvec_free(tp); free(tp);
    }
    pvec_free(&paths);
    return c;
}
static orc_hmm *fuse_tiling_path(pvec *tp) { /* coordination.c:244-261 */
    orc_hmm *right = pvec_pop(tp);
    while (tp->n > 0) {
        orc_hmm *left = pvec_pop(tp);
        right = orc_hmm_fuse(left, right);
        if (!right) break;
    }
    pvec_free(tp); free(tp);
    return right;
}

/* getOverlappingComponents coordination.c:69-184.  Components are kept in creation order. */
typedef struct { pvec members; } component;
static pvec overlapping_components(pvec *tp1, pvec *tp2) {
    pvec comps = {0};
    umap compOf; umap_init(&compOf, tp1->n + tp2->n + 1);
#define COMP_OF(h) ((component *) umap_get(&compOf, (uint64_t) (uintptr_t) (h)))
#define MAKE_COMP(h) ({ component *c_ = xcalloc(1, sizeof(component)); pvec_push(&c_->members, (h)); \
                        pvec_push(&comps, c_); umap_put(&compOf, (uint64_t) (uintptr_t) (h), c_); c_; })
    int64_t j = 0;
    for (int64_t i = 0; i < tp1->n; i++) {
        orc_hmm *h1 = tp1->a[i];
        component *comp = NULL;
        int64_t k = 0;
        while (j + k < tp2->n) {
            orc_hmm *h2 = tp2->a[j + k];
            if (orc_hmm_overlapOnReference(h1, h2)) {
                k++;
                if (comp == NULL) {
                    comp = COMP_OF(h2);
                    if (comp == NULL) comp = MAKE_COMP(h2);
                    pvec_push(&comp->members, h1);
                    umap_put(&compOf, (uint64_t) (uintptr_t) h1, comp);
                } else {
                    pvec_push(&comp->members, h2);
                    umap_put(&compOf, (uint64_t) (uintptr_t) h2, comp);
                }
            } else {
                if (orc_hmm_cmp(h1, h2) < 0) {
                    if (comp == NULL) comp = MAKE_COMP(h1);
                    break;
                } else {
                    if (COMP_OF(h2) == NULL) MAKE_COMP(h2);
                    j++;
                }
            }
        }
        if (comp == NULL) MAKE_COMP(h1);
    }
    while (j < tp2->n) {
        orc_hmm *h2 = tp2->a[j++];
        if (COMP_OF(h2) == NULL) MAKE_COMP(h2);
    }
#undef COMP_OF
#undef MAKE_COMP
    umap_free(&compOf);
    return comps;
}

static pvec *merge_two_tiling_paths(pvec *tp1, pvec *tp2) { /* coordination.c:263-339 */
    pvec comps = overlapping_components(tp1, tp2);
    pvec_free(tp1); free(tp1); pvec_free(tp2); free(tp2);
    pvec *out = xcalloc(1, sizeof(*out));
    pvec crossed = {0}; /* batched seam only: the cross products of this call, swept together below */
    for (int64_t i = 0; i < comps.n; i++) {
        component *comp = comps.a[i];
        pvec sub = tiling_paths_from((orc_hmm **) comp->members.a, comp->members.n);
        orc_hmm *hmm = NULL;
        if (sub.n == 2) {
            orc_hmm *h1 = fuse_tiling_path(sub.a[0]);
            orc_hmm *h2 = fuse_tiling_path(sub.a[1]);
            if (h1 && h2) {
                orc_hmm_alignColumns(h1, h2);
                hmm = orc_hmm_createCrossProductOfTwoAlignedHmm(h1, h2);
                orc_hmm_destruct(h1, 1);
                orc_hmm_destruct(h2, 1);
                if (hmm && g_fb_many_override) pvec_push(&crossed, hmm);
                else if (hmm) {
                    fb_and_notify(hmm);   /* coordination.c:312 */
                    orc_hmm_prune(hmm);   /* coordination.c:313 */
                }
            }
        } else if (sub.n == 1) {
            pvec *only = sub.a[0];
            hmm = pvec_pop(only);
            pvec_free(only); free(only);
        } else {
            orc_fail("component with %lld tiling paths", (long long) sub.n);
        }
        if (hmm) pvec_push(out, hmm);
        pvec_free(&sub);
        pvec_free(&comp->members); free(comp);
    }
    pvec_free(&comps);
    if (crossed.n > 0) { /* the components are independent (coordination.c:285-328): sweep all, then prune each */
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        g_fb_many_override((orc_hmm **) crossed.a, crossed.n);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        g_fb_seconds += (double) (t1.tv_sec - t0.tv_sec) + 1e-9 * (double) (t1.tv_nsec - t0.tv_nsec);
        g_fb_calls += crossed.n;
        for (int64_t i = 0; i < crossed.n; i++) {
            if (g_observer) g_observer(crossed.a[i], g_observer_user);
            orc_hmm_prune(crossed.a[i]);
        }
    }
    pvec_free(&crossed);
    qsort(out->a, (size_t) out->n, sizeof(void *), hmm_cmp_qsort); /* :336 */
    return out;
}
static pvec *merge_tiling_paths(pvec *paths /* of pvec*; consumed */) { /* coordination.c:341-409 */
    if (paths->n == 0) { pvec_free(paths); free(paths); return xcalloc(1, sizeof(pvec)); }
    if (paths->n == 1) { pvec *tp = paths->a[0]; pvec_free(paths); free(paths); return tp; }
    pvec *tp1, *tp2;
    if (paths->n > 2) {
        pvec *half1 = xcalloc(1, sizeof(pvec)), *half2 = xcalloc(1, sizeof(pvec));
        for (int64_t i = 0; i < paths->n / 2; i++) pvec_push(half1, paths->a[i]);
        for (int64_t i = paths->n / 2; i < paths->n; i++) pvec_push(half2, paths->a[i]);
        tp1 = merge_tiling_paths(half1);
        tp2 = merge_tiling_paths(half2);
    } else {
        tp1 = paths->a[0]; tp2 = paths->a[1];
    }
    pvec_free(paths); free(paths);
    return merge_two_tiling_paths(tp1, tp2);
}
orc_hmm **orc_getRPHmms(orc_profile_seq **seqs, int64_t n, const orc_params *params, int64_t *nOut) { /* coordination.c:490-516 */
    pvec paths = tiling_paths2(seqs, n, params);
    if (paths.n > ORC_MAX_READ_PARTITIONING_DEPTH || paths.n > params->maxCoverageDepth) {
        orc_fail("Coverage depth: read depth of %lld exceeds hard maximum of %d with configured maximum of %lld",
                 (long long) paths.n, ORC_MAX_READ_PARTITIONING_DEPTH, (long long) params->maxCoverageDepth);
    }
    pvec *heap = xcalloc(1, sizeof(pvec));
    *heap = paths;
    pvec *final = merge_tiling_paths(heap);
    *nOut = final->n;
    orc_hmm **out = (orc_hmm **) final->a;
    if (out == NULL) out = xmalloc(sizeof(*out));
    free(final);
    return out;
}

typedef struct { pvec *tp; int64_t size; } sized_path;
void orc_filterReadsByCoverageDepth(orc_profile_seq **seqs, int64_t n, const orc_params *params,
                                    orc_profile_seq **filtered, int64_t *nFiltered, orc_profile_seq **discarded,
                                    int64_t *nDiscarded) { /* coordination.c:443-488 */
    pvec paths = tiling_paths2(seqs, n, params);
    keyed *a = xmalloc(sizeof(keyed) * (size_t) (paths.n ? paths.n : 1));
    for (int64_t i = 0; i < paths.n; i++) {
        pvec *tp = paths.a[i];
        int64_t total = 0; /* tilingPathSize :422-434 */
        for (int64_t j = 0; j < tp->n; j++) total += (int64_t) ((orc_hmm *) tp->a[j])->profileSeqs[0]->length;
        a[i].p = tp; a[i].key = (double) total;
    }
    keyed_sort_desc(a, paths.n); /* tilingPathsCmpFn :436-441: longer first */
    int64_t np = paths.n, nf = 0, nd = 0;
    while (np > params->maxCoverageDepth) {
        pvec *tp = a[--np].p;
        while (tp->n > 0) { /* getProfileSeqs :411-420 */
            orc_hmm *h = pvec_pop(tp);
            discarded[nd++] = h->profileSeqs[0];
            orc_hmm_destruct(h, 1);
        }
        pvec_free(tp); free(tp);
    }
    while (np > 0) {
        pvec *tp = a[--np].p;
        while (tp->n > 0) {
            orc_hmm *h = pvec_pop(tp);
            filtered[nf++] = h->profileSeqs[0];
            orc_hmm_destruct(h, 1);
        }
        pvec_free(tp); free(tp);
    }
    *nFiltered = nf; *nDiscarded = nd;
    free(a); pvec_free(&paths);
}

/* ------------------------------------------------------------------------------------------ */
/* emissions.c:246-343 + genomeFragment.c                                                      */
/* ------------------------------------------------------------------------------------------ */
static uint64_t ml_allele(orc_site *site, uint64_t *alleleLogProbs, uint64_t ancestor) { /* emissions.c:246-261 */
    uint64_t maxAllele = 0;
    uint64_t maxProb = alleleLogProbs[0] + *site_sub(site, (int64_t) ancestor, 0);
    for (uint64_t i = 1; i < site->alleleNumber; i++) {
        uint64_t h = alleleLogProbs[i] + *site_sub(site, (int64_t) ancestor, (int64_t) i);
        if (h < maxProb) { maxProb = h; maxAllele = i; }
    }
    return maxAllele;
}
static void fill_position(orc_genome_fragment *gF, uint64_t siteIndex, uint64_t partition, orc_column *column,
                          uint64_t *bcv) { /* emissions.c:263-321 */
    orc_site *site = &gF->reference->sites[siteIndex];
    uint64_t firstAllele = gF->reference->sites[column->refStart].alleleOffset;
    uint64_t siteOffset = site->alleleOffset - firstAllele;
    uint64_t A = site->alleleNumber;
    uint64_t h1[A], h2[A], a1[A], a2[A];
    allele_hap_probs(site, (uint64_t) column->depth, siteOffset, partition, bcv, h1);
    allele_hap_probs(site, (uint64_t) column->depth, siteOffset, ~partition, bcv, h2);
    ancestor_hap_probs(site, h1, a1);
    ancestor_hap_probs(site, h2, a2);
    uint64_t best = a1[0] + a2[0] + site->allelePriorLogProbs[0];
    uint64_t anc = 0;
    for (uint64_t i = 1; i < A; i++) {
        uint64_t j = a1[i] + a2[i] + site->allelePriorLogProbs[i];
        if (j < best) { best = j; anc = i; }
    }
    uint64_t hap1 = ml_allele(site, h1, anc), hap2 = ml_allele(site, h2, anc);
    uint64_t k = siteIndex - gF->refStart;
    gF->ancestorString[k] = anc;
    gF->haplotypeString1[k] = hap1;
    gF->haplotypeString2[k] = hap2;
    gF->genotypeString[k] = hap1 < hap2 ? hap1 * A + hap2 : hap2 * A + hap1;
    gF->genotypeProbs[k] = -((float) best);
    gF->haplotypeProbs1[k] = -(float) h1[hap1];
    gF->haplotypeProbs2[k] = -(float) h2[hap2];
    gF->readsSupportingHaplotype1[k] = (uint64_t) orc_popcount64(partition);
    gF->readsSupportingHaplotype2[k] = (uint64_t) column->depth - (uint64_t) orc_popcount64(partition);
}
static void fill_in_predicted_genome(orc_genome_fragment *gF, uint64_t partition, orc_column *column) { /* emissions.c:323-343 */
    uint64_t *bcv = orc_calculateCountBitVectors(column->seqs, gF->reference, (uint64_t) column->refStart,
                                                 (uint64_t) column->length, (uint64_t) column->depth);
    for (int64_t i = 0; i < column->length; i++)
        fill_position(gF, (uint64_t) (i + column->refStart), partition, column, bcv);
    free(bcv);
}
static orc_genome_fragment *gf_empty(orc_reference *ref, uint64_t refStart, uint64_t length) { /* genomeFragment.c:9-38 */
    orc_genome_fragment *gF = xcalloc(1, sizeof(*gF));
    gF->reference = ref; gF->refStart = refStart; gF->length = length;
    gF->genotypeString = xcalloc(length, sizeof(uint64_t));
    gF->genotypeProbs = xcalloc(length, sizeof(float));
    gF->haplotypeProbs1 = xcalloc(length, sizeof(float));
    gF->haplotypeProbs2 = xcalloc(length, sizeof(float));
    gF->haplotypeString1 = xcalloc(length, sizeof(uint64_t));
    gF->haplotypeString2 = xcalloc(length, sizeof(uint64_t));
    gF->ancestorString = xcalloc(length, sizeof(uint64_t));
    gF->readsSupportingHaplotype1 = xcalloc(length, sizeof(uint64_t));
    gF->readsSupportingHaplotype2 = xcalloc(length, sizeof(uint64_t));
    return gF;
}
/* stRPHmm_partitionSequencesByStatePath hmm.c:221-248; set semantics, insertion ordered */
static int64_t *partition_seqs_by_path(orc_hmm *hmm, orc_cell **path, int64_t pathLength, int partition1,
                                       int64_t *nOut) {
    int64_t *ids = xmalloc(sizeof(int64_t) * (size_t) (hmm->nProfileSeqs + 1));
    int64_t n = 0;
    umap seen; umap_init(&seen, hmm->nProfileSeqs + 1);
    orc_column *column = hmm->firstColumn;
    for (int64_t i = 0; i < pathLength; i++) {
        orc_cell *cell = path[i];
        for (int64_t j = 0; j < column->depth; j++) {
            int in1 = orc_seqInHap1(cell->partition, j);
            if ((in1 && partition1) || (!in1 && !partition1)) {
                orc_profile_seq *s = column->seqHeaders[j];
                if (umap_get(&seen, (uint64_t) (uintptr_t) s) == NULL) {
                    umap_put(&seen, (uint64_t) (uintptr_t) s, s);
                    ids[n++] = s->id;
                }
            }
        }
        if (column->nColumn != NULL) column = column->nColumn->nColumn;
    }
    umap_free(&seen);
    *nOut = n;
    return ids;
}
orc_genome_fragment *orc_genome_fragment_construct(orc_hmm *hmm, orc_cell **path, int64_t pathLength) { /* genomeFragment.c:40-69 */
    orc_genome_fragment *gF = gf_empty(hmm->ref, (uint64_t) hmm->refStart, (uint64_t) hmm->refLength);
    gF->reads1 = partition_seqs_by_path(hmm, path, pathLength, 1, &gF->nReads1);
    gF->reads2 = partition_seqs_by_path(hmm, path, pathLength, 0, &gF->nReads2);
    /* room for later growth (refinement moves reads; discarded reads are appended) */
    orc_column *column = hmm->firstColumn;
    for (int64_t i = 0; i < pathLength - 1; i++) {
        fill_in_predicted_genome(gF, path[i]->partition, column);
        column = column->nColumn->nColumn;
    }
    fill_in_predicted_genome(gF, path[pathLength - 1]->partition, column);
    return gF;
}
void orc_genome_fragment_destroy(orc_genome_fragment *gF) { /* genomeFragment.c:278-300 */
    if (!gF) return;
    free(gF->genotypeString); free(gF->genotypeProbs); free(gF->haplotypeProbs1); free(gF->haplotypeProbs2);
    free(gF->haplotypeString1); free(gF->haplotypeString2); free(gF->ancestorString);
    free(gF->readsSupportingHaplotype1); free(gF->readsSupportingHaplotype2);
    free(gF->reads1); free(gF->reads2);
    free(gF);
}
double orc_getLogProbOfReadGivenHaplotype(const uint64_t *hap, int64_t start, int64_t length, orc_profile_seq *seq,
                                          orc_reference *ref) { /* genomeFragment.c:71-89 */
    double total = 0.0;
    uint64_t firstAllele = ref->sites[seq->refStart].alleleOffset;
    for (int64_t i = 0; i < (int64_t) seq->length; i++) {
        int64_t j = i + (int64_t) seq->refStart - start;
        if (j >= 0 && j < length) {
            uint64_t allele = hap[j];
            orc_site *site = &ref->sites[i + (int64_t) seq->refStart];
            total -= seq->profileProbs[site->alleleOffset - firstAllele + allele];
        }
    }
    return total / ORC_PROFILE_PROB_SCALAR;
}
static orc_profile_seq *find_seq(orc_hmm *hmm, int64_t id) {
    for (int64_t i = 0; i < hmm->nProfileSeqs; i++) if (hmm->profileSeqs[i]->id == id) return hmm->profileSeqs[i];
    return NULL;
}
void orc_genome_fragment_refine(orc_genome_fragment *gF, orc_hmm *hmm, orc_cell **path, int64_t pathLength,
                                int64_t maxIterations) { /* genomeFragment.c:165-232 */
    uint64_t *p = xmalloc(sizeof(uint64_t) * (size_t) pathLength);
    for (int64_t i = 0; i < pathLength; i++) p[i] = path[i]->partition;
    int64_t total = gF->nReads1 + gF->nReads2;
    gF->reads1 = realloc(gF->reads1, sizeof(int64_t) * (size_t) (total + 1));
    gF->reads2 = realloc(gF->reads2, sizeof(int64_t) * (size_t) (total + 1));
    int64_t iteration = 0;
    while (iteration++ < maxIterations) {
        /* findReadsThatWereMoreProbablyGeneratedByTheOtherHaplotype :126-151 */
        umap move12, move21; umap_init(&move12, total + 1); umap_init(&move21, total + 1);
        int64_t n12 = 0, n21 = 0;
        for (int64_t i = 0; i < gF->nReads1; i++) {
            orc_profile_seq *s = find_seq(hmm, gF->reads1[i]);
            double a = orc_getLogProbOfReadGivenHaplotype(gF->haplotypeString1, (int64_t) gF->refStart, (int64_t) gF->length, s, gF->reference);
            double b = orc_getLogProbOfReadGivenHaplotype(gF->haplotypeString2, (int64_t) gF->refStart, (int64_t) gF->length, s, gF->reference);
            if (a < b) { umap_put(&move12, (uint64_t) (uintptr_t) s, s); n12++; }
        }
        for (int64_t i = 0; i < gF->nReads2; i++) {
            orc_profile_seq *s = find_seq(hmm, gF->reads2[i]);
            double a = orc_getLogProbOfReadGivenHaplotype(gF->haplotypeString2, (int64_t) gF->refStart, (int64_t) gF->length, s, gF->reference);
            double b = orc_getLogProbOfReadGivenHaplotype(gF->haplotypeString1, (int64_t) gF->refStart, (int64_t) gF->length, s, gF->reference);
            if (a < b) { umap_put(&move21, (uint64_t) (uintptr_t) s, s); n21++; }
        }
        if (n12 + n21 == 0) { umap_free(&move12); umap_free(&move21); break; }
        /* update read sets (:203-207): remove movers, then append the incoming ones */
        int64_t *new1 = xmalloc(sizeof(int64_t) * (size_t) (total + 1)), *new2 = xmalloc(sizeof(int64_t) * (size_t) (total + 1));
        int64_t c1 = 0, c2 = 0;
        for (int64_t i = 0; i < gF->nReads1; i++)
            if (!umap_get(&move12, (uint64_t) (uintptr_t) find_seq(hmm, gF->reads1[i]))) new1[c1++] = gF->reads1[i];
        for (int64_t i = 0; i < gF->nReads2; i++)
            if (!umap_get(&move21, (uint64_t) (uintptr_t) find_seq(hmm, gF->reads2[i]))) new2[c2++] = gF->reads2[i];
        for (int64_t i = 0; i < gF->nReads2; i++)
            if (umap_get(&move21, (uint64_t) (uintptr_t) find_seq(hmm, gF->reads2[i]))) new1[c1++] = gF->reads2[i];
        for (int64_t i = 0; i < gF->nReads1; i++)
            if (umap_get(&move12, (uint64_t) (uintptr_t) find_seq(hmm, gF->reads1[i]))) new2[c2++] = gF->reads1[i];
        free(gF->reads1); free(gF->reads2);
        gF->reads1 = new1; gF->reads2 = new2; gF->nReads1 = c1; gF->nReads2 = c2;
        /* update path + genome fragment (:211-226) */
        orc_column *column = hmm->firstColumn;
        for (int64_t i = 0; i < pathLength; i++) {
            for (int64_t r = 0; r < column->depth; r++) { /* flipReadsBetweenPartitions :153-163, both sets */
                orc_profile_seq *s = column->seqHeaders[r];
                if (umap_get(&move12, (uint64_t) (uintptr_t) s)) p[i] = orc_flipAReadsPartition(p[i], (uint64_t) r);
            }
            for (int64_t r = 0; r < column->depth; r++) {
                orc_profile_seq *s = column->seqHeaders[r];
                if (umap_get(&move21, (uint64_t) (uintptr_t) s)) p[i] = orc_flipAReadsPartition(p[i], (uint64_t) r);
            }
            fill_in_predicted_genome(gF, p[i], column);
            if (i + 1 < pathLength) column = column->nColumn->nColumn;
        }
        umap_free(&move12); umap_free(&move21);
    }
    free(p);
}

/* ------------------------------------------------------------------------------------------ */
/* bubbleGraph.c:2673-2801 phasing driver                                                      */
/* ------------------------------------------------------------------------------------------ */
orc_genome_fragment *orc_phase_profile_seqs(orc_profile_seq **seqs, const uint8_t *strands, int64_t n,
                                            const orc_params *params, orc_hmm **finalHmm) {
    if (finalHmm) *finalHmm = NULL;
    if (n == 0) return gf_empty(NULL, 0, 0); /* :2719-2728 */
    /* filterReadsByCoverageDepth2 (bubbleGraph.c:2651-2671) -> coordination.c:443 */
    orc_profile_seq **filtered = xmalloc(sizeof(*filtered) * (size_t) n), **discarded = xmalloc(sizeof(*discarded) * (size_t) n);
    int64_t nf, nd;
    orc_filterReadsByCoverageDepth(seqs, n, params, filtered, &nf, discarded, &nd);
    umap disc; umap_init(&disc, nd + 1);
    for (int64_t i = 0; i < nd; i++) umap_put(&disc, (uint64_t) (uintptr_t) discarded[i], discarded[i]);
    /* strand split in read order (:2705-2716) */
    orc_profile_seq **fwd = xmalloc(sizeof(*fwd) * (size_t) n), **rev = xmalloc(sizeof(*rev) * (size_t) n);
    int64_t nfwd = 0, nrev = 0;
    for (int64_t i = 0; i < n; i++) {
        if (umap_get(&disc, (uint64_t) (uintptr_t) seqs[i])) continue;
        if (strands[i]) fwd[nfwd++] = seqs[i]; else rev[nrev++] = seqs[i];
    }
    orc_params *pc = xmalloc(sizeof(*pc)); /* stRPHmmParameters_copy :2732 */
    *pc = *params;
    pc->includeAncestorSubProb = 0; /* :2733 */
    int64_t nF, nR;
    orc_hmm **tpF = orc_getRPHmms(fwd, nfwd, pc, &nF); /* :2736 */
    orc_hmm **tpR = orc_getRPHmms(rev, nrev, pc, &nR); /* :2740 */
    pvec *a = xcalloc(1, sizeof(pvec)), *b = xcalloc(1, sizeof(pvec));
    for (int64_t i = 0; i < nF; i++) pvec_push(a, tpF[i]);
    for (int64_t i = 0; i < nR; i++) pvec_push(b, tpR[i]);
    free(tpF); free(tpR);
    orc_genome_fragment *gF = NULL;
    pvec *joined = merge_two_tiling_paths(a, b); /* :2745 */
    if (joined->n == 0 || g_err[0]) {
        pvec_free(joined); free(joined);
        gF = gf_empty(seqs[0]->ref, 0, 0);
        goto done;
    }
    orc_hmm *hmm = fuse_tiling_path(joined);
    pc->includeAncestorSubProb = 1; /* :2748 */
    fb_and_notify(hmm);             /* :2749 */
    int64_t pathLength;
    orc_cell **path = orc_hmm_forwardTraceBack(hmm, &pathLength); /* :2755 */
    gF = orc_genome_fragment_construct(hmm, path, pathLength);      /* :2761 */
    orc_genome_fragment_refine(gF, hmm, path, pathLength, params->roundsOfIterativeRefinement); /* :2764 */
    /* re-add coverage-filtered reads (:2772-2779); stSet iteration order = discard order here */
    gF->reads1 = realloc(gF->reads1, sizeof(int64_t) * (size_t) (gF->nReads1 + nd + 1));
    gF->reads2 = realloc(gF->reads2, sizeof(int64_t) * (size_t) (gF->nReads2 + nd + 1));
    for (int64_t i = 0; i < nd; i++) {
        double x = orc_getLogProbOfReadGivenHaplotype(gF->haplotypeString1, (int64_t) gF->refStart, (int64_t) gF->length, discarded[i], gF->reference);
        double y = orc_getLogProbOfReadGivenHaplotype(gF->haplotypeString2, (int64_t) gF->refStart, (int64_t) gF->length, discarded[i], gF->reference);
        if (x < y) gF->reads2[gF->nReads2++] = discarded[i]->id; else gF->reads1[gF->nReads1++] = discarded[i]->id;
    }
    free(path);
    if (finalHmm) { hmm->parameters = params; *finalHmm = hmm; } else orc_hmm_destruct(hmm, 1);
done:
    umap_free(&disc); free(filtered); free(discarded); free(fwd); free(rev); free(pc);
    return gF;
}

/* ------------------------------------------------------------------------------------------ */
/* flattening into the mrp_hmm_job arrays                                                      */
/* ------------------------------------------------------------------------------------------ */
void orc_hmm_flat_sizes(orc_hmm *hmm, int64_t sizes[4]) {
    int64_t K = 0, C = 0, M = 0, D = 0;
    orc_column *column = hmm->firstColumn;
    while (1) {
        K++; D += column->depth;
        for (orc_cell *c = column->head; c; c = c->nCell) C++;
        if (column->nColumn == NULL) break;
        M += column->nColumn->cells.n;
        column = column->nColumn->nColumn;
    }
    sizes[0] = K; sizes[1] = C; sizes[2] = M; sizes[3] = D;
}
void orc_hmm_flatten(orc_hmm *hmm, int32_t *colRefStart, int32_t *colLength, int32_t *colDepth, int64_t *colCellOff,
                     int64_t *colReadOff, int64_t *readByteOff, int64_t *readIds, uint64_t *partition,
                     uint64_t *maskFrom, uint64_t *maskTo, int64_t *mcolCellOff, uint64_t *mergeFrom,
                     uint64_t *mergeTo, uint32_t *cellNext, uint32_t *cellPrev, double *cellF, double *cellB,
                     double *mergeF, double *mergeB, double *colTotal) {
    int64_t k = 0, c = 0, m = 0, d = 0;
    orc_column *column = hmm->firstColumn;
    orc_merge_column *prevM = NULL;
    umap prevIdx = {0};
    colCellOff[0] = 0; colReadOff[0] = 0; mcolCellOff[0] = 0;
    while (1) {
        colRefStart[k] = (int32_t) column->refStart; colLength[k] = (int32_t) column->length;
        colDepth[k] = (int32_t) column->depth; colTotal[k] = column->totalLogProb;
        for (int64_t i = 0; i < column->depth; i++) {
            readIds[d] = column->seqHeaders[i]->id;
            readByteOff[d] = (int64_t) (column->seqs[i] - column->seqHeaders[i]->profileProbs);
            d++;
        }
        orc_merge_column *nextM = column->nColumn;
        umap nextIdx = {0};
        if (nextM) {
            umap_init(&nextIdx, nextM->cells.n + 1);
            maskFrom[k] = nextM->maskFrom; maskTo[k] = nextM->maskTo;
            for (int64_t i = 0; i < nextM->cells.n; i++) {
                orc_merge_cell *mc = nextM->cells.a[i];
                mergeFrom[m] = mc->fromPartition; mergeTo[m] = mc->toPartition;
                mergeF[m] = mc->forwardLogProb; mergeB[m] = mc->backwardLogProb;
                umap_put(&nextIdx, (uint64_t) (uintptr_t) mc, (void *) (uintptr_t) (i + 1));
                m++;
            }
            mcolCellOff[k + 1] = m;
        }
        for (orc_cell *cell = column->head; cell; cell = cell->nCell) {
            partition[c] = cell->partition; cellF[c] = cell->forwardLogProb; cellB[c] = cell->backwardLogProb;
            cellNext[c] = 0xFFFFFFFFu; cellPrev[c] = 0xFFFFFFFFu;
            if (nextM) {
                orc_merge_cell *mc = orc_mcol_getNextMergeCell(cell, nextM);
                if (mc) cellNext[c] = (uint32_t) ((uintptr_t) umap_get(&nextIdx, (uint64_t) (uintptr_t) mc) - 1);
            }
            if (prevM) {
                orc_merge_cell *mc = orc_mcol_getPreviousMergeCell(cell, prevM);
                if (mc) cellPrev[c] = (uint32_t) ((uintptr_t) umap_get(&prevIdx, (uint64_t) (uintptr_t) mc) - 1);
            }
            c++;
        }
        k++;
        colCellOff[k] = c; colReadOff[k] = d;
        umap_free(&prevIdx);
        prevIdx = nextIdx; prevM = nextM;
        if (nextM == NULL) break;
        column = nextM->nColumn;
    }
    umap_free(&prevIdx);
}
