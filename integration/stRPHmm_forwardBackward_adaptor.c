/*
 * stRPHmm_forwardBackward_adaptor.c -- reference-side code: routes margin's forward/backward through libmargin_rphmm.so.
 *
 * This file is meant to be compiled INTO margin (UCSC-nanopore-cgl/margin), next to impl/hmm.c, with the body of
 *     void stRPHmm_forwardBackward(stRPHmm *hmm)              inc/margin.h:369, impl/hmm.c:931-942
 * removed from impl/hmm.c (stRPHmm_initialiseProbs / stRPHmm_forward / stRPHmm_backward become unused).  Everything else of
 * margin links unchanged: the callers impl/coordination.c:312, impl/bubbleGraph.c:2749 and impl/hmm.c:1332 see the same
 * function with the same post-conditions:
 *     every stRPCell.forwardLogProb / backwardLogProb (the latter excluding the cell's own emission, hmm.c:881-892),
 *     every stRPMergeCell.forwardLogProb / backwardLogProb (unreachable ones keep ST_MATH_LOG_ZERO = -inf),
 *     every stRPColumn.totalLogProb, stRPHmm.forwardLogProb / backwardLogProb;
 *     cell list order, both merge cell hashes and all links untouched; nothing of the graph is retained after return.
 *
 * What it does: flatten (cells in LIST order, merge cells in the order stHash_getValues returns them during this call)
 * -> mrp_fb_run (include/margin_rphmm.h; the library resolves cell -> merge cell transitions from partition & mask as
 * mergeColumn.c:63-79 does) -> scatter back.  The profile bytes of a chunk's reads and its site tables are uploaded
 * once per chunk when the caller registers them (two lines in bubbleGraph_phaseBubbleGraph); an
 * unregistered hmm is served through a transient chunk built from the hmm's own reads.
 *
 * stRPHmm_forwardBackwardMany sweeps several independent hmms in ONE device batch: what mergeTwoTilingPaths
 * (coordination.c:285-328) should issue for the cross products of all overlap components of a call.
 *
 * Threading: one context, and one registry of chunks, per host thread (phase.c:276 runs one chunk per OpenMP thread).
 * Errors: the library returns a status; like the code it replaces, the adaptor calls st_errAbort on failure.
 *
 * The test suite of the MI355X build compiles this very file against its CPU oracle's linked-list hmm through a small
 * binding header (MRP_ADAPTOR_BINDING_HEADER; tests/adaptor_binding/) and checks every post-condition bit for bit.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "margin_rphmm.h"

#ifdef MRP_ADAPTOR_BINDING_HEADER
#include MRP_ADAPTOR_BINDING_HEADER
#else
#include "margin.h"
/* how the adaptor reads a merge column (struct _stRPMergeColumn, inc/margin.h:439-445; sonLib's stHash / stList) */
typedef stList *adp_mcells;
#define ADP_MCOL_MASK_FROM(m) ((m)->maskFrom)
#define ADP_MCOL_MASK_TO(m) ((m)->maskTo)
#define ADP_MCOL_NEXT(m) ((m)->nColumn)
#define ADP_MCELLS_GET(m) stHash_getValues((m)->mergeCellsFrom)
#define ADP_MCELLS_LEN(l) stList_length(l)
#define ADP_MCELLS_AT(l, i) ((stRPMergeCell *) stList_get((l), (i)))
#define ADP_MCELLS_FREE(l) stList_destruct(l)
#define ADP_ABORT(...) st_errAbort(__VA_ARGS__)
#define ADP_MALLOC(n) st_malloc(n)
#define ADP_DEVICE_FOR_THIS_THREAD() (omp_get_thread_num() % mrp_device_count())
/* the body this file replaces, kept in impl/hmm.c under this name: hmms below the size threshold stay on it */
void stRPHmm_forwardBackward_cpu(stRPHmm *hmm);
#define ADP_CPU_FORWARD_BACKWARD(hmm) stRPHmm_forwardBackward_cpu(hmm)
#include <omp.h>
#endif

/* An hmm with fewer cells than this is swept by the CPU body: a device sweep costs a flatten, an upload, a launch and a
 * download (~0.3 ms whatever the size), the CPU body ~0.1 us per cell.  Most hmms of a chunk are that small (the first
 * merge levels: a few reads each), most CELLS are in the few large ones.  MRP_ADAPTOR_MIN_CELLS overrides (0: everything on
 * the device); measured with tools/adaptor_probe.py, INTEGRATION.md. */
#ifndef MRP_ADAPTOR_MIN_CELLS_DEFAULT
#define MRP_ADAPTOR_MIN_CELLS_DEFAULT 4096
#endif
static int64_t adpMinCells = -1;
static int64_t adp_min_cells(void) {
    if (adpMinCells < 0) {
        const char *e = getenv("MRP_ADAPTOR_MIN_CELLS");
        adpMinCells = e != NULL ? atoll(e) : MRP_ADAPTOR_MIN_CELLS_DEFAULT;
        if (adpMinCells < 0) adpMinCells = 0;
    }
    return adpMinCells;
}
void mrpAdaptor_setMinCells(int64_t cells) { adpMinCells = cells < 0 ? 0 : cells; } /* overrides the environment / the default */
static int adp_is_small(stRPHmm *hmm, int64_t limit) { /* fewer than `limit` cells? (stops counting at the limit) */
    int64_t n = 0;
    for (stRPColumn *c = hmm->firstColumn;; c = ADP_MCOL_NEXT(c->nColumn)) {
        for (stRPCell *cell = c->head; cell != NULL; cell = cell->nCell)
            if (++n >= limit) return 0;
        if (c->nColumn == NULL) break;
    }
    return 1;
}

/* ---- per-thread state ----------------------------------------------------------------------------------------- */
typedef struct {
    const stReference *ref;
    mrp_chunk *chunk;
    /* open-addressing map stProfileSeq* -> offset of its profileProbs in the chunk's pool */
    const stProfileSeq **keys;
    int64_t *vals;
    uint64_t mask;
    int transient;
} adp_chunk;

static __thread mrp_context *adpCtx = NULL;
static __thread adp_chunk *adpChunks = NULL; /* registered chunks of this thread */
static __thread int64_t adpChunkNo = 0, adpChunkCap = 0;

static mrp_context *adp_context(void) {
    if (adpCtx == NULL) {
        if (mrp_device_count() <= 0) ADP_ABORT("margin_rphmm: no HIP device visible");
        if (mrp_abi_version() != MRP_ABI_VERSION) ADP_ABORT("margin_rphmm: library speaks ABI %d, the adaptor was built against %d", mrp_abi_version(), MRP_ABI_VERSION);
        if (mrp_context_create(ADP_DEVICE_FOR_THIS_THREAD(), &adpCtx) != MRP_OK) ADP_ABORT("margin_rphmm: %s", mrp_last_error());
    }
    return adpCtx;
}

static uint64_t adp_hash_ptr(const void *p) {
    uint64_t x = (uint64_t) (uintptr_t) p;
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
static int64_t adp_pool_offset(const adp_chunk *c, const stProfileSeq *seq) {
    uint64_t i = adp_hash_ptr(seq) & c->mask;
    while (c->keys[i] != NULL) {
        if (c->keys[i] == seq) return c->vals[i];
        i = (i + 1) & c->mask;
    }
    return -1;
}
/* number of profile bytes of a read: the alleles of its sites (profileSeq.c:13-29) */
static int64_t adp_profile_bytes(const stReference *ref, const stProfileSeq *seq) {
    const uint64_t end = seq->refStart + seq->length;
    const uint64_t last = end < ref->length ? ref->sites[end].alleleOffset : ref->totalAlleles;
    return (int64_t) (last - seq->alleleOffset);
}

static void adp_chunk_free(adp_chunk *c) {
    mrp_chunk_destroy(c->chunk);
    free((void *) c->keys);
    free(c->vals);
    memset(c, 0, sizeof(*c));
}

/* stReference (inc/margin.h:164-180) + the profileProbs of the given reads -> one mrp_chunk on this thread's device */
static void adp_chunk_build(adp_chunk *c, const stReference *ref, stProfileSeq *const *seqs, int64_t n) {
    memset(c, 0, sizeof(*c));
    c->ref = ref;
    uint64_t cap = 16;
    while (cap < (uint64_t) n * 2) cap *= 2;
    c->mask = cap - 1;
    c->keys = calloc(cap, sizeof(*c->keys));
    c->vals = calloc(cap, sizeof(*c->vals));
    int64_t poolBytes = 0;
    for (int64_t i = 0; i < n; i++) {
        if (adp_pool_offset(c, seqs[i]) >= 0) continue; /* listed twice */
        uint64_t h = adp_hash_ptr(seqs[i]) & c->mask;
        while (c->keys[h] != NULL) h = (h + 1) & c->mask;
        c->keys[h] = seqs[i];
        c->vals[h] = poolBytes;
        poolBytes += adp_profile_bytes(ref, seqs[i]);
    }
    uint8_t *pool = ADP_MALLOC((size_t) poolBytes + 1);
    for (uint64_t h = 0; h < cap; h++)
        if (c->keys[h] != NULL) memcpy(pool + c->vals[h], c->keys[h]->profileProbs, (size_t) adp_profile_bytes(ref, c->keys[h]));
    uint32_t *alleleNumber = ADP_MALLOC(sizeof(uint32_t) * (size_t) (ref->length + 1));
    uint64_t subNo = 0;
    for (uint64_t s = 0; s < ref->length; s++) {
        alleleNumber[s] = (uint32_t) ref->sites[s].alleleNumber;
        subNo += ref->sites[s].alleleNumber * ref->sites[s].alleleNumber;
    }
    uint16_t *sub = ADP_MALLOC(sizeof(uint16_t) * (size_t) (subNo + 1));
    uint16_t *prior = ADP_MALLOC(sizeof(uint16_t) * (size_t) (ref->totalAlleles + 1));
    uint64_t so = 0, po = 0;
    for (uint64_t s = 0; s < ref->length; s++) { /* site-major, [from * A + to] (emissions.c:13-19) */
        const uint64_t A = ref->sites[s].alleleNumber;
        memcpy(sub + so, ref->sites[s].substitutionLogProbs, sizeof(uint16_t) * (size_t) (A * A));
        memcpy(prior + po, ref->sites[s].allelePriorLogProbs, sizeof(uint16_t) * (size_t) A);
        so += A * A; po += A;
    }
    if (mrp_chunk_create(adp_context(), (int64_t) ref->length, alleleNumber, sub, prior, pool, poolBytes, &c->chunk) != MRP_OK)
        ADP_ABORT("margin_rphmm: %s", mrp_last_error());
    free(pool); free(alleleNumber); free(sub); free(prior);
}

/* Call once per chunk before its hmms are swept (bubbleGraph_phaseBubbleGraph, after bubbleGraph_getProfileSeqs,
 * bubbleGraph.c:2687): uploads the site tables and every read's profile bytes once.  seqs = the profile sequences. */
void mrpAdaptor_registerProfileSeqs(stReference *ref, stProfileSeq **seqs, int64_t n) {
    for (int64_t i = 0; i < adpChunkNo; i++)
        if (adpChunks[i].ref == ref) { adp_chunk_free(&adpChunks[i]); adpChunks[i] = adpChunks[--adpChunkNo]; break; }
    if (adpChunkNo == adpChunkCap) {
        adpChunkCap = adpChunkCap ? 2 * adpChunkCap : 4;
        adpChunks = realloc(adpChunks, sizeof(*adpChunks) * (size_t) adpChunkCap);
    }
    adp_chunk_build(&adpChunks[adpChunkNo++], ref, seqs, n);
}
/* and once when the chunk is done (before stReference_destruct / the profile sequences are freed) */
void mrpAdaptor_unregister(stReference *ref) {
    for (int64_t i = 0; i < adpChunkNo; i++)
        if (adpChunks[i].ref == ref) { adp_chunk_free(&adpChunks[i]); adpChunks[i] = adpChunks[--adpChunkNo]; return; }
}
#ifndef MRP_ADAPTOR_BINDING_HEADER
void mrpAdaptor_registerChunk(stReference *ref, stList *profileSeqs) { /* the stList of bubbleGraph.c:2687 / :2699 */
    int64_t n = stList_length(profileSeqs);
    stProfileSeq **seqs = st_malloc(sizeof(*seqs) * (size_t) (n + 1));
    for (int64_t i = 0; i < n; i++) seqs[i] = stList_get(profileSeqs, i);
    mrpAdaptor_registerProfileSeqs(ref, seqs, n);
    free(seqs);
}
#endif

static adp_chunk *adp_chunk_of(const stReference *ref) {
    for (int64_t i = 0; i < adpChunkNo; i++)
        if (adpChunks[i].ref == ref) return &adpChunks[i];
    return NULL;
}

/* ---- one hmm flattened ------------------------------------------------------------------------------------------ */
typedef struct {
    mrp_hmm_job job;
    stRPHmm *hmm;
    adp_mcells *mcells; /* [K - 1] the merge cells of every merge column, in the order they were flattened */
    void *block;        /* one allocation behind every array of the job */
    adp_chunk transient;
    int hasTransient;
} adp_flat;

static void adp_flatten(stRPHmm *hmm, adp_flat *f) {
    memset(f, 0, sizeof(*f));
    f->hmm = hmm;
    const int64_t K = hmm->columnNumber;
    /* sizes: walk columns / cells in list order */
    int64_t nC = 0, nM = 0, nD = 0;
    f->mcells = ADP_MALLOC(sizeof(*f->mcells) * (size_t) (K > 1 ? K - 1 : 1));
    {
        int64_t k = 0;
        for (stRPColumn *c = hmm->firstColumn;; c = ADP_MCOL_NEXT(c->nColumn), k++) {
            for (stRPCell *cell = c->head; cell != NULL; cell = cell->nCell) nC++;
            nD += c->depth;
            if (c->nColumn == NULL) break;
            f->mcells[k] = ADP_MCELLS_GET(c->nColumn);
            nM += ADP_MCELLS_LEN(f->mcells[k]);
        }
        if (k + 1 != K) ADP_ABORT("margin_rphmm adaptor: hmm has %lld columns, columnNumber says %lld", (long long) (k + 1), (long long) K);
    }
    /* the chunk: registered, or built from the reads of this hmm alone */
    adp_chunk *chunk = adp_chunk_of(hmm->ref);
    if (chunk == NULL) {
        stProfileSeq **seqs = ADP_MALLOC(sizeof(*seqs) * (size_t) (nD + 1));
        int64_t n = 0;
        for (stRPColumn *c = hmm->firstColumn;; c = ADP_MCOL_NEXT(c->nColumn)) {
            for (int64_t i = 0; i < c->depth; i++) seqs[n++] = c->seqHeaders[i];
            if (c->nColumn == NULL) break;
        }
        adp_chunk_build(&f->transient, hmm->ref, seqs, n);
        free(seqs);
        f->transient.transient = 1;
        f->hasTransient = 1;
        chunk = &f->transient;
    }
    /* one block for all arrays */
#define AL8(x) (((size_t) (x) + 7) & ~(size_t) 7)
    size_t bytes = 0;
    const size_t oStart = bytes; bytes += AL8(4 * K);
    const size_t oLen = bytes; bytes += AL8(4 * K);
    const size_t oDepth = bytes; bytes += AL8(4 * K);
    const size_t oCellOff = bytes; bytes += AL8(8 * (K + 1));
    const size_t oReadOff = bytes; bytes += AL8(8 * (K + 1));
    const size_t oRbo = bytes; bytes += AL8(8 * (nD + 1));
    const size_t oPart = bytes; bytes += AL8(8 * (nC + 1));
    const size_t oMaskFrom = bytes; bytes += AL8(8 * K);
    const size_t oMaskTo = bytes; bytes += AL8(8 * K);
    const size_t oMcellOff = bytes; bytes += AL8(8 * (K + 1));
    const size_t oMFrom = bytes; bytes += AL8(8 * (nM + 1));
    const size_t oMTo = bytes; bytes += AL8(8 * (nM + 1));
    const size_t oF = bytes; bytes += AL8(8 * (nC + 1));
    const size_t oB = bytes; bytes += AL8(8 * (nC + 1));
    const size_t oMF = bytes; bytes += AL8(8 * (nM + 1));
    const size_t oMB = bytes; bytes += AL8(8 * (nM + 1));
    const size_t oTotal = bytes; bytes += AL8(8 * K);
    const size_t oFB = bytes; bytes += 16;
#undef AL8
    char *blk = ADP_MALLOC(bytes);
    f->block = blk;
    int32_t *colStart = (int32_t *) (blk + oStart), *colLen = (int32_t *) (blk + oLen), *colDepth = (int32_t *) (blk + oDepth);
    int64_t *cellOff = (int64_t *) (blk + oCellOff), *readOff = (int64_t *) (blk + oReadOff), *rbo = (int64_t *) (blk + oRbo);
    uint64_t *part = (uint64_t *) (blk + oPart), *maskFrom = (uint64_t *) (blk + oMaskFrom), *maskTo = (uint64_t *) (blk + oMaskTo);
    int64_t *mcellOff = (int64_t *) (blk + oMcellOff);
    uint64_t *mFrom = (uint64_t *) (blk + oMFrom), *mTo = (uint64_t *) (blk + oMTo);
    int64_t c_ = 0, m_ = 0, d_ = 0, k = 0;
    cellOff[0] = 0; readOff[0] = 0; mcellOff[0] = 0;
    for (stRPColumn *c = hmm->firstColumn;; c = ADP_MCOL_NEXT(c->nColumn), k++) {
        colStart[k] = (int32_t) c->refStart; colLen[k] = (int32_t) c->length; colDepth[k] = (int32_t) c->depth;
        for (int64_t i = 0; i < c->depth; i++) { /* column->seqs[i] points into seqHeaders[i]->profileProbs (hmm.c:121-122, column.c:78-84) */
            const int64_t base = adp_pool_offset(chunk, c->seqHeaders[i]);
            if (base < 0) ADP_ABORT("margin_rphmm adaptor: a read of the hmm was not registered with its chunk");
            rbo[d_++] = base + (int64_t) (c->seqs[i] - c->seqHeaders[i]->profileProbs);
        }
        readOff[k + 1] = d_;
        for (stRPCell *cell = c->head; cell != NULL; cell = cell->nCell) part[c_++] = cell->partition;
        cellOff[k + 1] = c_;
        if (c->nColumn == NULL) break;
        maskFrom[k] = ADP_MCOL_MASK_FROM(c->nColumn); maskTo[k] = ADP_MCOL_MASK_TO(c->nColumn);
        const int64_t n = ADP_MCELLS_LEN(f->mcells[k]);
        for (int64_t i = 0; i < n; i++) {
            const stRPMergeCell *mc = ADP_MCELLS_AT(f->mcells[k], i);
            mFrom[m_] = mc->fromPartition; mTo[m_] = mc->toPartition; m_++;
        }
        mcellOff[k + 1] = m_;
    }
    mrp_hmm_job *j = &f->job;
    j->chunk = chunk->chunk;
    j->n_columns = (int32_t) K;
    j->flags = (hmm->parameters->maxNotSumTransitions ? MRP_FLAG_MAX_NOT_SUM : 0u) |
               (hmm->parameters->includeAncestorSubProb ? MRP_FLAG_INCLUDE_ANCESTOR_SUB_PROB : 0u);
    j->col_ref_start = colStart; j->col_length = colLen; j->col_depth = colDepth;
    j->col_cell_off = cellOff; j->col_read_off = readOff; j->read_byte_off = rbo;
    j->partition = part; j->mask_from = maskFrom; j->mask_to = maskTo;
    j->mcol_cell_off = mcellOff; j->merge_from = mFrom; j->merge_to = mTo;
    j->cell_next = NULL; j->cell_prev = NULL; /* the library performs mergeColumn.c:63-79 from partition & mask */
    j->cell_forward = (double *) (blk + oF); j->cell_backward = (double *) (blk + oB);
    j->merge_forward = (double *) (blk + oMF); j->merge_backward = (double *) (blk + oMB);
    j->col_total = (double *) (blk + oTotal);
    j->hmm_forward = (double *) (blk + oFB); j->hmm_backward = (double *) (blk + oFB) + 1;
}

/* the post-conditions of stRPHmm_forwardBackward, written back in the order the graph was flattened */
static void adp_scatter(adp_flat *f) {
    stRPHmm *hmm = f->hmm;
    const mrp_hmm_job *j = &f->job;
    int64_t c_ = 0, m_ = 0, k = 0;
    for (stRPColumn *c = hmm->firstColumn;; c = ADP_MCOL_NEXT(c->nColumn), k++) {
        c->totalLogProb = j->col_total[k];
        for (stRPCell *cell = c->head; cell != NULL; cell = cell->nCell) {
            cell->forwardLogProb = j->cell_forward[c_];
            cell->backwardLogProb = j->cell_backward[c_];
            c_++;
        }
        if (c->nColumn == NULL) break;
        const int64_t n = ADP_MCELLS_LEN(f->mcells[k]);
        for (int64_t i = 0; i < n; i++) {
            stRPMergeCell *mc = ADP_MCELLS_AT(f->mcells[k], i);
            mc->forwardLogProb = j->merge_forward[m_];
            mc->backwardLogProb = j->merge_backward[m_];
            m_++;
        }
    }
    hmm->forwardLogProb = *j->hmm_forward;
    hmm->backwardLogProb = *j->hmm_backward;
}

static void adp_release(adp_flat *f) {
    const int64_t K = f->hmm->columnNumber;
    for (int64_t k = 0; k + 1 < K; k++) ADP_MCELLS_FREE(f->mcells[k]);
    free(f->mcells);
    free(f->block);
    if (f->hasTransient) adp_chunk_free(&f->transient);
}

/* ---- the entry points ------------------------------------------------------------------------------------------- */
void stRPHmm_forwardBackwardMany(stRPHmm **hmms, int64_t n) {
    if (n <= 0) return;
    adp_flat *flat = ADP_MALLOC(sizeof(*flat) * (size_t) n);
    mrp_hmm_job *jobs = ADP_MALLOC(sizeof(*jobs) * (size_t) n);
    const int64_t limit = adp_min_cells();
    int64_t m = 0;
    for (int64_t i = 0; i < n; i++) {
        if (limit > 0 && adp_is_small(hmms[i], limit)) { ADP_CPU_FORWARD_BACKWARD(hmms[i]); continue; }
        adp_flatten(hmms[i], &flat[m]);
        jobs[m] = flat[m].job;
        m++;
    }
    if (m > 0 && mrp_fb_run(adp_context(), m, jobs) != MRP_OK) ADP_ABORT("margin_rphmm: %s", mrp_last_error());
    for (int64_t i = 0; i < m; i++) { adp_scatter(&flat[i]); adp_release(&flat[i]); }
    free(jobs);
    free(flat);
}

void stRPHmm_forwardBackward(stRPHmm *hmm) { /* inc/margin.h:369, impl/hmm.c:931-942 */
    stRPHmm_forwardBackwardMany(&hmm, 1);
}

/* at thread exit (end of the OpenMP region of phase.c:276-473) */
void mrpAdaptor_threadCleanup(void) {
    for (int64_t i = 0; i < adpChunkNo; i++) adp_chunk_free(&adpChunks[i]);
    free(adpChunks);
    adpChunks = NULL; adpChunkNo = adpChunkCap = 0;
    mrp_context_destroy(adpCtx);
    adpCtx = NULL;
}
