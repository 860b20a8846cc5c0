/*
 * rphmm_host.c -- host pipeline of libmargin_rphmm.so, in C as the reference's host code is.
 *
 * The structural stRPHmm operations of impl/hmm.c, column.c, mergeColumn.c, coordination.c,
 * genomeFragment.c and the phasing driver bubbleGraph.c:2673-2801, re-designed around ONE flat
 * structure-of-arrays hmm (struct mrp_hmm) whose arrays are exactly the arrays of mrp_hmm_job:
 * cells are rows of (partition, next, prev), merge cells rows of (from, to), transitions are
 * indices instead of hash lookups.  Nothing is flattened before a sweep; the device batch is a
 * memcpy of these arrays.  Every forward/backward sweep runs on the GPU via mrp_fb_run /
 * mrp_batch_*; there is no CPU sweep in this file.
 *
 * Order conventions (DESIGN.md "Order semantics"): cell order is the reference's list order;
 * merge cells keep creation order; stList_sort2 is taken to be stable; stHash/stSet iteration
 * (address dependent in the reference) is creation order.
 */
#define _GNU_SOURCE
#include "rphmm_host.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return 1e3 * (double) ts.tv_sec + 1e-6 * (double) ts.tv_nsec;
}

/* host worker threads: the structural code is independent per chunk / per merge node */
#include <pthread.h>
typedef void (*par_fn)(int64_t i, void *arg);
static void parallel_for(int64_t n, par_fn fn, void *arg) { mrp_pool_run(n, 1, fn, arg); } /* persistent pool, mrp_api.cpp */

/* ------------------------------------------------------------------------------------------ */
/* helpers                                                                                     */
/* ------------------------------------------------------------------------------------------ */
/* Scratch arena of the calling thread.  One merge of the resident pipeline (r_prepare_merge) makes some thirty small
 * allocations that all die before it returns -- component lists, tiling paths, piece lists -- and a call makes 10^5 merges on
 * 16 threads: a quarter of the host CPU time of a call went into malloc/free.  While the arena is switched on, xmalloc /
 * xcalloc / xrealloc (hence VEC_PUSH) bump-allocate from it; free() of such a pointer is a no-op (this file's free() checks
 * the range), the arena is rewound when the merge is done.  What outlives the merge (shadows, the result path, the garbage
 * list) is allocated with the arena switched off.  A block never changes threads while the arena owns it. */
typedef struct { char *base; size_t cap, used; int active; } tl_arena;
static __thread tl_arena t_ar;
static pthread_key_t ar_key;
static pthread_once_t ar_once = PTHREAD_ONCE_INIT;
static void ar_release(void *p) { free(p); }
static void ar_make_key(void) { (void) pthread_key_create(&ar_key, ar_release); }
static inline int ar_owns(const void *p) { return t_ar.base && (const char *) p >= t_ar.base && (const char *) p < t_ar.base + t_ar.cap; }
static void *ar_alloc(size_t n) {
    if (!t_ar.base) {
        (void) pthread_once(&ar_once, ar_make_key);
        t_ar.cap = (size_t) 8 << 20;
        t_ar.base = malloc(t_ar.cap);
        if (!t_ar.base) { t_ar.cap = 0; return NULL; }
        (void) pthread_setspecific(ar_key, t_ar.base); /* freed when the thread ends */
    }
    n = (n + 15) & ~(size_t) 15;
    if (t_ar.used + n + 16 > t_ar.cap) return NULL; /* does not fit: the caller takes it from the heap */
    char *q = t_ar.base + t_ar.used;
    *(size_t *) q = n;
    t_ar.used += n + 16;
    return q + 16;
}
static inline void ar_on(void) { t_ar.active = 1; }
static inline void ar_off(void) { t_ar.active = 0; }
static inline void ar_rewind(void) { t_ar.used = 0; t_ar.active = 0; }

static void *xmalloc(size_t n) { /* like st_malloc: out of memory is fatal */
    if (t_ar.active) { void *a = ar_alloc(n ? n : 1); if (a) return a; }
    void *p = malloc(n ? n : 1);
    if (!p) { fprintf(stderr, "margin_rphmm: out of host memory\n"); abort(); }
    return p;
}
static void *xcalloc(size_t n, size_t s) {
    if (t_ar.active) { void *a = ar_alloc((n ? n : 1) * (s ? s : 1)); if (a) { memset(a, 0, (n ? n : 1) * (s ? s : 1)); return a; } }
    void *p = calloc(n ? n : 1, s ? s : 1);
    if (!p) { fprintf(stderr, "margin_rphmm: out of host memory\n"); abort(); }
    return p;
}
static void *xrealloc(void *q, size_t n) {
    if (q && ar_owns(q)) { /* grows inside the arena (or moves to the heap when the arena is full or off) */
        const size_t old = *(size_t *) ((char *) q - 16);
        if (n <= old) return q;
        void *p = xmalloc(n);
        memcpy(p, q, old);
        return p;
    }
    if (!q) return xmalloc(n);
    void *p = realloc(q, n ? n : 1);
    if (!p) { fprintf(stderr, "margin_rphmm: out of host memory\n"); abort(); }
    return p;
}
void mrp_free(void *p) { free(p); }
static inline void tl_free(void *p) { if (p && !ar_owns(p)) free(p); }
#define free(p) tl_free(p) /* from here on: a block of the thread's scratch arena is not given to the heap */

#define VEC(T) struct { T *a; int64_t n, cap; }
#define VEC_PUSH(v, x)                                                                        \
    do {                                                                                      \
        if ((v).n == (v).cap) {                                                               \
            (v).cap = (v).cap ? (v).cap * 2 : 16;                                             \
            (v).a = xrealloc((v).a, sizeof(*(v).a) * (size_t) (v).cap);                       \
        }                                                                                     \
        (v).a[(v).n++] = (x);                                                                 \
    } while (0)
#define VEC_RESERVE(v, extra)                                                                 \
    do {                                                                                      \
        if ((v).n + (int64_t) (extra) > (v).cap) {                                            \
            while ((v).n + (int64_t) (extra) > (v).cap) (v).cap = (v).cap ? (v).cap * 2 : 16; \
            (v).a = xrealloc((v).a, sizeof(*(v).a) * (size_t) (v).cap);                       \
        }                                                                                     \
    } while (0)

static inline uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
/* uint64 -> uint32 open-addressing map (stands in for the stHash of mergeColumn.c:27-31) */
typedef struct { uint64_t *k; uint32_t *v; uint64_t mask; } u64map;
#define U64MAP_EMPTY 0xFFFFFFFFu
static void u64map_init(u64map *m, int64_t expect) {
    uint64_t cap = 16;
    while (cap < (uint64_t) expect * 2) cap *= 2;
    m->mask = cap - 1;
    m->k = xmalloc(sizeof(uint64_t) * cap);
    m->v = xmalloc(sizeof(uint32_t) * cap);
    memset(m->v, 0xFF, sizeof(uint32_t) * cap);
}
static void u64map_free(u64map *m) { free(m->k); free(m->v); m->k = NULL; m->v = NULL; }
static inline uint32_t u64map_get(const u64map *m, uint64_t key) {
    uint64_t i = mix64(key) & m->mask;
    while (m->v[i] != U64MAP_EMPTY) {
        if (m->k[i] == key) return m->v[i];
        i = (i + 1) & m->mask;
    }
    return U64MAP_EMPTY;
}
static inline void u64map_put(u64map *m, uint64_t key, uint32_t val) { /* caller sized the map */
    uint64_t i = mix64(key) & m->mask;
    while (m->v[i] != U64MAP_EMPTY) {
        if (m->k[i] == key) { m->v[i] = val; return; }
        i = (i + 1) & m->mask;
    }
    m->k[i] = key; m->v[i] = val;
}

/* partitions.c */
static inline uint64_t accept_mask(int64_t depth) { /* :13-19 */
    return depth < 64 ? ~(0xFFFFFFFFFFFFFFFFULL << depth) : 0xFFFFFFFFFFFFFFFFULL;
}
static inline uint64_t merge_bits(uint64_t p1, uint64_t p2, int64_t d1) { /* :21-28 */
    return d1 < 64 ? ((p2 << d1) | p1) : p1;
}
static inline uint64_t invert_partition(uint64_t p, int64_t depth) { return accept_mask(depth) & ~p; } /* :37-42 */

/* Blocks of the shadow hmms of the device-resident merge: tens of thousands per call, a few hundred bytes to a few hundred
 * kilobytes each, all dead by the end of the call.  Through malloc the large ones are mapped and unmapped one by one (and
 * page-faulted in again by the next call); here they are kept on per-size-class stacks (powers of two) and reused warm. */
#define SHADOW_MIN_LOG 9
#define SHADOW_CLASSES 20 /* 512 B .. 256 MB */
static struct { pthread_mutex_t mu; void **stack; int64_t n, cap; } g_shadow[SHADOW_CLASSES];
static pthread_once_t g_shadow_once = PTHREAD_ONCE_INIT;
static int shadow_class(size_t bytes) {
    int c = 0;
    while (((size_t) 1 << (c + SHADOW_MIN_LOG)) < bytes) c++;
    return c;
}
/* per-thread front of the pool: the shared stacks are touched a batch at a time.  The fronts are large for the small classes
 * -- a level of the first merges makes tens of thousands of blocks of 512 B .. 2 KB on the worker threads and the thread that
 * drives the batch gives the parents' blocks back: everything flows through the shared stack, and with fronts of 16 the
 * class mutex was taken every eighth block by 24 threads (a fifth of a call's host CPU time in futex calls). */
#define SHADOW_TL 256
#define SHADOW_TL_LARGE 16
#define SHADOW_SMALL_CLASSES 6 /* up to 16 KB */
static inline int shadow_front(int c) { return c < SHADOW_SMALL_CLASSES ? SHADOW_TL : SHADOW_TL_LARGE; }
typedef struct { void *slot[SHADOW_CLASSES][SHADOW_TL]; int n[SHADOW_CLASSES]; int keyed; } shadow_tl;
static __thread shadow_tl t_shadow;
static pthread_key_t g_shadow_key;
static void shadow_spill(int c, shadow_tl *f, int keep) { /* front -> shared stack */
    pthread_mutex_lock(&g_shadow[c].mu);
    if (g_shadow[c].n + f->n[c] > g_shadow[c].cap) {
        while (g_shadow[c].n + f->n[c] > g_shadow[c].cap) g_shadow[c].cap = g_shadow[c].cap ? 2 * g_shadow[c].cap : 1024;
        void **grown = realloc(g_shadow[c].stack, sizeof(void *) * (size_t) g_shadow[c].cap);
        if (!grown) { fprintf(stderr, "margin_rphmm: out of host memory\n"); abort(); }
        g_shadow[c].stack = grown;
    }
    while (f->n[c] > keep) g_shadow[c].stack[g_shadow[c].n++] = f->slot[c][--f->n[c]];
    pthread_mutex_unlock(&g_shadow[c].mu);
}
static void shadow_thread_exit(void *arg) { /* a thread ends (the threads of a call's concurrent batches do): its front goes back */
    shadow_tl *f = arg;
    for (int c = 0; c < SHADOW_CLASSES; c++) if (f->n[c] > 0) shadow_spill(c, f, 0);
}
static void shadow_pool_init(void) {
    for (int i = 0; i < SHADOW_CLASSES; i++) pthread_mutex_init(&g_shadow[i].mu, NULL);
    (void) pthread_key_create(&g_shadow_key, shadow_thread_exit);
}
static inline void shadow_thread_enter(void) { /* every thread that holds blocks in its front is registered: its front is spilled when it ends */
    if (__builtin_expect(t_shadow.keyed, 1)) return;
    pthread_once(&g_shadow_once, shadow_pool_init);
    t_shadow.keyed = 1;
    (void) pthread_setspecific(g_shadow_key, &t_shadow);
}
static void *shadow_alloc(size_t bytes, int *cls_out) {
    const int c = shadow_class(bytes);
    if (c >= SHADOW_CLASSES) { *cls_out = -1; return xmalloc(bytes); }
    *cls_out = c;
    shadow_thread_enter();
    if (t_shadow.n[c] > 0) return t_shadow.slot[c][--t_shadow.n[c]];
    /* refill: up to half a front from the shared stack */
    pthread_mutex_lock(&g_shadow[c].mu);
    while (t_shadow.n[c] < shadow_front(c) / 2 && g_shadow[c].n > 0) t_shadow.slot[c][t_shadow.n[c]++] = g_shadow[c].stack[--g_shadow[c].n];
    pthread_mutex_unlock(&g_shadow[c].mu);
    if (t_shadow.n[c] > 0) return t_shadow.slot[c][--t_shadow.n[c]];
    const int was = t_ar.active; /* (never from the scratch arena: the block outlives the merge) */
    t_ar.active = 0;
    void *p = xmalloc((size_t) 1 << (c + SHADOW_MIN_LOG));
    t_ar.active = was;
    return p;
}
static void shadow_release(void *p, int cls) {
    if (cls < 0) { free(p); return; }
    shadow_thread_enter(); /* (a thread that only ever releases -- a batch's own thread, the clean-up of a call -- used to keep its front for good) */
    if (t_shadow.n[cls] == shadow_front(cls)) { /* spill half of the front */
        shadow_spill(cls, &t_shadow, shadow_front(cls) / 2);
    }
    t_shadow.slot[cls][t_shadow.n[cls]++] = p;
}

/* ------------------------------------------------------------------------------------------ */
/* the flat hmm                                                                                */
/* ------------------------------------------------------------------------------------------ */
struct mrp_hmm {
    int32_t ref_start, ref_length; /* stRPHmm.refStart / refLength */
    int32_t max_depth;
    VEC(int32_t) reads;            /* stRPHmm.profileSeqs (read indices) */
    /* columns */
    VEC(int32_t) col_start, col_len, col_depth;
    VEC(int64_t) cell_off, read_off;   /* K+1 */
    VEC(int32_t) col_reads;            /* per column, bit order */
    VEC(int64_t) read_byte_off;        /* per column per read: offset of column->seqs[i] in the pool */
    /* cells */
    VEC(uint64_t) part;
    VEC(uint32_t) next, prev;
    /* merge columns */
    VEC(uint64_t) mask_from, mask_to;  /* K-1 */
    VEC(int64_t) mcell_off;            /* K (first entry 0) */
    VEC(uint64_t) mfrom, mto;
    /* results of the last sweep */
    double *f, *b, *mf, *mb, *total;
    double fwd, bwd;
    int has_results;
};

static int64_t hmm_K(const mrp_hmm *h) { return h->col_start.n; }

static mrp_hmm *hmm_new(void) {
    mrp_hmm *h = xcalloc(1, sizeof(*h));
    VEC_PUSH(h->cell_off, 0);
    VEC_PUSH(h->read_off, 0);
    VEC_PUSH(h->mcell_off, 0);
    return h;
}
static void hmm_free_results(mrp_hmm *h) {
    free(h->f); free(h->b); free(h->mf); free(h->mb); free(h->total);
    h->f = h->b = h->mf = h->mb = h->total = NULL;
    h->has_results = 0;
}
static void hmm_free_array(const mrp_hmm *h, void *p) { (void) h; free(p); }
void mrp_hmm_destroy(mrp_hmm *h) {
    if (!h) return;
    void *arrays[] = {h->reads.a, h->col_start.a, h->col_len.a, h->col_depth.a, h->cell_off.a, h->read_off.a, h->col_reads.a,
                      h->read_byte_off.a, h->part.a, h->next.a, h->prev.a, h->mask_from.a, h->mask_to.a, h->mcell_off.a,
                      h->mfrom.a, h->mto.a};
    for (size_t i = 0; i < sizeof(arrays) / sizeof(arrays[0]); i++) hmm_free_array(h, arrays[i]);
    hmm_free_results(h);
    free(h);
}

/* per-job view of the reads + chunk the structural code works against */
typedef struct {
    const mrp_chunk *chunk;
    mrp_chunk_host ch;
    const mrp_read *reads;
    int64_t n_reads;
    mrp_context *ctx;
    mrp_batch *record;
    int64_t n_sweeps;
    uint32_t max_alleles;
    int failed; /* resident path: a kernel asked for this chunk to be redone on the hashing path */
} world;

static int64_t read_byte_offset(const world *w, int32_t read, int32_t site) { /* profileSeq.c:41-47 */
    const mrp_read *r = &w->reads[read];
    return r->pool_offset + (int64_t) (w->ch.allele_offset[site] - w->ch.allele_offset[r->ref_start]);
}

/* begin a column; cells are appended afterwards */
static void hmm_begin_column(mrp_hmm *h, const world *w, int32_t start, int32_t len, int32_t depth,
                             const int32_t *col_reads) {
    VEC_PUSH(h->col_start, start);
    VEC_PUSH(h->col_len, len);
    VEC_PUSH(h->col_depth, depth);
    for (int32_t i = 0; i < depth; i++) {
        VEC_PUSH(h->col_reads, col_reads[i]);
        VEC_PUSH(h->read_byte_off, read_byte_offset(w, col_reads[i], start));
    }
    if (depth > h->max_depth) h->max_depth = depth;
}
static void hmm_end_column(mrp_hmm *h) {
    VEC_PUSH(h->cell_off, h->part.n);
    VEC_PUSH(h->read_off, h->col_reads.n);
}
static inline void hmm_add_cell(mrp_hmm *h, uint64_t p, uint32_t prev) {
    VEC_PUSH(h->part, p);
    VEC_PUSH(h->prev, prev);
    VEC_PUSH(h->next, 0u);
}
static void hmm_begin_merge(mrp_hmm *h, uint64_t mask_from, uint64_t mask_to) {
    VEC_PUSH(h->mask_from, mask_from);
    VEC_PUSH(h->mask_to, mask_to);
}
static void hmm_end_merge(mrp_hmm *h) { VEC_PUSH(h->mcell_off, h->mfrom.n); }

/* stRPHmm_construct hmm.c:97-133: one column, cells {1, 0} */
static mrp_hmm *hmm_from_read(const world *w, int32_t read) {
    mrp_hmm *h = hmm_new();
    const mrp_read *r = &w->reads[read];
    h->ref_start = r->ref_start;
    h->ref_length = r->length;
    VEC_PUSH(h->reads, read);
    hmm_begin_column(h, w, r->ref_start, r->length, 1, &read);
    hmm_add_cell(h, 1, 0);
    hmm_add_cell(h, 0, 0);
    hmm_end_column(h);
    return h;
}

#define HMM_T mrp_hmm
#define PFX(x) x
#define H_NAME_READ(h) ((h)->reads.n > 0 ? (h)->reads.a[0] : -1)
#include "rphmm_paths.inc"
#undef HMM_T
#undef PFX
#undef H_NAME_READ

/* ------------------------------------------------------------------------------------------ */
/* fuse + align + cross product in one pass                                                    */
/* ------------------------------------------------------------------------------------------ */
/* A piece is a column of a source hmm (or a gap) restricted to a site interval; a connector is
 * the merge column that leads out of it.  stRPHmm_fuse (hmm.c:283-372) contributes ZERO
 * connectors and gap pieces, stRPHmm_alignColumns (hmm.c:374-507) the prefix/suffix gaps and,
 * through stRPColumn_split (column.c:70-130), the IDENT connectors. */
typedef enum { CONN_NONE = 0, CONN_REAL, CONN_ZERO, CONN_IDENT } conn_kind;
typedef struct {
    const mrp_hmm *h; /* NULL = gap column (depth 0, one cell, partition 0) */
    int32_t k;        /* column in h */
    int32_t start, len;
    conn_kind out;    /* connector to the next piece */
} piece;
typedef VEC(piece) piece_vec;

static void pieces_of_path(const hmm_vec *tp, int32_t S, int32_t E, piece_vec *out) {
    int32_t pos = S;
    for (int64_t i = 0; i < tp->n; i++) {
        const mrp_hmm *h = tp->a[i];
        if (h->ref_start > pos) { /* gap (hmm.c:335-359, :396-424) */
            piece g = {NULL, 0, pos, h->ref_start - pos, CONN_ZERO};
            VEC_PUSH(*out, g);
        }
        const int64_t K = hmm_K(h);
        for (int64_t k = 0; k < K; k++) {
            piece p = {h, (int32_t) k, h->col_start.a[k], h->col_len.a[k], k + 1 < K ? CONN_REAL : CONN_ZERO};
            VEC_PUSH(*out, p);
        }
        pos = h->ref_start + h->ref_length;
    }
    if (pos < E) { /* suffix gap (hmm.c:435-462) */
        piece g = {NULL, 0, pos, E - pos, CONN_ZERO};
        VEC_PUSH(*out, g);
    }
    out->a[out->n - 1].out = CONN_NONE;
}
/* cut both piece lists at the union of their boundaries (hmm.c:476-504) */
static void align_pieces(const piece_vec *a, const piece_vec *b, piece_vec *oa, piece_vec *ob) {
    int64_t i = 0, j = 0;
    piece pa = a->a[0], pb = b->a[0];
    while (1) {
        const int32_t len = pa.len < pb.len ? pa.len : pb.len;
        piece ca = pa, cb = pb;
        ca.len = len; cb.len = len;
        if (pa.len > len) ca.out = CONN_IDENT;
        if (pb.len > len) cb.out = CONN_IDENT;
        VEC_PUSH(*oa, ca);
        VEC_PUSH(*ob, cb);
        if (pa.len > len) { pa.start += len; pa.len -= len; } else { i++; if (i < a->n) pa = a->a[i]; }
        if (pb.len > len) { pb.start += len; pb.len -= len; } else { j++; if (j < b->n) pb = b->a[j]; }
        if (i >= a->n || j >= b->n) break;
    }
}

static const uint64_t ZERO_PART[1] = {0};
static inline int64_t piece_cells(const piece *p) { return p->h ? p->h->cell_off.a[p->k + 1] - p->h->cell_off.a[p->k] : 1; }
static inline const uint64_t *piece_parts(const piece *p) { return p->h ? p->h->part.a + p->h->cell_off.a[p->k] : ZERO_PART; }
static inline int32_t piece_depth(const piece *p) { return p->h ? p->h->col_depth.a[p->k] : 0; }
static inline const int32_t *piece_reads(const piece *p) { return p->h ? p->h->col_reads.a + p->h->read_off.a[p->k] : NULL; }

/* connector accessors */
typedef struct {
    uint64_t mask_from, mask_to;
    int64_t M;
    const uint64_t *from, *to;
} conn_view;
static void conn_of(const piece *p, conn_view *c) {
    switch (p->out) {
        case CONN_REAL: {
            const mrp_hmm *h = p->h;
            c->mask_from = h->mask_from.a[p->k]; c->mask_to = h->mask_to.a[p->k];
            c->M = h->mcell_off.a[p->k + 1] - h->mcell_off.a[p->k];
            c->from = h->mfrom.a + h->mcell_off.a[p->k]; c->to = h->mto.a + h->mcell_off.a[p->k];
            break;
        }
        case CONN_IDENT: { /* column.c:86-101 */
            c->mask_from = c->mask_to = accept_mask(piece_depth(p));
            c->M = piece_cells(p);
            c->from = c->to = piece_parts(p);
            break;
        }
        default: /* ZERO: hmm.c:324-331 */
            c->mask_from = c->mask_to = 0; c->M = 1; c->from = c->to = ZERO_PART;
    }
}

/* stRPHmm_createCrossProductOfTwoAlignedHmm hmm.c:534-750 over two aligned piece lists */
static mrp_hmm *cross_product(const world *w, const piece_vec *A, const piece_vec *B, const hmm_vec *tpA,
                              const hmm_vec *tpB, const mrp_params *params, int32_t S, int32_t E) {
    mrp_hmm *h = hmm_new();
    h->ref_start = S; h->ref_length = E - S;
    for (int64_t i = 0; i < tpA->n; i++) for (int64_t r = 0; r < tpA->a[i]->reads.n; r++) VEC_PUSH(h->reads, tpA->a[i]->reads.a[r]);
    for (int64_t i = 0; i < tpB->n; i++) for (int64_t r = 0; r < tpB->a[i]->reads.n; r++) VEC_PUSH(h->reads, tpB->a[i]->reads.a[r]);
    const int inv = params->include_inverted_partitions != 0;
    const int64_t n = A->n;
    u64map prev_to = {0}; /* toPartition -> merge index of the merge column before the current column */
    int have_prev = 0;
    uint64_t prev_mask_to = 0;
    int32_t colreads[MRP_MAX_READ_PARTITIONING_DEPTH];
    for (int64_t s = 0; s < n; s++) {
        const piece *pa = &A->a[s], *pb = &B->a[s];
        const int32_t d1 = piece_depth(pa), d2 = piece_depth(pb), depth = d1 + d2;
        if (depth > MRP_MAX_READ_PARTITIONING_DEPTH) {
            mrp_hmm_destroy(h); u64map_free(&prev_to);
            mrp_set_error(MRP_ERR_ARG, "cross product column depth %d exceeds %d", depth, MRP_MAX_READ_PARTITIONING_DEPTH);
            return NULL;
        }
        if (d1) memcpy(colreads, piece_reads(pa), sizeof(int32_t) * (size_t) d1);
        if (d2) memcpy(colreads + d1, piece_reads(pb), sizeof(int32_t) * (size_t) d2);
        hmm_begin_column(h, w, pa->start, pa->len, depth, colreads);
        const int64_t C1 = piece_cells(pa), C2 = piece_cells(pb);
        const uint64_t *P1 = piece_parts(pa), *P2 = piece_parts(pb);
        const int64_t cell0 = h->part.n;
        VEC_RESERVE(h->part, 2 * C1 * C2); VEC_RESERVE(h->prev, 2 * C1 * C2); VEC_RESERVE(h->next, 2 * C1 * C2);
        if (inv) { /* hmm.c:627-655 */
            u64map seen; u64map_init(&seen, 2 * C1 * C2);
            for (int64_t c1 = 0; c1 < C1; c1++)
                for (int64_t c2 = 0; c2 < C2; c2++) {
                    const uint64_t p = merge_bits(P1[c1], P2[c2], d1);
                    if (u64map_get(&seen, p) == U64MAP_EMPTY) {
                        u64map_put(&seen, p, 1);
                        hmm_add_cell(h, p, 0);
                        if (depth > 0) {
                            const uint64_t ip = invert_partition(p, depth);
                            u64map_put(&seen, ip, 1);
                            hmm_add_cell(h, ip, 0);
                        }
                    }
                }
            u64map_free(&seen);
        } else { /* hmm.c:657-668 */
            for (int64_t c1 = 0; c1 < C1; c1++)
                for (int64_t c2 = 0; c2 < C2; c2++) hmm_add_cell(h, merge_bits(P1[c1], P2[c2], d1), 0);
        }
        hmm_end_column(h);
        const int64_t nC = h->part.n - cell0;
        /* link to the previous merge column (mergeColumn.c:72-79) */
        if (have_prev) {
            for (int64_t c = 0; c < nC; c++) {
                const uint32_t m = u64map_get(&prev_to, h->part.a[cell0 + c] & prev_mask_to);
                if (m == U64MAP_EMPTY) {
                    mrp_hmm_destroy(h); u64map_free(&prev_to);
                    mrp_set_error(MRP_ERR_LOOKUP, "cross product: cell without previous merge cell");
                    return NULL;
                }
                h->prev.a[cell0 + c] = m;
            }
            u64map_free(&prev_to);
            have_prev = 0;
        }
        if (s + 1 == n) break;
        /* merge column hmm.c:686-740 */
        conn_view ca, cb;
        conn_of(pa, &ca); conn_of(pb, &cb);
        const int32_t d1n = piece_depth(&A->a[s + 1]), d2n = piece_depth(&B->a[s + 1]);
        const uint64_t from_mask = merge_bits(ca.mask_from, cb.mask_from, d1);
        const uint64_t to_mask = merge_bits(ca.mask_to, cb.mask_to, d1n);
        hmm_begin_merge(h, from_mask, to_mask);
        const int64_t m0 = h->mfrom.n;
        u64map from_map; u64map_init(&from_map, 2 * ca.M * cb.M);
        u64map_init(&prev_to, 2 * ca.M * cb.M);
        for (int64_t i = 0; i < ca.M; i++)
            for (int64_t j = 0; j < cb.M; j++) {
                const uint64_t from = merge_bits(ca.from[i], cb.from[j], d1);
                const uint64_t to = merge_bits(ca.to[i], cb.to[j], d1n);
                if (inv) {
                    if (u64map_get(&from_map, from) == U64MAP_EMPTY) {
                        u64map_put(&from_map, from, (uint32_t) (h->mfrom.n - m0));
                        u64map_put(&prev_to, to, (uint32_t) (h->mfrom.n - m0));
                        VEC_PUSH(h->mfrom, from); VEC_PUSH(h->mto, to);
                        if (__builtin_popcountll(from_mask) > 0) {
                            const uint64_t ifrom = from_mask & invert_partition(from, d1 + d2);
                            const uint64_t ito = to_mask & invert_partition(to, d1n + d2n);
                            u64map_put(&from_map, ifrom, (uint32_t) (h->mfrom.n - m0));
                            u64map_put(&prev_to, ito, (uint32_t) (h->mfrom.n - m0));
                            VEC_PUSH(h->mfrom, ifrom); VEC_PUSH(h->mto, ito);
                        }
                    }
                } else {
                    u64map_put(&from_map, from, (uint32_t) (h->mfrom.n - m0));
                    u64map_put(&prev_to, to, (uint32_t) (h->mfrom.n - m0));
                    VEC_PUSH(h->mfrom, from); VEC_PUSH(h->mto, to);
                }
            }
        hmm_end_merge(h);
        /* link this column's cells to it (mergeColumn.c:63-70) */
        for (int64_t c = 0; c < nC; c++) {
            const uint32_t m = u64map_get(&from_map, h->part.a[cell0 + c] & from_mask);
            if (m == U64MAP_EMPTY) {
                mrp_hmm_destroy(h); u64map_free(&from_map); u64map_free(&prev_to);
                mrp_set_error(MRP_ERR_LOOKUP, "cross product: cell without next merge cell");
                return NULL;
            }
            h->next.a[cell0 + c] = m;
        }
        u64map_free(&from_map);
        have_prev = 1;
        prev_mask_to = to_mask;
    }
    return h;
}

/* fuseTilingPath coordination.c:244-261 without a partner: concatenate hmms with ZERO connectors
 * and gap columns (hmm.c:283-372). */
static mrp_hmm *fuse_path(const world *w, const hmm_vec *tp) {
    if (tp->n == 1) return tp->a[0];
    mrp_hmm *h = hmm_new();
    h->ref_start = tp->a[0]->ref_start;
    h->ref_length = tp->a[tp->n - 1]->ref_start + tp->a[tp->n - 1]->ref_length - h->ref_start;
    piece_vec ps = {0};
    pieces_of_path(tp, h->ref_start, h->ref_start + h->ref_length, &ps);
    for (int64_t i = 0; i < tp->n; i++) for (int64_t r = 0; r < tp->a[i]->reads.n; r++) VEC_PUSH(h->reads, tp->a[i]->reads.a[r]);
    for (int64_t s = 0; s < ps.n; s++) {
        const piece *p = &ps.a[s];
        hmm_begin_column(h, w, p->start, p->len, piece_depth(p), piece_reads(p));
        const int64_t C = piece_cells(p);
        const uint64_t *P = piece_parts(p);
        const int real_prev = s > 0 && ps.a[s - 1].out == CONN_REAL;
        const int real_next = p->out == CONN_REAL;
        for (int64_t c = 0; c < C; c++) {
            hmm_add_cell(h, P[c], real_prev ? p->h->prev.a[p->h->cell_off.a[p->k] + c] : 0);
            h->next.a[h->next.n - 1] = real_next ? p->h->next.a[p->h->cell_off.a[p->k] + c] : 0;
        }
        hmm_end_column(h);
        if (p->out == CONN_NONE) break;
        conn_view cv; conn_of(p, &cv);
        hmm_begin_merge(h, cv.mask_from, cv.mask_to);
        for (int64_t m = 0; m < cv.M; m++) { VEC_PUSH(h->mfrom, cv.from[m]); VEC_PUSH(h->mto, cv.to[m]); }
        hmm_end_merge(h);
    }
    free(ps.a);
    for (int64_t i = 0; i < tp->n; i++) mrp_hmm_destroy(tp->a[i]);
    return h;
}

/* ------------------------------------------------------------------------------------------ */
/* sweeps on the device                                                                        */
/* ------------------------------------------------------------------------------------------ */
static uint32_t sweep_flags(const mrp_params *p) {
    return (p->max_not_sum_transitions ? MRP_FLAG_MAX_NOT_SUM : 0u) |
           (p->include_ancestor_sub_prob ? MRP_FLAG_INCLUDE_ANCESTOR_SUB_PROB : 0u);
}
static void hmm_alloc_results(mrp_hmm *h) {
    hmm_free_results(h);
    const int64_t K = hmm_K(h);
    h->f = xmalloc(sizeof(double) * (size_t) h->part.n);
    h->b = xmalloc(sizeof(double) * (size_t) h->part.n);
    h->mf = xmalloc(sizeof(double) * (size_t) (h->mfrom.n + 1));
    h->mb = xmalloc(sizeof(double) * (size_t) (h->mfrom.n + 1));
    h->total = xmalloc(sizeof(double) * (size_t) K);
    h->has_results = 1;
}
static void hmm_job(const world *w, mrp_hmm *h, uint32_t flags, mrp_hmm_job *j, int with_outputs) {
    memset(j, 0, sizeof(*j));
    j->chunk = w->chunk;
    j->n_columns = (int32_t) hmm_K(h);
    j->flags = flags;
    j->col_ref_start = h->col_start.a; j->col_length = h->col_len.a; j->col_depth = h->col_depth.a;
    j->col_cell_off = h->cell_off.a; j->col_read_off = h->read_off.a; j->read_byte_off = h->read_byte_off.a;
    j->partition = h->part.a; j->mask_from = h->mask_from.a; j->mask_to = h->mask_to.a;
    j->mcol_cell_off = h->mcell_off.a; j->merge_from = h->mfrom.a; j->merge_to = h->mto.a;
    j->cell_next = h->next.a; j->cell_prev = h->prev.a;
    if (with_outputs) {
        j->cell_forward = h->f; j->cell_backward = h->b; j->merge_forward = h->mf; j->merge_backward = h->mb;
        j->col_total = h->total; j->hmm_forward = &h->fwd; j->hmm_backward = &h->bwd;
    }
}
/* stRPHmm_forwardBackward for a set of independent hmms: one device batch */
static int sweep_many(world *w, mrp_hmm **hmms, int64_t n, const mrp_params *params) {
    if (n == 0) return MRP_OK;
    const uint32_t flags = sweep_flags(params);
    mrp_hmm_job *jobs = xcalloc((size_t) n, sizeof(*jobs));
    for (int64_t i = 0; i < n; i++) {
        hmm_alloc_results(hmms[i]);
        hmm_job(w, hmms[i], flags, &jobs[i], 1);
    }
    int rc = mrp_fb_run(w->ctx, n, jobs);
    if (rc == MRP_OK && w->record) {
        for (int64_t i = 0; rc == MRP_OK && i < n; i++) {
            mrp_hmm_job dj;
            hmm_job(w, hmms[i], flags, &dj, 0);
            rc = mrp_batch_add(w->record, &dj);
        }
    }
    w->n_sweeps += n;
    free(jobs);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* prune (hmm.c:944-1163)                                                                      */
/* ------------------------------------------------------------------------------------------ */
typedef struct { int64_t idx; double key; } keyed;
static void keyed_sort_desc(keyed *a, int64_t n, keyed *tmp) { /* stable, descending */
    for (int64_t wdt = 1; wdt < n; wdt *= 2) {
        for (int64_t lo = 0; lo < n; lo += 2 * wdt) {
            const int64_t mid = lo + wdt < n ? lo + wdt : n, hi = lo + 2 * wdt < n ? lo + 2 * wdt : n;
            int64_t i = lo, j = mid, o = lo;
            while (i < mid && j < hi) { if (a[j].key > a[i].key) tmp[o++] = a[j++]; else tmp[o++] = a[i++]; }
            while (i < mid) tmp[o++] = a[i++];
            while (j < hi) tmp[o++] = a[j++];
        }
        memcpy(a, tmp, sizeof(keyed) * (size_t) n);
    }
}
static int posterior(double f, double b, double total, double limit, double *out) { /* column.c:177-193, mergeColumn.c:129-146 */
    const double p = exp(f + b - total);
    if (p > limit || p < 0.0) return mrp_set_error(MRP_ERR_ARG, "ERROR: invalid prob %f", p);
    *out = p > 1.0 ? 1.0 : p;
    return MRP_OK;
}

int mrp_hmm_prune(mrp_hmm *h, const mrp_params *P) {
    if (!h || !P) return mrp_set_error(MRP_ERR_ARG, "mrp_hmm_prune: NULL argument");
    if (!h->has_results) return mrp_set_error(MRP_ERR_ARG, "mrp_hmm_prune before a forward/backward sweep");
    const int64_t K = hmm_K(h);
    int64_t max_c = 1, max_m = 1;
    for (int64_t k = 0; k < K; k++) {
        const int64_t c = h->cell_off.a[k + 1] - h->cell_off.a[k];
        if (c > max_c) max_c = c;
        if (k + 1 < K) { const int64_t m = h->mcell_off.a[k + 1] - h->mcell_off.a[k]; if (m > max_m) max_m = m; }
    }
    /* kept cells per column (old cell indices, new order) and kept-flag per merge cell */
    int64_t *keep_off = xmalloc(sizeof(int64_t) * (size_t) (K + 1));
    int64_t *keep_idx = xmalloc(sizeof(int64_t) * (size_t) (h->part.n + 1));
    uint8_t *keep_m = xcalloc((size_t) (h->mfrom.n + 1), 1);
    keyed *ka = xmalloc(sizeof(keyed) * (size_t) (max_c > max_m ? max_c : max_m));
    keyed *kt = xmalloc(sizeof(keyed) * (size_t) (max_c > max_m ? max_c : max_m));
    uint8_t *chosen = xmalloc((size_t) max_m);
    int rc = MRP_OK;
    /* stRPHmm_pruneForwards hmm.c:1049-1109 */
    keep_off[0] = 0;
    for (int64_t k = 0; k < K && rc == MRP_OK; k++) {
        const int64_t c0 = h->cell_off.a[k], nc = h->cell_off.a[k + 1] - c0;
        int64_t n = 0;
        for (int64_t c = 0; c < nc; c++) { /* getLinkedCells :1021-1047 */
            if (k > 0 && !keep_m[h->mcell_off.a[k - 1] + h->prev.a[c0 + c]]) continue;
            ka[n].idx = c;
            rc = posterior(h->f[c0 + c], h->b[c0 + c], h->total[k], 1.1, &ka[n].key);
            if (rc != MRP_OK) break;
            n++;
        }
        if (rc != MRP_OK) break;
        keyed_sort_desc(ka, n, kt);
        while (n > P->min_partitions_in_a_column &&
               (n > P->max_partitions_in_a_column || ka[n - 1].key < P->min_posterior_probability_for_partition))
            n--;
        for (int64_t i = 0; i < n; i++) keep_idx[keep_off[k] + i] = ka[i].idx;
        keep_off[k + 1] = keep_off[k] + n;
        if (k + 1 == K) break;
        /* getLinkedMergeCells :989-1004, sort + shrink :1088-1101 */
        const int64_t m0 = h->mcell_off.a[k], nm = h->mcell_off.a[k + 1] - m0;
        memset(chosen, 0, (size_t) nm);
        int64_t mn = 0;
        for (int64_t i = 0; i < n; i++) {
            const uint32_t m = h->next.a[c0 + keep_idx[keep_off[k] + i]];
            if (!chosen[m]) {
                chosen[m] = 1;
                ka[mn].idx = m;
                rc = posterior(h->mf[m0 + m], h->mb[m0 + m], h->total[k + 1], 1.001, &ka[mn].key);
                if (rc != MRP_OK) break;
                mn++;
            }
        }
        if (rc != MRP_OK) break;
        keyed_sort_desc(ka, mn, kt);
        while (mn > P->min_partitions_in_a_column &&
               (mn > P->max_partitions_in_a_column || ka[mn - 1].key < P->min_posterior_probability_for_partition))
            mn--;
        for (int64_t i = 0; i < mn; i++) keep_m[m0 + ka[i].idx] = 1;
    }
    /* stRPHmm_pruneBackwards hmm.c:1111-1158 */
    for (int64_t k = K - 1; k >= 0 && rc == MRP_OK; k--) {
        const int64_t c0 = h->cell_off.a[k];
        int64_t n = 0;
        for (int64_t i = keep_off[k]; i < keep_off[k + 1]; i++) {
            const int64_t c = keep_idx[i];
            if (k + 1 < K && !keep_m[h->mcell_off.a[k] + h->next.a[c0 + c]]) continue;
            keep_idx[keep_off[k] + n++] = c; /* order kept: the re-sort of an already sorted list is a no-op */
        }
        /* entries past the new length are marked unused */
        for (int64_t i = keep_off[k] + n; i < keep_off[k + 1]; i++) keep_idx[i] = -1;
        if (k == 0) break;
        const int64_t m0 = h->mcell_off.a[k - 1], nm = h->mcell_off.a[k] - m0;
        memset(chosen, 0, (size_t) nm);
        for (int64_t i = 0; i < n; i++) chosen[h->prev.a[c0 + keep_idx[keep_off[k] + i]]] = 1;
        for (int64_t m = 0; m < nm; m++) keep_m[m0 + m] = keep_m[m0 + m] && chosen[m];
    }
    if (rc == MRP_OK) {
        /* rebuild compactly: merge cells keep their relative order (filterMergeCells :964-987),
         * cells are relinked in sorted order (relinkCells :1006-1019) */
        uint32_t *remap = xmalloc(sizeof(uint32_t) * (size_t) (h->mfrom.n + 1));
        int64_t nm_new = 0;
        int64_t *new_moff = xmalloc(sizeof(int64_t) * (size_t) K);
        new_moff[0] = 0;
        for (int64_t k = 0; k + 1 < K; k++) {
            const int64_t m0 = h->mcell_off.a[k], nm = h->mcell_off.a[k + 1] - m0;
            uint32_t local = 0;
            for (int64_t m = 0; m < nm; m++) {
                if (keep_m[m0 + m]) {
                    remap[m0 + m] = local++;
                    h->mfrom.a[nm_new] = h->mfrom.a[m0 + m];
                    h->mto.a[nm_new] = h->mto.a[m0 + m];
                    h->mf[nm_new] = h->mf[m0 + m];
                    h->mb[nm_new] = h->mb[m0 + m];
                    nm_new++;
                } else remap[m0 + m] = U64MAP_EMPTY;
            }
            new_moff[k + 1] = nm_new;
        }
        const int64_t nC_old = h->part.n;
        uint64_t *np = xmalloc(sizeof(uint64_t) * (size_t) (nC_old + 1));
        uint32_t *nn = xmalloc(sizeof(uint32_t) * (size_t) (nC_old + 1)), *npv = xmalloc(sizeof(uint32_t) * (size_t) (nC_old + 1));
        double *nf = xmalloc(sizeof(double) * (size_t) (nC_old + 1)), *nb = xmalloc(sizeof(double) * (size_t) (nC_old + 1));
        int64_t o = 0;
        int64_t *new_coff = xmalloc(sizeof(int64_t) * (size_t) (K + 1));
        new_coff[0] = 0;
        for (int64_t k = 0; k < K; k++) {
            const int64_t c0 = h->cell_off.a[k];
            for (int64_t i = keep_off[k]; i < keep_off[k + 1]; i++) {
                const int64_t c = keep_idx[i];
                if (c < 0) break;
                np[o] = h->part.a[c0 + c];
                nn[o] = k + 1 < K ? remap[h->mcell_off.a[k] + h->next.a[c0 + c]] : 0;
                npv[o] = k > 0 ? remap[h->mcell_off.a[k - 1] + h->prev.a[c0 + c]] : 0;
                nf[o] = h->f[c0 + c]; nb[o] = h->b[c0 + c];
                o++;
            }
            new_coff[k + 1] = o;
        }
        memcpy(h->part.a, np, sizeof(uint64_t) * (size_t) o);
        memcpy(h->next.a, nn, sizeof(uint32_t) * (size_t) o);
        memcpy(h->prev.a, npv, sizeof(uint32_t) * (size_t) o);
        memcpy(h->f, nf, sizeof(double) * (size_t) o);
        memcpy(h->b, nb, sizeof(double) * (size_t) o);
        h->part.n = h->next.n = h->prev.n = o;
        h->mfrom.n = h->mto.n = nm_new;
        memcpy(h->cell_off.a, new_coff, sizeof(int64_t) * (size_t) (K + 1));
        memcpy(h->mcell_off.a, new_moff, sizeof(int64_t) * (size_t) K);
        free(remap); free(new_moff); free(np); free(nn); free(npv); free(nf); free(nb); free(new_coff);
    }
    free(keep_off); free(keep_idx); free(keep_m); free(ka); free(kt); free(chosen);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* coordination.c                                                                              */
/* ------------------------------------------------------------------------------------------ */
/* mergeTwoTilingPaths coordination.c:263-339.  All cross products of the call are swept in one
 * device batch (the components are independent), then pruned. */
static int merge_two_tiling_paths(world *w, hmm_vec *tp1, hmm_vec *tp2, const mrp_params *params, hmm_vec **out) {
    comp_vec comps = overlapping_components(w, tp1, tp2);
    free(tp1->a); free(tp1); free(tp2->a); free(tp2);
    hmm_vec *res = xcalloc(1, sizeof(*res));
    hmm_vec crossed = {0};
    int rc = MRP_OK;
    for (int64_t i = 0; i < comps.n; i++) {
        component *comp = comps.a[i];
        if (rc == MRP_OK) {
            path_vec sub = tiling_paths_from(w, comp->members.a, comp->members.n);
            if (sub.n == 2) {
                hmm_vec *a = sub.a[0], *b = sub.a[1];
                int32_t S = a->a[0]->ref_start < b->a[0]->ref_start ? a->a[0]->ref_start : b->a[0]->ref_start;
                int32_t Ea = a->a[a->n - 1]->ref_start + a->a[a->n - 1]->ref_length;
                int32_t Eb = b->a[b->n - 1]->ref_start + b->a[b->n - 1]->ref_length;
                int32_t E = Ea > Eb ? Ea : Eb;
                piece_vec pa = {0}, pb = {0}, qa = {0}, qb = {0};
                pieces_of_path(a, S, E, &pa);
                pieces_of_path(b, S, E, &pb);
                align_pieces(&pa, &pb, &qa, &qb);
                mrp_hmm *x = cross_product(w, &qa, &qb, a, b, params, S, E);
                free(pa.a); free(pb.a); free(qa.a); free(qb.a);
                for (int64_t t = 0; t < a->n; t++) mrp_hmm_destroy(a->a[t]);
                for (int64_t t = 0; t < b->n; t++) mrp_hmm_destroy(b->a[t]);
                if (x) { VEC_PUSH(crossed, x); VEC_PUSH(*res, x); } else rc = MRP_ERR_ARG;
            } else if (sub.n == 1 && sub.a[0]->n == 1) {
                VEC_PUSH(*res, sub.a[0]->a[0]);
            } else {
                rc = mrp_set_error(MRP_ERR_ARG, "overlap component with %lld tiling paths", (long long) sub.n);
            }
            for (int64_t t = 0; t < sub.n; t++) { free(sub.a[t]->a); free(sub.a[t]); }
            free(sub.a);
        }
        free(comp->members.a); free(comp);
    }
    free(comps.a);
    if (rc == MRP_OK) rc = sweep_many(w, crossed.a, crossed.n, params);       /* coordination.c:312 */
    for (int64_t i = 0; rc == MRP_OK && i < crossed.n; i++) rc = mrp_hmm_prune(crossed.a[i], params); /* :313 */
    free(crossed.a);
    if (rc == MRP_OK) sort_hmms(w, res->a, res->n);                           /* :336 */
    *out = res;
    return rc;
}

static void free_path(hmm_vec *tp, int destroy_hmms) {
    if (!tp) return;
    if (destroy_hmms) for (int64_t i = 0; i < tp->n; i++) mrp_hmm_destroy(tp->a[i]);
    free(tp->a); free(tp);
}

/* mergeTilingPaths coordination.c:341-409 */
static int merge_tiling_paths(world *w, hmm_vec **paths, int64_t n, const mrp_params *params, hmm_vec **out) {
    if (n == 0) { *out = xcalloc(1, sizeof(hmm_vec)); return MRP_OK; }
    if (n == 1) { *out = paths[0]; return MRP_OK; }
    hmm_vec *tp1 = NULL, *tp2 = NULL;
    int rc = MRP_OK;
    if (n > 2) {
        rc = merge_tiling_paths(w, paths, n / 2, params, &tp1);
        if (rc == MRP_OK) rc = merge_tiling_paths(w, paths + n / 2, n - n / 2, params, &tp2);
        else for (int64_t i = n / 2; i < n; i++) free_path(paths[i], 1);
        if (rc != MRP_OK) { free_path(tp1, 1); free_path(tp2, 1); *out = NULL; return rc; }
    } else {
        tp1 = paths[0]; tp2 = paths[1];
    }
    return merge_two_tiling_paths(w, tp1, tp2, params, out);
}

static path_vec tiling_paths2(const world *w, const int32_t *read_index, int64_t n) { /* coordination.c:224-242 */
    mrp_hmm **hmms = xmalloc(sizeof(*hmms) * (size_t) (n + 1));
    for (int64_t i = 0; i < n; i++) hmms[i] = hmm_from_read(w, read_index[i]);
    path_vec paths = tiling_paths_from(w, hmms, n);
    free(hmms);
    return paths;
}

static int get_rp_hmms(world *w, const int32_t *read_index, int64_t n, const mrp_params *params, hmm_vec **out) {
    path_vec paths = tiling_paths2(w, read_index, n); /* coordination.c:498 */
    if (paths.n > MRP_MAX_READ_PARTITIONING_DEPTH || paths.n > params->max_coverage_depth) { /* :500-504 */
        for (int64_t i = 0; i < paths.n; i++) free_path(paths.a[i], 1);
        const int64_t np = paths.n;
        free(paths.a);
        *out = NULL;
        return mrp_set_error(MRP_ERR_ARG,
                             "Coverage depth: read depth of %lld exceeds hard maximum of %d with configured maximum of %lld",
                             (long long) np, MRP_MAX_READ_PARTITIONING_DEPTH, (long long) params->max_coverage_depth);
    }
    int rc = merge_tiling_paths(w, paths.a, paths.n, params, out);
    free(paths.a);
    return rc;
}

static int check_reads(const world *w, const mrp_read *reads, int64_t n) {
    for (int64_t i = 0; i < n; i++) {
        const mrp_read *r = &reads[i];
        if (!r->name || r->length < 1 || r->ref_start < 0 || (int64_t) r->ref_start + r->length > w->ch.n_sites)
            return mrp_set_error(MRP_ERR_ARG, "read %lld: bad interval [%d,+%d)", (long long) i, r->ref_start, r->length);
        const int64_t nb = w->ch.allele_offset[r->ref_start + r->length] - w->ch.allele_offset[r->ref_start];
        if (r->pool_offset < 0 || r->pool_offset + nb > w->ch.pool_bytes)
            return mrp_set_error(MRP_ERR_ARG, "read %lld: profile bytes outside the pool", (long long) i);
    }
    return MRP_OK;
}
static int world_init(world *w, mrp_context *ctx, const mrp_chunk *chunk, const mrp_read *reads, int64_t n_reads,
                      mrp_batch *record) {
    if (!ctx || !chunk || (n_reads > 0 && !reads)) return mrp_set_error(MRP_ERR_ARG, "NULL argument");
    if (mrp_context_device(mrp_chunk_context(chunk)) != mrp_context_device(ctx)) return mrp_set_error(MRP_ERR_ARG, "chunk lives on a different device");
    memset(w, 0, sizeof(*w));
    w->chunk = chunk; w->reads = reads; w->n_reads = n_reads; w->ctx = ctx; w->record = record;
    mrp_chunk_host_view(chunk, &w->ch);
    w->max_alleles = 1;
    for (int64_t i = 0; i < w->ch.n_sites; i++) if (w->ch.allele_number[i] > w->max_alleles) w->max_alleles = w->ch.allele_number[i];
    return check_reads(w, reads, n_reads);
}

int mrp_get_rp_hmms(mrp_context *ctx, const mrp_chunk *chunk, const mrp_read *reads, const int32_t *read_index,
                    int64_t n, const mrp_params *params, mrp_batch *record, mrp_hmm ***hmms_out, int64_t *n_out) {
    if (!params || !hmms_out || !n_out || n < 0 || (n > 0 && !read_index)) return mrp_set_error(MRP_ERR_ARG, "mrp_get_rp_hmms: bad arguments");
    int64_t max_idx = -1;
    for (int64_t i = 0; i < n; i++) { if (read_index[i] < 0) return mrp_set_error(MRP_ERR_ARG, "negative read index"); if (read_index[i] > max_idx) max_idx = read_index[i]; }
    world w;
    int rc = world_init(&w, ctx, chunk, reads, max_idx + 1, record);
    if (rc != MRP_OK) return rc;
    hmm_vec *tp = NULL;
    rc = get_rp_hmms(&w, read_index, n, params, &tp);
    if (rc != MRP_OK) { free_path(tp, 1); return rc; }
    *n_out = tp->n;
    *hmms_out = tp->a ? tp->a : xmalloc(sizeof(mrp_hmm *));
    free(tp);
    return MRP_OK;
}

int mrp_hmm_view(const mrp_hmm *hmm, mrp_hmm_job *view, const int32_t **col_reads_out, int32_t *ref_start,
                 int32_t *ref_length) {
    if (!hmm || !view) return mrp_set_error(MRP_ERR_ARG, "mrp_hmm_view: NULL argument");
    world w; memset(&w, 0, sizeof(w));
    hmm_job(&w, (mrp_hmm *) hmm, 0, view, hmm->has_results);
    if (col_reads_out) *col_reads_out = hmm->col_reads.a;
    if (ref_start) *ref_start = hmm->ref_start;
    if (ref_length) *ref_length = hmm->ref_length;
    return MRP_OK;
}

int mrp_hmm_forward_backward(mrp_context *ctx, const mrp_chunk *chunk, mrp_hmm *hmm, const mrp_params *params,
                             mrp_batch *record) {
    if (!hmm || !params) return mrp_set_error(MRP_ERR_ARG, "mrp_hmm_forward_backward: NULL argument");
    world w;
    int rc = world_init(&w, ctx, chunk, NULL, 0, record);
    if (rc != MRP_OK) return rc;
    return sweep_many(&w, &hmm, 1, params);
}

/* stRPHmm_forwardTraceBack hmm.c:165-219 */
int mrp_hmm_forward_trace_back(const mrp_hmm *h, int32_t *path) {
    if (!h || !path) return mrp_set_error(MRP_ERR_ARG, "mrp_hmm_forward_trace_back: NULL argument");
    if (!h->has_results) return mrp_set_error(MRP_ERR_ARG, "trace back before a forward/backward sweep");
    const int64_t K = hmm_K(h);
    int64_t c0 = h->cell_off.a[K - 1], nc = h->cell_off.a[K] - c0;
    int64_t best = 0;
    double max_prob = h->f[c0];
    for (int64_t c = 1; c < nc; c++) if (h->f[c0 + c] > max_prob) { max_prob = h->f[c0 + c]; best = c; }
    path[K - 1] = (int32_t) best;
    for (int64_t k = K - 1; k > 0; k--) {
        const uint32_t m = h->prev.a[h->cell_off.a[k] + path[k]];
        c0 = h->cell_off.a[k - 1]; nc = h->cell_off.a[k] - c0;
        best = -1; max_prob = -INFINITY;
        for (int64_t c = 0; c < nc; c++)
            if (h->next.a[c0 + c] == m && h->f[c0 + c] > max_prob) { max_prob = h->f[c0 + c]; best = c; }
        if (best < 0) return mrp_set_error(MRP_ERR_LOOKUP, "trace back: no cell feeds the chosen merge cell in column %lld", (long long) (k - 1));
        path[k - 1] = (int32_t) best;
    }
    return MRP_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* split (hmm.c:1192-1383)                                                                      */
/* ------------------------------------------------------------------------------------------ */
/* columns [k0, k1) of src appended to dst; the first one may start later, the last one end earlier (stRPColumn_split
 * column.c:86-101 leaves both halves with the same cells and reads) */
static void hmm_append_columns(const world *w, mrp_hmm *dst, const mrp_hmm *src, int64_t k0, int64_t k1, int32_t first_start,
                               int32_t last_end) {
    for (int64_t k = k0; k < k1; k++) {
        int32_t start = src->col_start.a[k], end = start + src->col_len.a[k];
        if (k == k0 && first_start > start) start = first_start;
        if (k == k1 - 1 && last_end < end) end = last_end;
        hmm_begin_column(dst, w, start, end - start, src->col_depth.a[k], src->col_reads.a + src->read_off.a[k]);
        for (int64_t c = src->cell_off.a[k]; c < src->cell_off.a[k + 1]; c++) {
            hmm_add_cell(dst, src->part.a[c], k == k0 ? 0u : src->prev.a[c]);
            dst->next.a[dst->next.n - 1] = k == k1 - 1 ? 0u : src->next.a[c];
        }
        hmm_end_column(dst);
        if (k + 1 < k1) {
            hmm_begin_merge(dst, src->mask_from.a[k], src->mask_to.a[k]);
            for (int64_t m = src->mcell_off.a[k]; m < src->mcell_off.a[k + 1]; m++) { VEC_PUSH(dst->mfrom, src->mfrom.a[m]); VEC_PUSH(dst->mto, src->mto.a[m]); }
            hmm_end_merge(dst);
        }
    }
}
/* h takes the arrays of `from` (which is consumed); h's own bookkeeping as an allocation stays */
static void hmm_take(mrp_hmm *h, mrp_hmm *from) {
    void *arrays[] = {h->reads.a, h->col_start.a, h->col_len.a, h->col_depth.a, h->cell_off.a, h->read_off.a, h->col_reads.a,
                      h->read_byte_off.a, h->part.a, h->next.a, h->prev.a, h->mask_from.a, h->mask_to.a, h->mcell_off.a,
                      h->mfrom.a, h->mto.a};
    for (size_t i = 0; i < sizeof(arrays) / sizeof(arrays[0]); i++) hmm_free_array(h, arrays[i]);
    hmm_free_results(h);
    h->ref_start = from->ref_start; h->ref_length = from->ref_length; h->max_depth = from->max_depth;
    h->reads = from->reads; h->col_start = from->col_start; h->col_len = from->col_len; h->col_depth = from->col_depth;
    h->cell_off = from->cell_off; h->read_off = from->read_off; h->col_reads = from->col_reads; h->read_byte_off = from->read_byte_off;
    h->part = from->part; h->next = from->next; h->prev = from->prev; h->mask_from = from->mask_from; h->mask_to = from->mask_to;
    h->mcell_off = from->mcell_off; h->mfrom = from->mfrom; h->mto = from->mto;
    free(from);
}
/* stRPHmm_split hmm.c:1231-1300: h keeps [refStart, split_point), the returned hmm holds the rest.  The column that
 * contains the split point is cut in two (both halves keep its cells); the merge column in front of the suffix goes. */
static mrp_hmm *hmm_split(const world *w, mrp_hmm *h, int32_t sp) {
    const int64_t K = hmm_K(h);
    int64_t ks = 0; /* getColumn :1192-1209 */
    while (ks < K && sp >= h->col_start.a[ks] + h->col_len.a[ks]) ks++;
    const int inside = sp > h->col_start.a[ks];
    mrp_hmm *L = hmm_new(), *R = hmm_new();
    for (int64_t i = 0; i < h->reads.n; i++) { /* :1247-1262 */
        const mrp_read *r = &w->reads[h->reads.a[i]];
        if (r->ref_start < sp) VEC_PUSH(L->reads, h->reads.a[i]);
        if (r->ref_start + r->length > sp) VEC_PUSH(R->reads, h->reads.a[i]);
    }
    hmm_append_columns(w, L, h, 0, inside ? ks + 1 : ks, h->col_start.a[0], sp);
    hmm_append_columns(w, R, h, ks, K, sp, h->col_start.a[K - 1] + h->col_len.a[K - 1]);
    L->ref_start = h->ref_start; L->ref_length = sp - h->ref_start;
    R->ref_start = sp; R->ref_length = h->ref_start + h->ref_length - sp;
    hmm_take(h, L);
    return R;
}
static int world_host(world *w, const mrp_chunk *chunk, const mrp_read *reads, int64_t n_reads) {
    if (!chunk || (n_reads > 0 && !reads)) return mrp_set_error(MRP_ERR_ARG, "NULL argument");
    memset(w, 0, sizeof(*w));
    w->chunk = chunk; w->reads = reads; w->n_reads = n_reads;
    mrp_chunk_host_view(chunk, &w->ch);
    return check_reads(w, reads, n_reads);
}
static int hmm_reads_known(const mrp_hmm *h, int64_t n_reads) {
    for (int64_t i = 0; i < h->reads.n; i++) if (h->reads.a[i] < 0 || h->reads.a[i] >= n_reads) return 0;
    return 1;
}
int mrp_hmm_split(const mrp_chunk *chunk, const mrp_read *reads, int64_t n_reads, mrp_hmm *hmm, int32_t split_point,
                  mrp_hmm **suffix_out) {
    if (!hmm || !suffix_out) return mrp_set_error(MRP_ERR_ARG, "mrp_hmm_split: bad arguments");
    if (split_point <= hmm->ref_start) return mrp_set_error(MRP_ERR_ARG, "The split point is at or before the start of the reference interval");
    if (split_point >= hmm->ref_start + hmm->ref_length) return mrp_set_error(MRP_ERR_ARG, "The split point is after the last position of the reference interval");
    world w;
    int rc = world_host(&w, chunk, reads, n_reads);
    if (rc != MRP_OK) return rc;
    if (!hmm_reads_known(hmm, n_reads)) return mrp_set_error(MRP_ERR_ARG, "mrp_hmm_split: the hmm names reads beyond n_reads");
    *suffix_out = hmm_split(&w, hmm, split_point);
    return MRP_OK;
}

static void genome_fragment(const world *w, mrp_phase_result *g, const mrp_hmm *h, const uint64_t *chosen, int64_t max_iterations);
static mrp_phase_result *result_new(int32_t ref_start, int32_t length, int64_t n_reads);
/* sitesLinkageIsWellSupported hmm.c:1302-1320: reads shared by the columns that hold the two sites */
static int sites_linkage_well_supported(const mrp_hmm *h, const mrp_params *params, int32_t left, int32_t right) {
    const int64_t K = hmm_K(h);
    int64_t kl = 0, kr;
    while (kl < K - 1 && left >= h->col_start.a[kl] + h->col_len.a[kl]) kl++;
    kr = kl;
    while (kr < K - 1 && right >= h->col_start.a[kr] + h->col_len.a[kr]) kr++;
    const int32_t *a = h->col_reads.a + h->read_off.a[kl], *b = h->col_reads.a + h->read_off.a[kr];
    int64_t common = 0;
    for (int32_t i = 0; i < h->col_depth.a[kl]; i++)
        for (int32_t j = 0; j < h->col_depth.a[kr]; j++)
            if (a[i] == b[j]) { common++; break; }
    return common >= params->min_read_coverage_to_support_phasing_between_heterozygous_sites;
}
/* stRPHMM_splitWherePhasingIsUncertain hmm.c:1322-1383: sweep, trace back, predicted haplotypes; between two consecutive
 * heterozygous sites that too few reads span, the hmm is cut half way.  The input hmm becomes the first of the list. */
int mrp_hmm_split_where_phasing_is_uncertain(mrp_context *ctx, const mrp_chunk *chunk, const mrp_read *reads, int64_t n_reads,
                                             mrp_hmm *hmm, const mrp_params *params, mrp_hmm ***hmms_out, int64_t *n_out) {
    if (!hmm || !params || !hmms_out || !n_out) return mrp_set_error(MRP_ERR_ARG, "mrp_hmm_split_where_phasing_is_uncertain: bad arguments");
    world w;
    int rc = world_init(&w, ctx, chunk, reads, n_reads, NULL);
    if (rc != MRP_OK) return rc;
    if (!hmm_reads_known(hmm, n_reads)) return mrp_set_error(MRP_ERR_ARG, "the hmm names reads beyond n_reads");
    mrp_hmm *one = hmm;
    rc = sweep_many(&w, &one, 1, params);
    if (rc != MRP_OK) return rc;
    const int64_t K = hmm_K(hmm);
    int32_t *path = xmalloc(sizeof(int32_t) * (size_t) K);
    rc = mrp_hmm_forward_trace_back(hmm, path);
    if (rc != MRP_OK) { free(path); return rc; }
    uint64_t *chosen = xmalloc(sizeof(uint64_t) * (size_t) K);
    for (int64_t k = 0; k < K; k++) chosen[k] = hmm->part.a[hmm->cell_off.a[k] + path[k]];
    mrp_phase_result *g = result_new(hmm->ref_start, hmm->ref_length, n_reads);
    genome_fragment(&w, g, hmm, chosen, 0); /* stGenomeFragment_construct only, :1330 */
    hmm_vec out = {0};
    int32_t prev_het = -1;
    for (int32_t i = 0; i < g->length; i++) {
        if (g->haplotype_string1[i] == g->haplotype_string2[i]) continue;
        const int32_t site = g->ref_start + i;
        if (prev_het >= 0 && !sites_linkage_well_supported(hmm, params, prev_het, site)) {
            mrp_hmm *right = hmm_split(&w, hmm, prev_het + (site - prev_het + 1) / 2); /* :1361 */
            VEC_PUSH(out, hmm);
            hmm = right;
        }
        prev_het = site;
    }
    VEC_PUSH(out, hmm);
    free(path); free(chosen); mrp_phase_result_destroy(g);
    *hmms_out = out.a;
    *n_out = out.n;
    return MRP_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* genome fragment (emissions.c:246-343, genomeFragment.c)                                     */
/* ------------------------------------------------------------------------------------------ */
static mrp_phase_result *result_new(int32_t ref_start, int32_t length, int64_t n_reads) {
    mrp_phase_result *r = xcalloc(1, sizeof(*r));
    r->ref_start = ref_start; r->length = length;
    const size_t n = (size_t) length;
    r->genotype_string = xcalloc(n, sizeof(uint64_t)); r->haplotype_string1 = xcalloc(n, sizeof(uint64_t));
    r->haplotype_string2 = xcalloc(n, sizeof(uint64_t)); r->ancestor_string = xcalloc(n, sizeof(uint64_t));
    r->reads_supporting_haplotype1 = xcalloc(n, sizeof(uint64_t)); r->reads_supporting_haplotype2 = xcalloc(n, sizeof(uint64_t));
    r->genotype_probs = xcalloc(n, sizeof(float)); r->haplotype_probs1 = xcalloc(n, sizeof(float));
    r->haplotype_probs2 = xcalloc(n, sizeof(float));
    /* (a read that inconsistent columns put on both sides sits in both lists and may be moved into a list that already holds it) */
    r->reads1 = xcalloc(2 * (size_t) n_reads + 2, sizeof(int32_t)); r->reads2 = xcalloc(2 * (size_t) n_reads + 2, sizeof(int32_t));
    return r;
}
void mrp_phase_result_destroy(mrp_phase_result *r) {
    if (!r) return;
    free(r->genotype_string); free(r->haplotype_string1); free(r->haplotype_string2); free(r->ancestor_string);
    free(r->reads_supporting_haplotype1); free(r->reads_supporting_haplotype2); free(r->genotype_probs);
    free(r->haplotype_probs1); free(r->haplotype_probs2); free(r->reads1); free(r->reads2);
    free(r);
}
/* fillInPredictedGenome emissions.c:323-343 for column k with the given partition.  The allele
 * costs are the same integers getLogProbOfAllele returns (sum of the bytes of the reads in the
 * partition), summed directly. */
static void fill_in_predicted_genome(const world *w, mrp_phase_result *g, const mrp_hmm *h, int64_t k, uint64_t partition, uint64_t *scratch) {
    const int32_t depth = h->col_depth.a[k];
    const int64_t *off = h->read_byte_off.a + h->read_off.a[k];
    const int32_t start = h->col_start.a[k];
    const uint32_t first_allele = w->ch.allele_offset[start];
    uint64_t *h1 = scratch, *h2 = h1 + w->max_alleles, *a1 = h2 + w->max_alleles, *a2 = a1 + w->max_alleles; /* [4 * max_alleles] */
    for (int32_t s = 0; s < h->col_len.a[k]; s++) {
        const int32_t site = start + s;
        const uint32_t A = w->ch.allele_number[site], so = w->ch.allele_offset[site] - first_allele;
        const uint16_t *sub = w->ch.sub + w->ch.sub_offset[site], *prior = w->ch.prior + w->ch.allele_offset[site];
        if (A == 2) { /* the common case: both bytes of a read together, no branch on the partition bit */
            uint64_t t0 = 0, t1 = 0, x0 = 0, x1 = 0; /* totals over the column's reads, and over those in the partition */
            for (int32_t i = 0; i < depth; i++) {
                const uint8_t *b = w->ch.pool + off[i] + so;
                const uint64_t in = 0 - ((partition >> i) & 1);
                t0 += b[0]; t1 += b[1];
                x0 += b[0] & in; x1 += b[1] & in;
            }
            h1[0] = x0; h1[1] = x1; h2[0] = t0 - x0; h2[1] = t1 - x1;
        } else {
            for (uint32_t a = 0; a < A; a++) { h1[a] = 0; h2[a] = 0; }
            for (int32_t i = 0; i < depth; i++) {
                const uint8_t *b = w->ch.pool + off[i] + so;
                uint64_t *dst = ((partition >> i) & 1) ? h1 : h2;
                for (uint32_t a = 0; a < A; a++) dst[a] += b[a];
            }
        }
        for (uint32_t i = 0; i < A; i++) { /* ancestorHapProbabilities emissions.c:156-172 */
            uint64_t x = h1[0] + sub[i * A], y = h2[0] + sub[i * A];
            for (uint32_t q = 1; q < A; q++) {
                if (h1[q] + sub[i * A + q] < x) x = h1[q] + sub[i * A + q];
                if (h2[q] + sub[i * A + q] < y) y = h2[q] + sub[i * A + q];
            }
            a1[i] = x; a2[i] = y;
        }
        uint64_t best = a1[0] + a2[0] + prior[0], anc = 0; /* :283-292 */
        for (uint32_t i = 1; i < A; i++) {
            const uint64_t j = a1[i] + a2[i] + prior[i];
            if (j < best) { best = j; anc = i; }
        }
        uint64_t hap1 = 0, hap2 = 0, m1 = h1[0] + sub[anc * A], m2 = h2[0] + sub[anc * A]; /* getMLAllele :246-261 */
        for (uint32_t i = 1; i < A; i++) {
            if (h1[i] + sub[anc * A + i] < m1) { m1 = h1[i] + sub[anc * A + i]; hap1 = i; }
            if (h2[i] + sub[anc * A + i] < m2) { m2 = h2[i] + sub[anc * A + i]; hap2 = i; }
        }
        const int64_t q = site - g->ref_start;
        g->ancestor_string[q] = anc;
        g->haplotype_string1[q] = hap1;
        g->haplotype_string2[q] = hap2;
        g->genotype_string[q] = hap1 < hap2 ? hap1 * A + hap2 : hap2 * A + hap1;
        g->genotype_probs[q] = -((float) best);
        g->haplotype_probs1[q] = -(float) h1[hap1];
        g->haplotype_probs2[q] = -(float) h2[hap2];
        g->reads_supporting_haplotype1[q] = (uint64_t) __builtin_popcountll(partition);
        g->reads_supporting_haplotype2[q] = (uint64_t) depth - (uint64_t) __builtin_popcountll(partition);
    }
}
/* getLogProbOfReadGivenHaplotype genomeFragment.c:71-89, for both haplotypes in one walk over the read's sites: *x for hap1,
 * *y for hap2 (each sum in site order, then divided by PROFILE_PROB_SCALAR inc/margin.h:189) */
static void read_log_prob2(const world *w, const uint64_t *hap1, const uint64_t *hap2, int32_t start, int32_t length, int32_t read, double *x, double *y) {
    const mrp_read *r = &w->reads[read];
    double t1 = 0.0, t2 = 0.0;
    const uint32_t first = w->ch.allele_offset[r->ref_start];
    int32_t lo = start - r->ref_start, hi = start + length - r->ref_start;
    if (lo < 0) lo = 0;
    if (hi > r->length) hi = r->length;
    const uint8_t *pool = w->ch.pool + r->pool_offset;
    const uint32_t *ao = w->ch.allele_offset + r->ref_start;
    const uint64_t *a1 = hap1 + (r->ref_start - start), *a2 = hap2 + (r->ref_start - start);
    for (int32_t i = lo; i < hi; i++) {
        const uint32_t o = ao[i] - first;
        t1 -= pool[o + a1[i]];
        t2 -= pool[o + a2[i]];
    }
    *x = t1 / 30.0; *y = t2 / 30.0;
}

/* stGenomeFragment_construct genomeFragment.c:40-69 (+ hmm.c:221-248) then
 * stGenomeFragment_refineGenomeFragment genomeFragment.c:165-232 */
static void genome_fragment(const world *w, mrp_phase_result *g, const mrp_hmm *h, const uint64_t *chosen,
                            int64_t max_iterations) {
    const int64_t K = hmm_K(h);
    /* side[read]: 0 = unseen, 1 = reads1, 2 = reads2; first sighting along the path wins per set
     * (a read can be put in both sets by inconsistent columns; set semantics as in the reference) */
    uint8_t *in1 = xcalloc((size_t) w->n_reads + 1, 1), *in2 = xcalloc((size_t) w->n_reads + 1, 1);
    uint64_t *p = xmalloc(sizeof(uint64_t) * (size_t) K);
    uint64_t *scratch = xmalloc(sizeof(uint64_t) * 4 * (size_t) w->max_alleles);
    for (int64_t k = 0; k < K; k++) {
        p[k] = chosen[k]; /* partition of the traced-back cell of column k */
        const int32_t *cr = h->col_reads.a + h->read_off.a[k];
        for (int32_t i = 0; i < h->col_depth.a[k]; i++) {
            if ((p[k] >> i) & 1) { if (!in1[cr[i]]) { in1[cr[i]] = 1; g->reads1[g->n_reads1++] = cr[i]; } }
            else { if (!in2[cr[i]]) { in2[cr[i]] = 1; g->reads2[g->n_reads2++] = cr[i]; } }
        }
        fill_in_predicted_genome(w, g, h, k, p[k], scratch);
    }
    int64_t iteration = 0;
    uint8_t *m12 = xcalloc((size_t) w->n_reads + 1, 1), *m21 = xcalloc((size_t) w->n_reads + 1, 1);
    int32_t *n1 = xmalloc(sizeof(int32_t) * (size_t) (2 * w->n_reads + 2)), *n2 = xmalloc(sizeof(int32_t) * (size_t) (2 * w->n_reads + 2));
    while (iteration++ < max_iterations) {
        int64_t c12 = 0, c21 = 0;
        memset(m12, 0, (size_t) w->n_reads + 1); memset(m21, 0, (size_t) w->n_reads + 1);
        for (int64_t i = 0; i < g->n_reads1; i++) { /* :126-151 */
            const int32_t r = g->reads1[i];
            double x, y;
            read_log_prob2(w, g->haplotype_string1, g->haplotype_string2, g->ref_start, g->length, r, &x, &y);
            if (x < y) { m12[r] = 1; c12++; }
        }
        for (int64_t i = 0; i < g->n_reads2; i++) {
            const int32_t r = g->reads2[i];
            double x, y;
            read_log_prob2(w, g->haplotype_string1, g->haplotype_string2, g->ref_start, g->length, r, &x, &y);
            if (y < x) { m21[r] = 1; c21++; }
        }
        if (c12 + c21 == 0) break;
        int64_t a = 0, b = 0;
        for (int64_t i = 0; i < g->n_reads1; i++) if (!m12[g->reads1[i]]) n1[a++] = g->reads1[i];
        for (int64_t i = 0; i < g->n_reads2; i++) if (!m21[g->reads2[i]]) n2[b++] = g->reads2[i];
        for (int64_t i = 0; i < g->n_reads2; i++) if (m21[g->reads2[i]]) n1[a++] = g->reads2[i];
        for (int64_t i = 0; i < g->n_reads1; i++) if (m12[g->reads1[i]]) n2[b++] = g->reads1[i];
        memcpy(g->reads1, n1, sizeof(int32_t) * (size_t) a); memcpy(g->reads2, n2, sizeof(int32_t) * (size_t) b);
        g->n_reads1 = a; g->n_reads2 = b;
        for (int64_t k = 0; k < K; k++) { /* :211-226 */
            const int32_t *cr = h->col_reads.a + h->read_off.a[k];
            uint64_t flip = 0; /* (a read moved both ways -- it sat in both lists -- is flipped twice: not at all) */
            for (int32_t i = 0; i < h->col_depth.a[k]; i++) flip |= (uint64_t) (m12[cr[i]] ^ m21[cr[i]]) << i;
            if (!flip) continue; /* fillInPredictedGenome is a function of the column and its partition: unchanged */
            p[k] ^= flip;
            fill_in_predicted_genome(w, g, h, k, p[k], scratch);
        }
    }
    free(in1); free(in2); free(p); free(m12); free(m21); free(n1); free(n2); free(scratch);
}

/* filterReadsByCoverageDepth coordination.c:443-488 */
static void filter_reads_by_coverage_depth(const world *w, const mrp_params *params, int32_t *filtered, int64_t *nf,
                                           int32_t *discarded, int64_t *nd) {
    int32_t *all = xmalloc(sizeof(int32_t) * (size_t) (w->n_reads + 1));
    for (int64_t i = 0; i < w->n_reads; i++) all[i] = (int32_t) i;
    path_vec paths = tiling_paths2(w, all, w->n_reads);
    free(all);
    keyed *a = xmalloc(sizeof(keyed) * (size_t) (paths.n + 1)), *t = xmalloc(sizeof(keyed) * (size_t) (paths.n + 1));
    for (int64_t i = 0; i < paths.n; i++) {
        int64_t total = 0;
        for (int64_t j = 0; j < paths.a[i]->n; j++) total += w->reads[paths.a[i]->a[j]->reads.a[0]].length;
        a[i].idx = i; a[i].key = (double) total;
    }
    keyed_sort_desc(a, paths.n, t);
    int64_t np = paths.n;
    *nf = 0; *nd = 0;
    while (np > params->max_coverage_depth) {
        hmm_vec *tp = paths.a[a[--np].idx];
        for (int64_t j = tp->n - 1; j >= 0; j--) discarded[(*nd)++] = tp->a[j]->reads.a[0];
    }
    while (np > 0) {
        hmm_vec *tp = paths.a[a[--np].idx];
        for (int64_t j = tp->n - 1; j >= 0; j--) filtered[(*nf)++] = tp->a[j]->reads.a[0];
    }
    for (int64_t i = 0; i < paths.n; i++) free_path(paths.a[i], 1);
    free(paths.a); free(a); free(t);
}

/* bubbleGraph.c:2761-2779: genome fragment from the traced-back partitions, refinement, re-adding the filtered reads */
static void finish_phase_parts(world *w, const mrp_hmm *hmm, const uint64_t *chosen, double fwd, double bwd, const mrp_params *params,
                               const int32_t *discarded, int64_t nd, mrp_phase_result **out) {
    mrp_phase_result *g = result_new(hmm->ref_start, hmm->ref_length, w->n_reads);
    genome_fragment(w, g, hmm, chosen, params->rounds_of_iterative_refinement); /* :2761-2764 */
    for (int64_t i = 0; i < nd; i++) { /* :2772-2779 */
        double x, y;
        read_log_prob2(w, g->haplotype_string1, g->haplotype_string2, g->ref_start, g->length, discarded[i], &x, &y);
        if (x < y) g->reads2[g->n_reads2++] = discarded[i]; else g->reads1[g->n_reads1++] = discarded[i];
    }
    g->hmm_forward = fwd; g->hmm_backward = bwd; g->n_sweeps = w->n_sweeps;
    *out = g;
}
/* bubbleGraph.c:2755-2779 on a swept host hmm */
static int finish_phase(world *w, mrp_hmm *hmm, const mrp_params *params, const int32_t *discarded, int64_t nd,
                        mrp_phase_result **out) {
    const int64_t K = hmm_K(hmm);
    int32_t *path = xmalloc(sizeof(int32_t) * (size_t) K);
    int rc = mrp_hmm_forward_trace_back(hmm, path); /* :2755 */
    if (rc == MRP_OK) {
        uint64_t *chosen = xmalloc(sizeof(uint64_t) * (size_t) K);
        for (int64_t k = 0; k < K; k++) chosen[k] = hmm->part.a[hmm->cell_off.a[k] + path[k]];
        finish_phase_parts(w, hmm, chosen, hmm->fwd, hmm->bwd, params, discarded, nd, out);
        free(chosen);
    }
    free(path);
    return rc;
}

/* bubbleGraph_phaseBubbleGraph bubbleGraph.c:2673-2801 */
int mrp_phase_reads(mrp_context *ctx, const mrp_chunk *chunk, const mrp_read *reads, int64_t n_reads,
                    const mrp_params *params, mrp_batch *record, mrp_phase_result **out) {
    if (!params || !out || n_reads < 0) return mrp_set_error(MRP_ERR_ARG, "mrp_phase_reads: bad arguments");
    *out = NULL;
    world w;
    int rc = world_init(&w, ctx, chunk, reads, n_reads, record);
    if (rc != MRP_OK) return rc;
    if (n_reads == 0) { *out = result_new(0, 0, 0); return MRP_OK; } /* :2719-2728 */
    int32_t *filtered = xmalloc(sizeof(int32_t) * (size_t) n_reads), *discarded = xmalloc(sizeof(int32_t) * (size_t) n_reads);
    int64_t nf, nd;
    filter_reads_by_coverage_depth(&w, params, filtered, &nf, discarded, &nd); /* :2699 */
    uint8_t *is_disc = xcalloc((size_t) n_reads, 1);
    for (int64_t i = 0; i < nd; i++) is_disc[discarded[i]] = 1;
    int32_t *fwd = xmalloc(sizeof(int32_t) * (size_t) n_reads), *rev = xmalloc(sizeof(int32_t) * (size_t) n_reads);
    int64_t nfwd = 0, nrev = 0;
    for (int64_t i = 0; i < n_reads; i++) { /* :2705-2716 */
        if (is_disc[i]) continue;
        if (reads[i].forward_strand) fwd[nfwd++] = (int32_t) i; else rev[nrev++] = (int32_t) i;
    }
    mrp_params pc = *params;
    pc.include_ancestor_sub_prob = 0; /* :2733 */
    hmm_vec *tpF = NULL, *tpR = NULL, *joined = NULL;
    mrp_hmm *hmm = NULL;
    rc = get_rp_hmms(&w, fwd, nfwd, &pc, &tpF);                  /* :2736 */
    if (rc == MRP_OK) rc = get_rp_hmms(&w, rev, nrev, &pc, &tpR); /* :2740 */
    if (rc == MRP_OK) { rc = merge_two_tiling_paths(&w, tpF, tpR, &pc, &joined); tpF = tpR = NULL; } /* :2745 */
    if (rc == MRP_OK && joined->n > 0) {
        hmm = fuse_path(&w, joined);
        free(joined->a); free(joined); joined = NULL;
        pc.include_ancestor_sub_prob = 1; /* :2748 */
        rc = sweep_many(&w, &hmm, 1, &pc); /* :2749 */
        if (rc == MRP_OK) rc = finish_phase(&w, hmm, params, discarded, nd, out);
    } else if (rc == MRP_OK) {
        *out = result_new(0, 0, n_reads);
    }
    free_path(tpF, 1); free_path(tpR, 1); free_path(joined, 1);
    mrp_hmm_destroy(hmm);
    free(filtered); free(discarded); free(is_disc); free(fwd); free(rev);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* device-resident merge                                                     */
/*                                                                                             */
/* The same recursion as merge_tiling_paths / merge_two_tiling_paths above, but the hmms never  */
/* leave HBM.  The structural decisions depend on read intervals only and are split in two:     */
/*   host    WHICH hmms are merged -- tiling paths, overlap components (coordination.c:69-339):   */
/*           a few hundred intervals per chunk and level -- and the merged column BOUNDARIES of   */
/*           every cross product (two sorted lists per hmm: 8 bytes per column);                 */
/*   device  everything else a column needs (which parent column each side is cut from, the      */
/*           connector kinds, the column's reads and where their profile bytes start, allele      */
/*           slots): mrp_structure_kernel, one thread per column, from the parents' own column    */
/*           tables, which stay in HBM (mrp_engine.h).                                           */
/* A "shadow" is what the host keeps of an hmm: interval, reads, column boundaries, and WHERE its */
/* pruned form will be on the device (segment, first column).  Cross product, sweep and prune    */
/* of every overlap component of a recursion level -- of every chunk and strand handed in --     */
/* run as ONE batch of kernels (mrp_engine_level_*).                                            */
/* ------------------------------------------------------------------------------------------ */
typedef struct rhmm {
    int32_t ref_start, ref_length;
    int32_t first_read;        /* reads[0]: its name breaks ties in stRPHmm_cmpFn */
    int32_t n_cols, n_reads;
    int32_t seg;               /* segment of the engine that holds the pruned hmm; -1: stRPHmm_construct leaf */
    int64_t col0;              /* first column in the segment; leaf: offset of the read's profile bytes in the pool */
    int32_t *starts;           /* [n_cols] first site of every column */
    int32_t *roff;             /* [n_cols + 1] prefix sums of the column depths */
    int32_t *reads;            /* [n_reads] stRPHmm.profileSeqs; a column's reads are those of them that cover it, in this order */
    mrp_xpar *par;             /* [n_a + n_b] the two tiling paths it is the cross product of */
    int32_t n_a, n_b;
    int64_t bound_cells, bound_merge, depth_sites; /* static bounds, see mrp_xhmm */
    int32_t bound_max_cells, bound_max_merge;
    int32_t pool_class;        /* size class of the block in the shadow pool + 1; 0: lives in a block owned by someone else */
} rhmm;

#define HMM_T rhmm
#define PFX(x) r_##x
#define H_NAME_READ(h) ((h)->first_read)
#include "rphmm_paths.inc"
#undef HMM_T
#undef PFX
#undef H_NAME_READ

static void rhmm_destroy(rhmm *h) {
    if (h && h->pool_class != 0) shadow_release(h, h->pool_class - 1);
}
/* A tiling path is either a vector of its own (heap) or the head of ONE block of the shadow pool that also holds its array and
 * the lists of the merge that made it (level_prepare): cap = -(size class + 2) marks the latter. */
static void r_path_release(r_hmm_vec *tp) {
    if (!tp) return;
    if (tp->cap < -1) { shadow_release(tp, (int) (-tp->cap - 2)); return; }
    free(tp->a); free(tp);
}
static void r_free_path(r_hmm_vec *tp, int destroy_hmms) {
    if (!tp) return;
    if (destroy_hmms) for (int64_t i = 0; i < tp->n; i++) rhmm_destroy(tp->a[i]);
    r_path_release(tp);
}

/* stRPHmm_construct (hmm.c:97-133): one column {1, 0} over the read's sites.  All leaves of a chunk live in one block. */
typedef struct { rhmm h; int32_t starts[1], roff[2], reads[1]; } rleaf;
static void r_leaf_init(rleaf *l, const world *w, int32_t read) {
    const mrp_read *r = &w->reads[read];
    rhmm *h = &l->h;
    memset(h, 0, sizeof(*h));
    h->ref_start = r->ref_start; h->ref_length = r->length; h->first_read = read;
    h->n_cols = 1; h->n_reads = 1;
    h->seg = -1; h->col0 = r->pool_offset; /* profileSeq.c:41-47 at the read's first site */
    l->starts[0] = r->ref_start; l->roff[0] = 0; l->roff[1] = 1; l->reads[0] = read;
    h->starts = l->starts; h->roff = l->roff; h->reads = l->reads;
}
static rleaf *r_leaves_of_chunk(const world *w, int *cls) { /* (a block of the shadow pool: a few hundred KB, warm from the call before) */
    rleaf *lv = shadow_alloc(sizeof(rleaf) * (size_t) (w->n_reads + 1), cls);
    for (int64_t i = 0; i < w->n_reads; i++) r_leaf_init(&lv[i], w, (int32_t) i);
    return lv;
}

/* filterReadsByCoverageDepth coordination.c:443-488 on the chunk's leaves */
static void r_filter_reads_by_coverage_depth(const world *w, rhmm *const *sorted, const mrp_params *params, int32_t *filtered, int64_t *nf,
                                             int32_t *discarded, int64_t *nd) {
    r_path_vec paths = r_tiling_paths_sorted(sorted, w->n_reads);
    keyed *a = xmalloc(sizeof(keyed) * (size_t) (paths.n + 1)), *t = xmalloc(sizeof(keyed) * (size_t) (paths.n + 1));
    for (int64_t i = 0; i < paths.n; i++) {
        int64_t total = 0;
        for (int64_t j = 0; j < paths.a[i]->n; j++) total += paths.a[i]->a[j]->ref_length;
        a[i].idx = i; a[i].key = (double) total;
    }
    keyed_sort_desc(a, paths.n, t);
    int64_t np = paths.n;
    *nf = 0; *nd = 0;
    while (np > params->max_coverage_depth) {
        r_hmm_vec *tp = paths.a[a[--np].idx];
        for (int64_t j = tp->n - 1; j >= 0; j--) discarded[(*nd)++] = tp->a[j]->first_read;
    }
    while (np > 0) {
        r_hmm_vec *tp = paths.a[a[--np].idx];
        for (int64_t j = tp->n - 1; j >= 0; j--) filtered[(*nf)++] = tp->a[j]->first_read;
    }
    for (int64_t i = 0; i < paths.n; i++) r_free_path(paths.a[i], 0);
    free(paths.a); free(a); free(t);
}

/* The pieces of a tiling path between S and E, one after the other: a column of one of its hmms, or a gap between two of
 * them / in front of the first / behind the last (stRPHmm_fuse hmm.c:283-372 and the prefix / suffix gaps of
 * stRPHmm_alignColumns hmm.c:396-462). */
typedef struct { const r_hmm_vec *tp; int64_t i; int32_t k, pos, E; } piter;
typedef struct { int32_t end, depth; uint8_t out; } rpiece; /* out: connector that leaves the piece at its own end */
static inline __attribute__((always_inline)) int piter_next(piter *it, rpiece *p) {
    if (it->i < it->tp->n) {
        const rhmm *h = it->tp->a[it->i];
        if (it->k == 0 && h->ref_start > it->pos) { /* gap: depth 0, one cell; (0, 0) merge column behind it (hmm.c:324-345) */
            p->end = h->ref_start; p->depth = 0; p->out = MRP_CONN_ZERO;
            it->pos = p->end;
            return 1;
        }
        const int32_t k = it->k;
        p->end = k + 1 < h->n_cols ? h->starts[k + 1] : h->ref_start + h->ref_length;
        p->depth = h->roff[k + 1] - h->roff[k];
        p->out = k + 1 < h->n_cols ? MRP_CONN_REAL : MRP_CONN_ZERO;
        it->pos = p->end;
        if (++it->k == h->n_cols) { it->i++; it->k = 0; }
        return 1;
    }
    if (it->pos < it->E) {
        p->end = it->E; p->depth = 0; p->out = MRP_CONN_ZERO;
        it->pos = it->E;
        return 1;
    }
    return 0;
}

/* The shadow of stRPHmm_createCrossProductOfTwoAlignedHmm (hmm.c:534-750) of two tiling paths: both are cut at the union
 * of their column boundaries (hmm.c:476-504, column.c:70-130); per column the host keeps its first site and its depth and
 * sums up the static bounds the engine sizes its launches with.  b may be empty: stRPHmm_fuse of path a alone. */
static int r_cross_build(const world *w, const r_hmm_vec *a, const r_hmm_vec *b, int32_t S, int32_t E, int32_t stride, rhmm **out) {
    (void) w;
    int64_t cap = 2, n_reads = 0;
    /* (the parents were written a level ago, often by another core: their arrays are asked for while the sizes are added up) */
    for (int64_t i = 0; i < a->n; i++) { const rhmm *p = a->a[i]; cap += p->n_cols + 1; n_reads += p->n_reads; __builtin_prefetch(p->starts); __builtin_prefetch(p->roff); __builtin_prefetch(p->reads); }
    for (int64_t i = 0; i < b->n; i++) { const rhmm *p = b->a[i]; cap += p->n_cols + 1; n_reads += p->n_reads; __builtin_prefetch(p->starts); __builtin_prefetch(p->roff); __builtin_prefetch(p->reads); }
    const size_t o_starts = (sizeof(rhmm) + 7) & ~(size_t) 7, o_roff = o_starts + 4 * (size_t) cap, o_reads = o_roff + 4 * (size_t) (cap + 1),
                 o_par = (o_reads + 4 * (size_t) n_reads + 7) & ~(size_t) 7, bytes = o_par + sizeof(mrp_xpar) * (size_t) (a->n + b->n);
    int cls = 0;
    char *blk = shadow_alloc(bytes, &cls);
    rhmm *h = (rhmm *) blk;
    memset(h, 0, sizeof(*h));
    h->pool_class = cls + 1;
    h->starts = (int32_t *) (blk + o_starts); h->roff = (int32_t *) (blk + o_roff); h->reads = (int32_t *) (blk + o_reads);
    h->par = (mrp_xpar *) (blk + o_par);
    h->ref_start = S; h->ref_length = E - S; h->seg = -2; /* set when its level is staged */
    h->n_a = (int32_t) a->n; h->n_b = (int32_t) b->n;
    {   /* stRPHmm.profileSeqs: path A's reads, then path B's (hmm.c:559-566); the parents as the device will look them up */
        int32_t *rd = h->reads;
        mrp_xpar *pr = h->par;
        const r_hmm_vec *side[2] = {a, b};
        for (int q = 0; q < 2; q++)
            for (int64_t i = 0; i < side[q]->n; i++) {
                const rhmm *p = side[q]->a[i];
                memcpy(rd, p->reads, sizeof(int32_t) * (size_t) p->n_reads);
                rd += p->n_reads;
                pr->start = p->ref_start; pr->end = p->ref_start + p->ref_length; pr->n_cols = p->n_cols; pr->seg = p->seg; pr->col0 = p->col0;
                pr++;
            }
        h->n_reads = (int32_t) n_reads;
        h->first_read = n_reads > 0 ? h->reads[0] : -1;
    }
    /* mrp_side_bound for every depth, once per thread and stride */
    static __thread int32_t sb_stride = -1;
    static __thread int32_t sb[MRP_MAX_READ_PARTITIONING_DEPTH + 2];
    if (sb_stride != stride) { for (int d = 0; d <= MRP_MAX_READ_PARTITIONING_DEPTH + 1; d++) sb[d] = (int32_t) mrp_side_bound(d, stride); sb_stride = stride; }
    piter ia = {a, 0, 0, S, E}, ib = {b, 0, 0, S, E};
    rpiece pa, pb;
    if (!piter_next(&ia, &pa) || !piter_next(&ib, &pb)) { rhmm_destroy(h); return mrp_set_error(MRP_ERR_ARG, "cross product of an empty interval"); }
    int32_t pos = S, n = 0;
    int64_t D = 0;
    int rc = MRP_OK;
    for (;;) {
        const int32_t end = pa.end < pb.end ? pa.end : pb.end;
        const int32_t d1 = pa.depth, d2 = pb.depth, depth = d1 + d2;
        if (depth > MRP_MAX_READ_PARTITIONING_DEPTH) {
            rc = mrp_set_error(MRP_ERR_ARG, "cross product column depth %d exceeds %d", depth, MRP_MAX_READ_PARTITIONING_DEPTH);
            break;
        }
        if (n >= cap || end <= pos) { rc = mrp_set_error(MRP_ERR_ARG, "cross product: inconsistent tiling paths"); break; }
        h->starts[n] = pos;
        h->roff[n] = (int32_t) D;
        D += depth;
        const int64_t C = (int64_t) sb[d1] * sb[d2];
        h->bound_cells += C;
        if (C > h->bound_max_cells) h->bound_max_cells = (int32_t) C;
        h->depth_sites += (int64_t) depth * (end - pos);
        if (end < E) { /* merge column hmm.c:686-740: a piece that is cut leaves through an accept-mask connector (column.c:86-101) */
            const uint8_t oa = pa.end > end ? MRP_CONN_IDENT : pa.out, ob = pb.end > end ? MRP_CONN_IDENT : pb.out;
            const int64_t Ma = oa == MRP_CONN_ZERO ? 1 : sb[d1], Mb = ob == MRP_CONN_ZERO ? 1 : sb[d2];
            h->bound_merge += Ma * Mb;
            if (Ma * Mb > h->bound_max_merge) h->bound_max_merge = (int32_t) (Ma * Mb);
        }
        n++;
        pos = end;
        if (pos >= E) break;
        if (pa.end == end && !piter_next(&ia, &pa)) { rc = mrp_set_error(MRP_ERR_ARG, "cross product: tiling path A ends early"); break; }
        if (pb.end == end && !piter_next(&ib, &pb)) { rc = mrp_set_error(MRP_ERR_ARG, "cross product: tiling path B ends early"); break; }
    }
    if (rc == MRP_OK && D > 0x7FFFFFFFll) rc = mrp_set_error(MRP_ERR_UNSUPPORTED, "cross product with %lld column reads", (long long) D);
    if (rc != MRP_OK) { rhmm_destroy(h); return rc; }
    h->roff[n] = (int32_t) D;
    h->n_cols = n;
    if (h->bound_max_cells < 1) h->bound_max_cells = 1;
    if (h->bound_max_merge < 1) h->bound_max_merge = 1;
    *out = h;
    return MRP_OK;
}

/* A shadow as an ordinary flat hmm WITHOUT cells: column intervals, the reads of every column in bit order and where their
 * profile bytes start (profileSeq.c:41-47), and -- with_masks -- the masks of the merge columns (the reads of a column that
 * go on into the next one, in the bit positions of either column: what hmm.c:686-700 computes by merging the parents'
 * masks).  A column's reads are the hmm's reads that cover it, in the order of stRPHmm.profileSeqs: side A's reads precede
 * side B's in both (partitions.c:21-28, hmm.c:559-566), recursively. */
static mrp_hmm *r_expand(const world *w, const rhmm *x, int with_masks) {
    mrp_hmm *h = hmm_new();
    const int64_t K = x->n_cols, D = x->roff[K];
    h->ref_start = x->ref_start; h->ref_length = x->ref_length;
    VEC_RESERVE(h->reads, x->n_reads);
    memcpy(h->reads.a, x->reads, sizeof(int32_t) * (size_t) x->n_reads); h->reads.n = x->n_reads;
    VEC_RESERVE(h->col_start, K); VEC_RESERVE(h->col_len, K); VEC_RESERVE(h->col_depth, K);
    VEC_RESERVE(h->read_off, K + 1); VEC_RESERVE(h->cell_off, K + 1);
    VEC_RESERVE(h->col_reads, D + 1); VEC_RESERVE(h->read_byte_off, D + 1);
    const int32_t E = x->ref_start + x->ref_length;
    h->read_off.n = 0; h->cell_off.n = 0;
    for (int64_t k = 0; k < K; k++) {
        h->col_start.a[k] = x->starts[k];
        h->col_len.a[k] = (k + 1 < K ? x->starts[k + 1] : E) - x->starts[k];
        h->col_depth.a[k] = x->roff[k + 1] - x->roff[k];
        if (h->col_depth.a[k] > h->max_depth) h->max_depth = h->col_depth.a[k];
        h->read_off.a[k] = x->roff[k];
        h->cell_off.a[k] = 0;
    }
    h->read_off.a[K] = D; h->cell_off.a[K] = 0;
    h->col_start.n = h->col_len.n = h->col_depth.n = K;
    h->read_off.n = h->cell_off.n = K + 1;
    h->col_reads.n = h->read_byte_off.n = D;
    int32_t *fill = xmalloc(sizeof(int32_t) * (size_t) (K + 1));
    memcpy(fill, x->roff, sizeof(int32_t) * (size_t) (K + 1));
    int ok = 1;
    for (int64_t i = 0; i < x->n_reads && ok; i++) {
        const int32_t rd = x->reads[i];
        const mrp_read *r = &w->reads[rd];
        int64_t lo = 0, hi = K; /* the column that starts where the read does */
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (x->starts[mid] < r->ref_start) lo = mid + 1; else hi = mid; }
        if (lo >= K || x->starts[lo] != r->ref_start) { ok = 0; break; }
        for (int64_t k = lo; k < K && x->starts[k] < r->ref_start + r->length; k++) {
            if (fill[k] >= x->roff[k + 1]) { ok = 0; break; }
            h->col_reads.a[fill[k]] = rd;
            h->read_byte_off.a[fill[k]] = read_byte_offset(w, rd, x->starts[k]);
            fill[k]++;
        }
    }
    for (int64_t k = 0; k < K && ok; k++) if (fill[k] != x->roff[k + 1]) ok = 0;
    free(fill);
    if (!ok) { mrp_hmm_destroy(h); mrp_set_error(MRP_ERR_ARG, "shadow hmm: the column depths do not match the reads' intervals"); return NULL; }
    if (with_masks)
        for (int64_t k = 0; k + 1 < K; k++) {
            uint64_t mf = 0, mt = 0;
            const int32_t cut = x->starts[k + 1];
            const int32_t *ra = h->col_reads.a + h->read_off.a[k], *rb = h->col_reads.a + h->read_off.a[k + 1];
            for (int32_t i = 0; i < h->col_depth.a[k]; i++) if (w->reads[ra[i]].ref_start + w->reads[ra[i]].length > cut) mf |= (uint64_t) 1 << i;
            for (int32_t i = 0; i < h->col_depth.a[k + 1]; i++) if (w->reads[rb[i]].ref_start < cut) mt |= (uint64_t) 1 << i;
            VEC_PUSH(h->mask_from, mf); VEC_PUSH(h->mask_to, mt);
        }
    return h;
}

/* a cross product of a level: its shadow, its chunk, and -- filled by the thread that built it, while the shadow is in its cache --
 * what the engine is told about it (r_describe) */
typedef struct { rhmm *x; const world *w; mrp_xhmm d; } xbuild;
typedef VEC(xbuild) xbuild_vec;

/* mergeTwoTilingPaths coordination.c:263-339, structure only: the overlap components that need a cross
 * product are appended to xs (and, unpruned, to res); the others pass through */
static int64_t g_ns[6]; /* MRP_TIMING: components, tiling paths, cross shadows, destroy */
/* (only under MRP_TIMING: the thread CPU clock is a system call, and r_prepare_merge would make six of them per overlap
 * component -- 700 000 per call, 0.2 s of CPU) */
static int g_prepare_timing;
static double tcpu_ms(void) {
    if (!g_prepare_timing) return 0.0;
    struct timespec t; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t); return 1e3 * (double) t.tv_sec + 1e-6 * (double) t.tv_nsec;
}
#define T_ADD(slot, t0) do { if (g_prepare_timing) __atomic_fetch_add(&g_ns[slot], (int64_t) ((tcpu_ms() - (t0)) * 1e6), __ATOMIC_RELAXED); } while (0)
static int r_prepare_merge(const world *w, int32_t stride, r_hmm_vec *tp1, r_hmm_vec *tp2, r_hmm_vec *res, xbuild_vec *xs) {
    double tq = tcpu_ms();
    ar_on();
    r_comp_vec comps = r_overlapping_components(w, tp1, tp2);
    ar_off();
    T_ADD(0, tq);
    r_path_release(tp1); r_path_release(tp2);
    int rc = MRP_OK;
    for (int64_t i = 0; i < comps.n; i++) {
        r_component *comp = comps.a[i];
        if (rc == MRP_OK && comp->members.n == 1) { /* nothing overlaps it: passes through (coordination.c:317-322) */
            VEC_PUSH(*res, comp->members.a[0]);
        } else if (rc == MRP_OK && comp->members.n == 2) {
            /* two hmms that overlap -- the common case -- are two tiling paths of one hmm each, the smaller one by stRPHmm_cmpFn
             * first (getTilingPaths sorts, coordination.c:186-190): no lists to build */
            rhmm *m0 = comp->members.a[0], *m1 = comp->members.a[1];
            if (r_hmm_cmp(w, m0, m1) > 0) { rhmm *t_ = m0; m0 = m1; m1 = t_; }
            rhmm *ma[1] = {m0}, *mb[1] = {m1};
            const r_hmm_vec va = {ma, 1, 1}, vb = {mb, 1, 1};
            const int32_t S = m0->ref_start < m1->ref_start ? m0->ref_start : m1->ref_start;
            const int32_t Ea = m0->ref_start + m0->ref_length, Eb = m1->ref_start + m1->ref_length;
            xbuild xb; xb.x = NULL; xb.w = w;
            tq = tcpu_ms();
            rc = r_cross_build(w, &va, &vb, S, Ea > Eb ? Ea : Eb, stride, &xb.x);
            T_ADD(2, tq);
            /* the parents' shadows are no longer needed (their cells stay in the engine's segments; what the engine was told of
             * them it copied when their level was staged): back to THIS thread's front of the pool, warm for its next hmm */
            rhmm_destroy(m0); rhmm_destroy(m1);
            if (rc == MRP_OK) { VEC_PUSH(*xs, xb); VEC_PUSH(*res, xb.x); }
        } else if (rc == MRP_OK) {
            tq = tcpu_ms();
            ar_on();
            r_path_vec sub = r_tiling_paths_from(w, comp->members.a, comp->members.n);
            ar_off();
            T_ADD(1, tq);
            if (sub.n == 2) {
                r_hmm_vec *a = sub.a[0], *b = sub.a[1];
                int32_t S = a->a[0]->ref_start < b->a[0]->ref_start ? a->a[0]->ref_start : b->a[0]->ref_start;
                int32_t Ea = a->a[a->n - 1]->ref_start + a->a[a->n - 1]->ref_length;
                int32_t Eb = b->a[b->n - 1]->ref_start + b->a[b->n - 1]->ref_length;
                int32_t E = Ea > Eb ? Ea : Eb;
                xbuild xb; xb.x = NULL; xb.w = w;
                tq = tcpu_ms();
                rc = r_cross_build(w, a, b, S, E, stride, &xb.x);
                T_ADD(2, tq);
                for (int64_t t = 0; t < a->n; t++) rhmm_destroy(a->a[t]);
                for (int64_t t = 0; t < b->n; t++) rhmm_destroy(b->a[t]);
                if (rc == MRP_OK) { VEC_PUSH(*xs, xb); VEC_PUSH(*res, xb.x); }
            } else if (sub.n == 1 && sub.a[0]->n == 1) {
                VEC_PUSH(*res, sub.a[0]->a[0]);
            } else {
                rc = mrp_set_error(MRP_ERR_ARG, "overlap component with %lld tiling paths", (long long) sub.n);
            }
            for (int64_t t = 0; t < sub.n; t++) { free(sub.a[t]->a); free(sub.a[t]); }
            free(sub.a);
        }
        free(comp->members.a); free(comp);
    }
    free(comps.a);
    ar_rewind(); /* every temporary of this merge is gone */
    return rc;
}

/* the recursion tree of mergeTilingPaths (coordination.c:341-409) over all problems of a run */
typedef struct {
    int left, right;   /* children (node indices) or -1 */
    int height;        /* 0: a tiling path as it is; h > 0: merged at level h */
    const world *w;
    r_hmm_vec *path;   /* the node's tiling path once its level is done (owned) */
} rnode;
typedef VEC(rnode) rnode_vec;

static int r_leaf_node(rnode_vec *t, const world *w, r_hmm_vec *path) {
    rnode nd = {-1, -1, 0, w, path};
    VEC_PUSH(*t, nd);
    return (int) t->n - 1;
}
static int r_merge_node(rnode_vec *t, const world *w, int l, int r) {
    const int hl = t->a[l].height, hr = t->a[r].height;
    rnode nd = {l, r, 1 + (hl > hr ? hl : hr), w, NULL};
    VEC_PUSH(*t, nd);
    return (int) t->n - 1;
}
static int r_tree_of_paths(rnode_vec *t, const world *w, r_hmm_vec **paths, int64_t n) {
    if (n == 0) return r_leaf_node(t, w, xcalloc(1, sizeof(r_hmm_vec)));
    if (n == 1) return r_leaf_node(t, w, paths[0]);
    if (n == 2) return r_merge_node(t, w, r_leaf_node(t, w, paths[0]), r_leaf_node(t, w, paths[1]));
    const int l = r_tree_of_paths(t, w, paths, n / 2);
    const int r = r_tree_of_paths(t, w, paths + n / 2, n - n / 2);
    return r_merge_node(t, w, l, r);
}
/* getRPHmms coordination.c:490-516 as a subtree over leaves given in stRPHmm_cmpFn order (a chunk's leaves are sorted once:
 * the coverage filter and both strands walk the same order); returns the root node or -1 */
static int r_tree_of_sorted(rnode_vec *t, const world *w, rhmm *const *hmms, int64_t n, const mrp_params *params) {
    r_path_vec paths = r_tiling_paths_sorted(hmms, n);
    if (paths.n > MRP_MAX_READ_PARTITIONING_DEPTH || paths.n > params->max_coverage_depth) { /* :500-504 */
        const int64_t np = paths.n;
        for (int64_t i = 0; i < paths.n; i++) r_free_path(paths.a[i], 0);
        free(paths.a);
        mrp_set_error(MRP_ERR_ARG, "Coverage depth: read depth of %lld exceeds hard maximum of %d with configured maximum of %lld",
                      (long long) np, MRP_MAX_READ_PARTITIONING_DEPTH, (long long) params->max_coverage_depth);
        return -1;
    }
    /* the leaf paths as blocks of the shadow pool: the merge that consumes one (another thread, a level later) gives a block back
     * to its own front instead of freeing into this thread's malloc arena */
    for (int64_t i = 0; i < paths.n; i++) {
        r_hmm_vec *tp = paths.a[i];
        const size_t o_a = (sizeof(r_hmm_vec) + 15) & ~(size_t) 15;
        int cls = 0;
        char *blk = shadow_alloc(o_a + sizeof(rhmm *) * (size_t) (tp->n + 1), &cls);
        if (cls < 0) { free(blk); continue; } /* (beyond the pool's classes: stays as it is) */
        r_hmm_vec *v = (r_hmm_vec *) blk;
        v->a = (rhmm **) (blk + o_a); v->n = tp->n; v->cap = -(int64_t) cls - 2;
        memcpy(v->a, tp->a, sizeof(rhmm *) * (size_t) tp->n);
        free(tp->a); free(tp);
        paths.a[i] = v;
    }
    const int root = r_tree_of_paths(t, w, paths.a, paths.n);
    free(paths.a);
    return root;
}

static void r_free_tree(rnode_vec *t) {
    for (int64_t i = 0; i < t->n; i++) r_free_path(t->a[i].path, 1);
    free(t->a);
    t->a = NULL; t->n = t->cap = 0;
}

/* run every merge node, level by level */
static __thread double g_t_prepare, g_t_level; /* MRP_TIMING diagnostics of the calling thread */
typedef struct {
    rnode_vec *t;
    int64_t node;
    int32_t stride;
    uint32_t flags;    /* sweep_flags of the run: part of what the engine is told about every cross product */
    r_hmm_vec *res;
    xbuild_vec xs;
    int rc, res_class;
    int64_t x0;        /* where the item's cross products start in the level's list */
    void *big_block;
    char err[256];
} level_item;
static void r_describe(const world *w, const rhmm *x, uint32_t flags, mrp_xhmm *d);
static void level_prepare(int64_t i, void *arg) {
    level_item *it = &((level_item *) arg)[i];
    rnode *nd = &it->t->a[it->node];
    r_hmm_vec *l = it->t->a[nd->left].path, *r = it->t->a[nd->right].path;
    it->t->a[nd->left].path = NULL; it->t->a[nd->right].path = NULL;
    {   /* the merged path and the cross products to build: at most one entry per hmm of the two paths each.  ONE block of the
         * shadow pool, owned by the path (released when the next level consumes it; the list of cross products is read by the
         * thread that drives the batch before that) -- no malloc / free across threads */
        const int64_t cap = l->n + r->n + 1;
        const size_t o_res = (sizeof(r_hmm_vec) + 15) & ~(size_t) 15, o_xs = o_res + sizeof(rhmm *) * (size_t) cap,
                     bytes = o_xs + sizeof(xbuild) * (size_t) cap;
        int cls = 0;
        char *blk = shadow_alloc(bytes, &cls);
        it->res = (r_hmm_vec *) blk;
        it->res->a = (rhmm **) (blk + o_res); it->res->n = 0; it->res->cap = cap;
        it->xs.a = (xbuild *) (blk + o_xs); it->xs.n = 0; it->xs.cap = cap;
        it->res_class = cls;
    }
    it->rc = r_prepare_merge(nd->w, it->stride, l, r, it->res, &it->xs);
    for (int64_t j = 0; j < it->xs.n && it->rc == MRP_OK; j++) r_describe(it->xs.a[j].w, it->xs.a[j].x, it->flags, &it->xs.a[j].d);
    if (it->res_class >= 0) it->res->cap = -(int64_t) it->res_class - 2;
    else { /* (a path beyond the pool's largest class: an ordinary vector again) */
        r_hmm_vec *v = xcalloc(1, sizeof(*v));
        v->a = xmalloc(sizeof(rhmm *) * (size_t) (it->res->n + 1)); memcpy(v->a, it->res->a, sizeof(rhmm *) * (size_t) it->res->n);
        v->n = it->res->n; v->cap = it->res->n + 1;
        it->big_block = it->res; it->res = v;
    }
    if (it->rc != MRP_OK) snprintf(it->err, sizeof(it->err), "%s", mrp_last_error());
}
static void level_drop_garbage(int64_t i, void *arg) { /* (the parents' shadows went back to the pool in r_prepare_merge) */
    level_item *it = &((level_item *) arg)[i];
    if (it->big_block) { free(it->big_block); it->big_block = NULL; }
}
static void level_finish(int64_t i, void *arg) {
    level_item *it = &((level_item *) arg)[i];
    rnode *nd = &it->t->a[it->node];
    r_sort_hmms(nd->w, it->res->a, it->res->n); /* coordination.c:336 */
}
/* a shadow as the engine is told about it */
static void r_describe(const world *w, const rhmm *x, uint32_t flags, mrp_xhmm *d) {
    memset(d, 0, sizeof(*d));
    d->chunk = w->chunk; d->flags = flags; d->discarded = &w->failed;
    d->ref_start = x->ref_start; d->ref_end = x->ref_start + x->ref_length;
    d->n_cols = x->n_cols; d->n_a = x->n_a; d->n_b = x->n_b; d->par = x->par;
    d->col_start = x->starts; d->col_read_off = x->roff;
    d->bound_cells = x->bound_cells; d->bound_merge = x->bound_merge; d->depth_sites = x->depth_sites;
    d->bound_max_cells = x->bound_max_cells; d->bound_max_merge = x->bound_max_merge;
    d->n_col_reads = x->roff[x->n_cols];
    d->n_slots = (int64_t) w->ch.allele_offset[x->ref_start + x->ref_length] - (int64_t) w->ch.allele_offset[x->ref_start];
}
typedef struct { rhmm *x; const world *w; } xowner; /* what the host keeps of a cross product while its level is on the device */
typedef struct { level_item *items; mrp_xhmm *xh; xowner *xb; } level_gather_ctl;
static void level_gather(int64_t i, void *arg) { /* (the descriptions were written by level_prepare, one item's side by side: a copy) */
    const level_gather_ctl *g = arg;
    const level_item *it = &g->items[i];
    for (int64_t j = 0; j < it->xs.n; j++) {
        g->xb[it->x0 + j].x = it->xs.a[j].x; g->xb[it->x0 + j].w = it->xs.a[j].w;
        g->xh[it->x0 + j] = it->xs.a[j].d;
    }
}
/* what the host keeps of a level while it is on the device */
typedef struct {
    level_item *items; int64_t n_items;
    mrp_xhmm *xh; xowner *xb; int64_t n_x;
    int cls_items, cls_xh, cls_xb; /* the three lists are blocks of the shadow pool (megabytes per level: warm instead of mapped afresh) */
    int64_t seq;                   /* the level is the seq-th the engine launched: over once mrp_engine_levels_ended() reaches seq */
} level_run;
/* the levels launched and not yet settled, oldest first (the engine ends small levels late: several are in flight at a time) */
#define LEVEL_RUNS_MAX 72
typedef struct { level_run r[LEVEL_RUNS_MAX]; int head, tail; int64_t launched; } level_runs;
static void level_run_settle(level_run *r, int rc_ok) { /* the level has ended: its error flags are in */
    for (int64_t i = 0; i < r->n_x; i++)
        /* outside what the kernels handle (a parent not in complement-pair order, ...): the later levels leave this
         * chunk out, its caller redoes it on the hashing path */
        if (rc_ok && r->xh[i].err != 0) ((world *) r->xb[i].w)->failed = 1;
    if (r->xh) shadow_release(r->xh, r->cls_xh);
    if (r->xb) shadow_release(r->xb, r->cls_xb);
    if (r->items) shadow_release(r->items, r->cls_items);
    memset(r, 0, sizeof(*r));
}
/* the levels the engine has ended (all of them: `all`, after mrp_engine_level_end or a launch that waited) */
static void level_runs_settle(level_runs *q, mrp_engine *e, int rc_ok, int all) {
    const int64_t ended = mrp_engine_levels_ended(e);
    while (q->head < q->tail && (all || q->r[q->head].seq <= ended)) level_run_settle(&q->r[q->head++], rc_ok);
    if (q->head == q->tail) q->head = q->tail = 0;
}
/* pending: if not NULL the levels still in flight are left there (the caller stages what comes next beside them, then launches / ends
 * and settles them); otherwise they are ended here */
static int r_run_tree(mrp_engine *e, rnode_vec *t, const mrp_params *params, level_runs *pending) {
    int max_h = 0;
    for (int64_t i = 0; i < t->n; i++) if (t->a[i].height > max_h) max_h = t->a[i].height;
    /* Level of a merge node = as late as its parent allows (every root at the last level), not its height: the merges
     * near the roots are the expensive ones (a workgroup walks ~10^3 columns of ~10^4 cells one after the other) and a
     * level costs what its longest hmm costs, so they should share their levels -- a problem one merge deeper than the
     * others then adds a level of small leaf merges instead of a level with a single large hmm on an idle device. */
    int *lvl = xmalloc(sizeof(int) * (size_t) (t->n + 1));
    for (int64_t i = 0; i < t->n; i++) lvl[i] = -1;
    for (int64_t i = t->n - 1; i >= 0; i--) { /* children precede their parent */
        if (lvl[i] < 0) lvl[i] = max_h;
        if (t->a[i].left >= 0) { lvl[t->a[i].left] = lvl[i] - 1; lvl[t->a[i].right] = lvl[i] - 1; }
    }
    const uint32_t flags = sweep_flags(params);
    const int32_t stride = mrp_engine_stride(e);
    int rc = MRP_OK;
    /* The host's part of level h -- which hmms are merged, their column boundaries -- needs nothing the device computes
     * (only WHERE level h - 1's results are, fixed when that level was staged): it is done while level h - 1 runs.  The one
     * wait per level is inside mrp_engine_level_launch. */
    level_runs own = {0}, *runs = pending ? pending : &own;
    for (int h = 1; h <= max_h && rc == MRP_OK; h++) {
        const double t0 = now_ms();
        int64_t n_items = 0;
        for (int64_t i = 0; i < t->n; i++) if (t->a[i].height > 0 && lvl[i] == h && !t->a[i].w->failed) n_items++;
        if (n_items == 0) continue;
        int cls_items = -1, cls_xh = -1, cls_xb = -1;
        level_item *items = shadow_alloc(sizeof(*items) * ((size_t) n_items + 1), &cls_items);
        memset(items, 0, sizeof(*items) * ((size_t) n_items + 1));
        n_items = 0;
        for (int64_t i = 0; i < t->n; i++) if (t->a[i].height > 0 && lvl[i] == h && !t->a[i].w->failed) { items[n_items].t = t; items[n_items].node = i; items[n_items].stride = stride; items[n_items].flags = flags; n_items++; }
        /* the merges of a level touch disjoint nodes: structure in parallel, device work as one batch */
        mrp_pool_set_tag(1); mrp_pool_set_weight(8000); mrp_pool_run(n_items, n_items > 512 ? (n_items > 16384 ? 64 : n_items / 256) : 1, level_prepare, items); mrp_pool_set_weight(0); mrp_pool_set_tag(0);
        const double ta = now_ms();
        int64_t n_x = 0;
        for (int64_t i = 0; i < n_items; i++) {
            n_x += items[i].xs.n;
            if (items[i].rc != MRP_OK && rc == MRP_OK) rc = mrp_set_error(items[i].rc, "%s", items[i].err);
        }
        mrp_xhmm *xh = shadow_alloc(sizeof(*xh) * ((size_t) n_x + 1), &cls_xh);
        xowner *xb = shadow_alloc(sizeof(*xb) * ((size_t) n_x + 1), &cls_xb);
        n_x = 0;
        for (int64_t i = 0; i < n_items; i++) { items[i].x0 = n_x; n_x += items[i].xs.n; }
        {   /* the level as the engine is told about it, in node order (worker threads: the shadows are in their caches) */
            level_gather_ctl gc = {items, xh, xb};
            mrp_pool_set_tag(3); mrp_pool_set_weight(400); mrp_pool_run(n_items, n_items > 512 ? n_items / 256 : 1, level_gather, &gc); mrp_pool_set_weight(0); mrp_pool_set_tag(0);
        }
        for (int64_t i = 0; i < n_items; i++) /* coordination.c:312: one forward/backward per overlap component */
            ((world *) t->a[items[i].node].w)->n_sweeps += (int) items[i].xs.n, items[i].xs.a = NULL; /* (the list is part of the path's block) */
        const double tb = now_ms();
        if (rc == MRP_OK) rc = mrp_engine_level_stage(e, n_x, xh);
        const double tc = now_ms();
        /* where the level's results will be is known from here on: the next level can be described against them */
        for (int64_t i = 0; i < n_x && rc == MRP_OK; i++) { xb[i].x->seg = xh[i].seg; xb[i].x->col0 = xh[i].col0; }
        mrp_pool_set_tag(2); mrp_pool_set_weight(300); if (rc == MRP_OK) mrp_pool_run(n_items, n_items > 512 ? (n_items > 16384 ? 64 : n_items / 256) : 1, level_finish, items); mrp_pool_set_weight(0); mrp_pool_set_tag(0);
        for (int64_t i = 0; i < n_items; i++) t->a[items[i].node].path = items[i].res;
        const double t1 = now_ms();
        /* level h goes to the device (a large level first waits for the levels before it, a small one does not) */
        int launched = 0;
        if (rc == MRP_OK) { rc = mrp_engine_level_launch(e); launched = rc == MRP_OK && n_x > 0; /* (an empty level is not staged: the launch only ends what runs) */ }
        else (void) mrp_engine_level_end(e);
        level_runs_settle(runs, e, rc == MRP_OK, rc != MRP_OK);
        const double t2 = now_ms();
        for (int64_t i = 0; i < n_items; i++) level_drop_garbage(i, items);
        (void) t2;
        g_t_prepare += t1 - t0; g_t_level += now_ms() - t1;
        if (getenv("MRP_TIMING"))
            fprintf(stderr, "    host level %d: prepare %.2f ms, gather %.2f, stage %.2f, sort %.2f | launch (waits for the level before) %.2f | settle+garbage %.2f\n",
                    h, ta - t0, tb - ta, tc - tb, t1 - tc, t2 - t1, now_ms() - t2);
        {
            level_run cur = {items, n_items, xh, xb, n_x, cls_items, cls_xh, cls_xb, 0};
            if (launched && runs->tail < LEVEL_RUNS_MAX) { cur.seq = ++runs->launched; runs->r[runs->tail++] = cur; }
            else { /* (not launched: the engine holds nothing of it; a queue that is full cannot happen -- the engine has 64 segments) */
                if (launched) { (void) mrp_engine_level_end(e); level_runs_settle(runs, e, rc == MRP_OK, 1); }
                level_run_settle(&cur, launched && rc == MRP_OK);
            }
        }
    }
    if (!(pending && rc == MRP_OK)) {
        const double t1 = now_ms();
        const int rc2 = mrp_engine_level_end(e);
        if (rc == MRP_OK) rc = rc2;
        level_runs_settle(runs, e, rc == MRP_OK, 1);
        g_t_level += now_ms() - t1;
    }
    free(lvl);
    return rc;
}

/* resident shadow -> ordinary flat hmm on the host; phase 0 queues the copies, phase 1 (after
 * mrp_engine_sync) unpacks them */
typedef struct { mrp_hmm *h; uint64_t *part; uint32_t *np; int32_t *n_cells, *n_merge; int32_t stride; } r_staging;
static int r_download_begin(mrp_engine *e, const world *w, const rhmm *x, r_staging *st) {
    memset(st, 0, sizeof(*st));
    st->h = r_expand(w, x, 1);
    if (!st->h) return MRP_ERR_ARG;
    if (x->seg < 0) return MRP_OK; /* a stRPHmm_construct hmm: nothing to fetch */
    const int64_t K = x->n_cols;
    st->stride = mrp_engine_stride(e);
    const int64_t n = K * st->stride;
    const uint64_t *d_part; const uint32_t *d_np; const int32_t *d_nc, *d_nm;
    int rc = mrp_engine_locate(e, x->seg, x->col0, &d_part, &d_np, &d_nc, &d_nm);
    if (rc != MRP_OK) return rc;
    st->part = xmalloc(sizeof(uint64_t) * (size_t) n);
    st->np = xmalloc(sizeof(uint32_t) * (size_t) n);
    st->n_cells = xmalloc(sizeof(int32_t) * (size_t) K);
    st->n_merge = xmalloc(sizeof(int32_t) * (size_t) K);
    rc = mrp_engine_fetch(e, st->part, d_part, (int64_t) sizeof(uint64_t) * n);
    if (rc == MRP_OK) rc = mrp_engine_fetch(e, st->np, d_np, (int64_t) sizeof(uint32_t) * n);
    if (rc == MRP_OK) rc = mrp_engine_fetch(e, st->n_cells, d_nc, (int64_t) sizeof(int32_t) * K);
    if (rc == MRP_OK) rc = mrp_engine_fetch(e, st->n_merge, d_nm, (int64_t) sizeof(int32_t) * K);
    return rc;
}
static void r_staging_free(r_staging *st) { free(st->part); free(st->np); free(st->n_cells); free(st->n_merge); memset(st, 0, sizeof(*st)); }
static mrp_hmm *r_download_end(r_staging *st) {
    mrp_hmm *h = st->h;
    const int64_t K = hmm_K(h);
    h->cell_off.n = 0; h->mcell_off.n = 0;
    VEC_PUSH(h->cell_off, 0);
    VEC_PUSH(h->mcell_off, 0);
    if (!st->part) { /* hmm.c:97-133 */
        hmm_add_cell(h, 1, 0);
        hmm_add_cell(h, 0, 0);
        VEC_PUSH(h->cell_off, h->part.n);
    } else {
        for (int64_t k = 0; k < K; k++) {
            const int64_t o = k * st->stride;
            for (int32_t i = 0; i < st->n_cells[k]; i++) {
                hmm_add_cell(h, st->part[o + i], st->np[o + i] >> 16);
                h->next.a[h->next.n - 1] = st->np[o + i] & 0xFFFFu;
            }
            VEC_PUSH(h->cell_off, h->part.n);
            if (k + 1 < K) {
                for (int32_t m = 0; m < st->n_merge[k]; m++) { VEC_PUSH(h->mfrom, 0); VEC_PUSH(h->mto, 0); }
                VEC_PUSH(h->mcell_off, h->mfrom.n);
            }
        }
        /* a merge cell's keys are the masked partition of any cell that feeds it / is fed by it
         * (mergeColumn.c:63-79); every merge cell the prune keeps has both */
        for (int64_t k = 0; k < K; k++)
            for (int64_t c = h->cell_off.a[k]; c < h->cell_off.a[k + 1]; c++) {
                if (k + 1 < K) h->mfrom.a[h->mcell_off.a[k] + h->next.a[c]] = h->part.a[c] & h->mask_from.a[k];
                if (k > 0) h->mto.a[h->mcell_off.a[k - 1] + h->prev.a[c]] = h->part.a[c] & h->mask_to.a[k - 1];
            }
    }
    st->h = NULL;
    r_staging_free(st);
    return h;
}
static int r_download_path(mrp_engine *e, const world *w, const r_hmm_vec *tp, mrp_hmm ***out) {
    r_staging *st = xcalloc((size_t) tp->n + 1, sizeof(*st));
    mrp_hmm **res = xcalloc((size_t) tp->n + 1, sizeof(*res));
    int rc = MRP_OK;
    for (int64_t i = 0; i < tp->n && rc == MRP_OK; i++) rc = r_download_begin(e, w, tp->a[i], &st[i]);
    if (rc == MRP_OK) rc = mrp_engine_sync(e);
    for (int64_t i = 0; i < tp->n; i++) {
        if (rc == MRP_OK) res[i] = r_download_end(&st[i]);
        else { mrp_hmm_destroy(st[i].h); st[i].h = NULL; r_staging_free(&st[i]); }
    }
    free(st);
    if (rc != MRP_OK) { free(res); res = NULL; }
    *out = res;
    return rc;
}

int mrp_get_rp_hmms_resident(mrp_context *ctx, const mrp_chunk *chunk, const mrp_read *reads, const int32_t *read_index,
                             int64_t n, const mrp_params *params, mrp_hmm ***hmms_out, int64_t *n_out) {
    if (!params || !hmms_out || !n_out || n < 0 || (n > 0 && !read_index)) return mrp_set_error(MRP_ERR_ARG, "mrp_get_rp_hmms_resident: bad arguments");
    int64_t max_idx = -1;
    for (int64_t i = 0; i < n; i++) { if (read_index[i] < 0) return mrp_set_error(MRP_ERR_ARG, "negative read index"); if (read_index[i] > max_idx) max_idx = read_index[i]; }
    world w;
    int rc = world_init(&w, ctx, chunk, reads, max_idx + 1, NULL);
    if (rc != MRP_OK) return rc;
    mrp_engine *e = NULL;
    rc = mrp_engine_create(ctx, params, &e);
    if (rc != MRP_OK) return rc;
    int leaves_class = -1;
    rleaf *leaves = r_leaves_of_chunk(&w, &leaves_class);
    rnode_vec tree = {0};
    rhmm **picked = xmalloc(sizeof(*picked) * (size_t) (n + 1));
    for (int64_t i = 0; i < n; i++) picked[i] = &leaves[read_index[i]].h;
    r_sort_hmms(&w, picked, n);
    const int root = r_tree_of_sorted(&tree, &w, picked, n, params);
    free(picked);
    rc = root < 0 ? MRP_ERR_ARG : r_run_tree(e, &tree, params, NULL);
    if (rc == MRP_OK && w.failed) rc = mrp_set_error(MRP_ERR_UNSUPPORTED, "device-resident merge: an hmm outside what the kernels handle (pair order / kept merge cells)");
    if (rc == MRP_OK) {
        mrp_hmm **res = NULL;
        rc = r_download_path(e, &w, tree.a[root].path, &res);
        if (rc == MRP_OK) { *n_out = tree.a[root].path->n; *hmms_out = res; }
    }
    r_free_tree(&tree);
    shadow_release(leaves, leaves_class);
    mrp_engine_destroy(e);
    return rc;
}

/* bubbleGraph_phaseBubbleGraph (bubbleGraph.c:2673-2801) for a set of chunks at once: the loop body of
 * phase.c:276-473 that phases one chunk, with the merge levels of all chunks run together */
typedef struct {
    world w;
    int32_t *discarded; int64_t nd;
    int root;            /* node of the joined tiling path (in the chunk's own tree, then in the run's tree) */
    rnode_vec tree;      /* the chunk's subtree while it is being set up */
    rhmm *hmm;           /* shadow of the fused final hmm */
    int32_t *path;       /* traced-back cell per column */
    uint64_t *chosen;    /* and its partition */
    double fwd, bwd;
    int64_t final_index;
    rleaf *leaves;       /* leaf shadows of all reads of the chunk */
    int leaves_class;
    int rc;
    char err[256];
    /* the genome fragment as the device leaves it (mrp_xhmm.frag_*): 20 bytes per site, the two read lists */
    int32_t *by_pool, *frag_reads1, *frag_reads2;
    void *frag_sites;
    int32_t frag_n1, frag_n2, frag_done;
} many_state;

typedef struct {
    many_state *st;
    mrp_context *ctx;
    const mrp_chunk *const *chunks;
    const mrp_read *const *reads;
    const int64_t *n_reads;
    const mrp_params *params, *pc;
    mrp_engine *e;
    rnode_vec *tree;
    mrp_phase_result **out;
    mrp_xhmm *xfinal;
    uint32_t final_flags;
} many_ctl;

/* bubbleGraph.c:2699-2745 for one chunk: coverage filter, strand split, the two getRPHmms subtrees and their join */
static void many_setup(int64_t c, void *arg) {
    many_ctl *ctl = arg;
    many_state *m = &ctl->st[c];
    m->root = -1;
    m->rc = world_init(&m->w, ctl->ctx, ctl->chunks[c], ctl->reads[c], ctl->n_reads[c], NULL);
    if (m->rc == MRP_OK && ctl->n_reads[c] > 0) {
        const int64_t nr = ctl->n_reads[c];
        m->leaves = r_leaves_of_chunk(&m->w, &m->leaves_class);
        /* every getTilingPaths of the chunk (coverage filter, either strand) starts by sorting its hmms with stRPHmm_cmpFn
         * (coordination.c:186-190): the leaves are sorted once, subsets keep the order */
        rhmm **sorted = xmalloc(sizeof(*sorted) * (size_t) (nr + 1));
        for (int64_t i = 0; i < nr; i++) sorted[i] = &m->leaves[i].h;
        r_sort_hmms(&m->w, sorted, nr);
        int32_t *filtered = xmalloc(sizeof(int32_t) * (size_t) nr);
        m->discarded = xmalloc(sizeof(int32_t) * (size_t) nr);
        int64_t nf;
        r_filter_reads_by_coverage_depth(&m->w, sorted, ctl->params, filtered, &nf, m->discarded, &m->nd); /* :2699 */
        uint8_t *is_disc = xcalloc((size_t) nr, 1);
        for (int64_t i = 0; i < m->nd; i++) is_disc[m->discarded[i]] = 1;
        rhmm **fwd = xmalloc(sizeof(*fwd) * (size_t) (nr + 1)), **rev = xmalloc(sizeof(*rev) * (size_t) (nr + 1));
        int64_t nfwd = 0, nrev = 0;
        for (int64_t i = 0; i < nr; i++) { /* :2705-2716 */
            const int32_t rd = sorted[i]->first_read;
            if (is_disc[rd]) continue;
            if (ctl->reads[c][rd].forward_strand) fwd[nfwd++] = sorted[i]; else rev[nrev++] = sorted[i];
        }
        const int rf = r_tree_of_sorted(&m->tree, &m->w, fwd, nfwd, ctl->pc);   /* :2736 */
        const int rr = rf < 0 ? -1 : r_tree_of_sorted(&m->tree, &m->w, rev, nrev, ctl->pc); /* :2740 */
        if (rf < 0 || rr < 0) m->rc = MRP_ERR_ARG;
        else m->root = r_merge_node(&m->tree, &m->w, rf, rr);                         /* :2745 */
        free(filtered); free(is_disc); free(fwd); free(rev); free(sorted);
    }
    if (m->rc != MRP_OK) snprintf(m->err, sizeof(m->err), "%s", mrp_last_error());
}
/* stRPHmm_fuse of the joined tiling path (hmm.c:283-372, gap columns :335-359) = its cross product with nothing:
 * the shadow of the final hmm as the device builds it */
static void many_final_shadow(int64_t c, void *arg) {
    many_ctl *ctl = arg;
    many_state *m = &ctl->st[c];
    if (m->root < 0 || m->rc != MRP_OK || m->w.failed) return;
    r_hmm_vec *joined = ctl->tree->a[m->root].path;
    if (joined->n == 0) return;
    const int32_t S = joined->a[0]->ref_start, E = joined->a[joined->n - 1]->ref_start + joined->a[joined->n - 1]->ref_length;
    r_hmm_vec nothing = {0};
    m->rc = r_cross_build(&m->w, joined, &nothing, S, E, mrp_engine_stride(ctl->e), &m->hmm);
    if (m->rc != MRP_OK) { m->hmm = NULL; snprintf(m->err, sizeof(m->err), "%s", mrp_last_error()); return; }
    const int64_t K = m->hmm->n_cols;
    m->path = xmalloc(sizeof(int32_t) * (size_t) K);
    m->chosen = xmalloc(sizeof(uint64_t) * (size_t) K);
    mrp_xhmm *x = &ctl->xfinal[c];
    r_describe(&m->w, m->hmm, ctl->final_flags, x);
    x->n_cells = m->path;
    x->path_part = m->chosen;
    m->w.n_sweeps += 1; /* bubbleGraph.c:2749 */
    {   /* the genome fragment on the device, behind the trace back: the chunk's reads, their order by pool offset (a column names a
         * read by where its profile bytes are), the reads the coverage filter took out */
        const int64_t nr = m->w.n_reads;
        m->by_pool = xmalloc(sizeof(int32_t) * (size_t) (nr + 1));
        int sorted = 1;
        for (int64_t i = 0; i < nr; i++) { m->by_pool[i] = (int32_t) i; if (i > 0 && m->w.reads[i].pool_offset < m->w.reads[i - 1].pool_offset) sorted = 0; }
        if (!sorted) { /* (rare: callers lay the profiles out in read order) insertion into a keyed array, then a plain sort */
            keyed *a = xmalloc(sizeof(keyed) * (size_t) (nr + 1)), *t = xmalloc(sizeof(keyed) * (size_t) (nr + 1));
            for (int64_t i = 0; i < nr; i++) { a[i].idx = i; a[i].key = -(double) m->w.reads[i].pool_offset; } /* keyed_sort_desc: descending key */
            keyed_sort_desc(a, nr, t);
            for (int64_t i = 0; i < nr; i++) m->by_pool[i] = (int32_t) a[i].idx;
            free(a); free(t);
        }
        const int64_t len = (int64_t) x->ref_end - x->ref_start;
        m->frag_sites = xmalloc(20 * (size_t) (len + 1));
        m->frag_reads1 = xmalloc(sizeof(int32_t) * (size_t) (2 * nr + 2));
        m->frag_reads2 = xmalloc(sizeof(int32_t) * (size_t) (2 * nr + 2));
        x->frag_reads = m->w.reads; x->frag_n_reads = (int32_t) nr; x->frag_by_pool = m->by_pool;
        x->frag_discarded = m->discarded; x->frag_n_discarded = (int32_t) m->nd;
        x->frag_iterations = (int32_t) ctl->params->rounds_of_iterative_refinement;
        x->frag_sites = m->frag_sites; x->frag_reads1 = m->frag_reads1; x->frag_reads2 = m->frag_reads2;
        x->frag_n1 = x->frag_n2 = 0; x->frag_done = 0;
    }
}
static void many_finish(int64_t c, void *arg) {
    many_ctl *ctl = arg;
    many_state *m = &ctl->st[c];
    if (m->w.failed) ctl->out[c] = NULL; /* redone by the caller on the hashing path */
    else if (m->hmm && m->frag_done) { /* the fragment came from the device: widen it into the result's arrays */
        const int32_t start = m->hmm->ref_start, len = m->hmm->ref_length;
        mrp_phase_result *g = result_new(start, len, m->w.n_reads);
        const struct { uint8_t anc, h1, h2, s1, s2, pad[3]; float gp, p1, p2; } *fs = m->frag_sites;
        for (int32_t q = 0; q < len; q++) {
            const uint64_t A = m->w.ch.allele_number[start + q], h1 = fs[q].h1, h2 = fs[q].h2;
            g->ancestor_string[q] = fs[q].anc; g->haplotype_string1[q] = h1; g->haplotype_string2[q] = h2;
            g->genotype_string[q] = h1 < h2 ? h1 * A + h2 : h2 * A + h1;
            g->genotype_probs[q] = fs[q].gp; g->haplotype_probs1[q] = fs[q].p1; g->haplotype_probs2[q] = fs[q].p2;
            g->reads_supporting_haplotype1[q] = fs[q].s1; g->reads_supporting_haplotype2[q] = fs[q].s2;
        }
        memcpy(g->reads1, m->frag_reads1, sizeof(int32_t) * (size_t) m->frag_n1); g->n_reads1 = m->frag_n1;
        memcpy(g->reads2, m->frag_reads2, sizeof(int32_t) * (size_t) m->frag_n2); g->n_reads2 = m->frag_n2;
        g->hmm_forward = m->fwd; g->hmm_backward = m->bwd; g->n_sweeps = m->w.n_sweeps;
        ctl->out[c] = g;
    }
    else if (m->hmm) {
        mrp_hmm *flat = r_expand(&m->w, m->hmm, 0);
        if (!flat) { m->rc = MRP_ERR_ARG; snprintf(m->err, sizeof(m->err), "%s", mrp_last_error()); return; }
        finish_phase_parts(&m->w, flat, m->chosen, m->fwd, m->bwd, ctl->params, m->discarded, m->nd, &ctl->out[c]);
        mrp_hmm_destroy(flat);
    }
    else ctl->out[c] = result_new(0, 0, ctl->n_reads[c]);
}

static int phase_many_resident(mrp_context *ctx, int64_t n_chunks, const mrp_chunk *const *chunks, const mrp_read *const *reads,
                               const int64_t *n_reads, const mrp_params *params, mrp_phase_result **out,
                               mrp_phase_many_stats *stats) {
    mrp_params pc = *params;
    pc.include_ancestor_sub_prob = 0; /* bubbleGraph.c:2733 */
    mrp_engine *e = NULL;
    const double t_enter = now_ms();
    if (getenv("MRP_TIMING")) fprintf(stderr, "  batch of %lld chunks enters at %.1f ms on the process clock\n", (long long) n_chunks, fmod(t_enter, 1e5));
    int rc = mrp_engine_create(ctx, &pc, &e);
    if (rc != MRP_OK) return rc;
    many_state *st = xcalloc((size_t) n_chunks + 1, sizeof(*st));
    rnode_vec tree = {0};
    const int timing = getenv("MRP_TIMING") != NULL;
    g_prepare_timing = getenv("MRP_TIMING_PREPARE") != NULL;
    double tt[6];
    tt[0] = now_ms(); g_t_prepare = g_t_level = 0;
    for (int q = 0; q < 6; q++) __atomic_store_n(&g_ns[q], 0, __ATOMIC_RELAXED); /* (diagnostics shared by the concurrent halves) */
    many_ctl ctl = {st, ctx, chunks, reads, n_reads, params, &pc, e, &tree, out, NULL, 0};
    tt[3] = tt[2] = 0;
    mrp_pool_set_tag(4); parallel_for(n_chunks, many_setup, &ctl); mrp_pool_set_tag(0);
    for (int64_t c = 0; c < n_chunks; c++) { /* splice the chunks' subtrees into one tree */
        many_state *m = &st[c];
        if (m->rc != MRP_OK && rc == MRP_OK) rc = mrp_set_error(m->rc, "%s", m->err);
        const int base = (int) tree.n;
        for (int64_t i = 0; i < m->tree.n; i++) {
            rnode nd = m->tree.a[i];
            if (nd.left >= 0) nd.left += base;
            if (nd.right >= 0) nd.right += base;
            VEC_PUSH(tree, nd);
        }
        if (m->root >= 0) m->root += base;
        free(m->tree.a);
        m->tree.a = NULL; m->tree.n = m->tree.cap = 0;
    }
    tt[1] = now_ms();
    level_runs pending = {0}; /* the last merge levels: still on the device while the final stage is described */
    if (rc == MRP_OK) rc = r_run_tree(e, &tree, &pc, &pending);
    tt[2] = now_ms();
    tt[3] = now_ms();
    /* fuse the joined path (:2745-2747), final sweep with the ancestor model (:2748-2749) and trace back (:2755) on
     * the device, all chunks in one batch: only the traced-back partition of every column comes back */
    pc.include_ancestor_sub_prob = 1;
    if (rc == MRP_OK) {
        ctl.final_flags = sweep_flags(&pc);
        ctl.xfinal = xcalloc((size_t) n_chunks + 1, sizeof(*ctl.xfinal));
        mrp_pool_set_tag(5); parallel_for(n_chunks, many_final_shadow, &ctl); mrp_pool_set_tag(0);
        mrp_xhmm *xh = xcalloc((size_t) n_chunks + 1, sizeof(*xh));
        int64_t nj = 0;
        for (int64_t c = 0; c < n_chunks; c++) {
            if (st[c].rc != MRP_OK && rc == MRP_OK) rc = mrp_set_error(st[c].rc, "%s", st[c].err);
            if (st[c].hmm) { st[c].final_index = nj; xh[nj++] = ctl.xfinal[c]; }
        }
        if (rc == MRP_OK) rc = mrp_engine_final_stage(e, nj, xh);
        /* the wait for the last merge level (its error flags mark the chunks to redo), then the final stage goes */
        if (rc == MRP_OK) rc = mrp_engine_level_launch(e);
        else (void) mrp_engine_level_end(e);
        level_runs_settle(&pending, e, rc == MRP_OK, 1); /* (the final level's launch waits for everything before it) */
        if (rc == MRP_OK) rc = mrp_engine_level_end(e);
        for (int64_t c = 0; c < n_chunks; c++)
            if (st[c].hmm) {
                st[c].fwd = xh[st[c].final_index].hmm_forward; st[c].bwd = xh[st[c].final_index].hmm_backward;
                st[c].frag_done = rc == MRP_OK ? xh[st[c].final_index].frag_done : 0;
                st[c].frag_n1 = xh[st[c].final_index].frag_n1; st[c].frag_n2 = xh[st[c].final_index].frag_n2;
                if (rc == MRP_OK && xh[st[c].final_index].err != 0) st[c].w.failed = 1;
            }
        free(xh);
        free(ctl.xfinal);
    } else {
        (void) mrp_engine_level_end(e);
        level_runs_settle(&pending, e, 0, 1);
    }
    tt[4] = now_ms();
    if (rc == MRP_OK) {
        mrp_pool_set_tag(6); parallel_for(n_chunks, many_finish, &ctl); mrp_pool_set_tag(0);
        for (int64_t c = 0; c < n_chunks; c++)
            if (st[c].rc != MRP_OK && rc == MRP_OK) rc = mrp_set_error(st[c].rc, "%s", st[c].err);
    }
    int64_t n_failed = 0;
    for (int64_t c = 0; c < n_chunks && rc == MRP_OK; c++) {
        if (!st[c].w.failed) continue;
        /* bubbleGraph_phaseBubbleGraph of this chunk alone, cross products by hashing and prune on the host (mrp_phase_reads) */
        n_failed++;
        if (timing) fprintf(stderr, "  chunk %lld left the resident path: redone on the hashing path\n", (long long) c);
        rc = mrp_phase_reads(ctx, chunks[c], reads[c], n_reads[c], params, NULL, &out[c]);
    }
    tt[5] = now_ms();
    if (timing)
        fprintf(stderr, "mrp_phase_reads_many: setup %.1f ms, merge levels %.1f ms (host prepare %.1f, engine %.1f), download %.1f, "
                        "final sweep %.1f, trace back + genome fragments %.1f\n", tt[1] - tt[0], tt[2] - tt[1], g_t_prepare, g_t_level,
                tt[3] - tt[2], tt[4] - tt[3], tt[5] - tt[4]);
    if (timing) {
        int64_t cached = 0, held = 0;
        mrp_context_pool_bytes(ctx, &cached, &held);
        fprintf(stderr, "  device memory: %.1f GB idle in this context's pool, %.1f GB held by all pools of the device\n", (double) cached * 1e-9, (double) held * 1e-9);
    }
    if (timing)
        fprintf(stderr, "  prepare (summed over threads): components %.1f ms, tiling paths %.1f, cross shadows %.1f, garbage %.1f\n",
                g_ns[0] * 1e-6, g_ns[1] * 1e-6, g_ns[2] * 1e-6, g_ns[3] * 1e-6);
    if (stats) {
        mrp_engine_stats es;
        mrp_engine_get_stats(e, &es);
        stats->resident = 1;
        stats->fallback_chunks = (int32_t) n_failed;
        stats->levels = es.levels; stats->hmms = es.hmms; stats->columns = es.columns; stats->cells = es.cells;
        stats->merge_cells = es.merge_cells;
        stats->device_ms = es.device_ms; stats->cross_ms = es.cross_ms; stats->sweep_ms = es.sweep_ms; stats->prune_ms = es.prune_ms;
        stats->pack_ms = es.pack_ms; stats->cross_emit_ms = es.cross_emit_ms; stats->recursion_ms = es.recursion_ms; stats->prune_kernel_ms = es.prune_kernel_ms; stats->compact_ms = es.compact_ms;
    }
    const double t_clean = now_ms();
    for (int64_t c = 0; c < n_chunks; c++) {
        rhmm_destroy(st[c].hmm); free(st[c].discarded); free(st[c].tree.a); free(st[c].path); free(st[c].chosen);
        free(st[c].by_pool); free(st[c].frag_sites); free(st[c].frag_reads1); free(st[c].frag_reads2);
    }
    r_free_tree(&tree);
    for (int64_t c = 0; c < n_chunks; c++) if (st[c].leaves) shadow_release(st[c].leaves, st[c].leaves_class);
    free(st);
    const double t_eng = now_ms();
    mrp_engine_destroy(e);
    if (timing)
        fprintf(stderr, "  engine create %.1f ms, host clean-up %.1f, engine destroy %.1f, whole call %.1f (ends at %.1f ms on the process clock)\n", tt[0] - t_enter, t_eng - t_clean,
                now_ms() - t_eng, now_ms() - t_enter, fmod(now_ms(), 1e5));
    return rc;
}


/* one concurrent batch of mrp_phase_reads_many: while its levels wait for the device, the other batch's host work runs */
typedef struct {
    mrp_context *ctx;
    int64_t n;
    const mrp_chunk **chunks;
    const mrp_read **reads;
    int64_t *n_reads;
    const mrp_params *params;
    mrp_phase_result **out;
    mrp_phase_many_stats stats;
    int rc, index;
    void *pool; /* the caller's host worker pool */
    char err[256];
} phase_group;
long long mrp_pool_task_cpu_ns(void);
long long mrp_pool_task_cpu_ns_this_thread(void);
static double thread_cpu_ms(void) { struct timespec t; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t); return 1e3 * t.tv_sec + 1e-6 * t.tv_nsec; }
/* The batch (0 .. G - 1) every chunk of a call goes to: a repeating pattern that gives batch g the share w_g / sum w of the chunks.
 * For calls of large chunks the first batches are the smaller ones (shares 2 : 3 : 4 : 5 : 5 ...): every batch starts with merge levels
 * that cost the host more than the device, the batches leave them one after the other (the pool serves batch 0 first), and the device
 * waits for the first batch to reach its large levels -- a small first batch gets there sooner, the later ones are prepared beside its
 * kernels (-1 to -2 % per call of 1 152 configs[1] chunks, A/B on three boxes).  Chunks of a few hundred sites keep equal shares: their
 * calls are the host's time throughout, and a larger last batch only lengthens them (640 chunks of 130 sites: 16.2 ms with equal shares,
 * 17.6 with graded ones).  MRP_GROUP_WEIGHTS=w0:w1:... (development) sets the shares.  A work queue's chunk block is uploaded in the
 * same groups (mrp_chunk_block_create): a batch waits for its own group's copy only. */
void mrp_phase_group_assign(int64_t n_chunks, int G, int64_t total_sites, uint8_t *group_of) {
    int w[16], W = 0, pat[256], np = 0;
    if (G < 1) G = 1;
    if (G > 16) G = 16;
    const int graded = G >= 4 && n_chunks >= 16 * (int64_t) G && total_sites >= 500 * n_chunks;
    for (int g = 0; g < G; g++) w[g] = graded ? (g + 2 < 5 ? g + 2 : 5) : 1;
    const char *we = getenv("MRP_GROUP_WEIGHTS");
    if (we) { int g = 0; for (const char *c = we; *c && g < G; g++) { w[g] = atoi(c); if (w[g] < 1) w[g] = 1; if (w[g] > 8) w[g] = 8; while (*c >= '0' && *c <= '9') c++; if (*c) c++; /* (any separator) */ } }
    for (int g = 0; g < G; g++) W += w[g];
    /* the pattern: round by round, every batch that still has weight left takes one place */
    for (int round = 0; np < W; round++) for (int g = 0; g < G && np < W; g++) if (w[g] > round) pat[np++] = g;
    for (int64_t i = 0; i < n_chunks; i++) group_of[i] = (uint8_t) pat[i % W];
}

static void *phase_group_main(void *p) {
    phase_group *g = p;
    const double cpu0 = thread_cpu_ms();
    const long long pool0 = mrp_pool_task_cpu_ns(), mine0 = mrp_pool_task_cpu_ns_this_thread();
    mrp_pool_adopt(g->pool);
    {   /* batch 0's host loops first: the batches reach their device-heavy levels one after the other.  MRP_POOL_PRIORITY (development):
         * 0 = no priorities (the oldest loop first), k > 1 = batches in groups of k share a priority */
        const char *pe = getenv("MRP_POOL_PRIORITY");
        const int pk = pe ? atoi(pe) : 1;
        mrp_pool_set_priority(pk <= 0 ? 0 : g->index / pk);
    }
    g->rc = phase_many_resident(g->ctx, g->n, g->chunks, g->reads, g->n_reads, g->params, g->out, &g->stats);
    mrp_pool_set_priority(0);
    if (getenv("MRP_TIMING")) {
        fprintf(stderr, "  batch %d: cpu of its own thread %.1f ms (%.1f of it pool tasks it ran itself); pool tasks (all batches, while it ran) %.1f ms; cumulative by loop:", g->index,
                thread_cpu_ms() - cpu0, (mrp_pool_task_cpu_ns_this_thread() - mine0) * 1e-6, (mrp_pool_task_cpu_ns() - pool0) * 1e-6);
        for (int t = 0; t < 12; t++) fprintf(stderr, " %d:%.0f", t, mrp_pool_tag_cpu_ns(t) * 1e-6);
        fprintf(stderr, "\n");
    }
    if (g->rc != MRP_OK) snprintf(g->err, sizeof(g->err), "%s", mrp_last_error());
    return NULL;
}

/* the chunks of a call that cannot take the resident path, pulled one at a time by up to eight host threads */
typedef struct {
    mrp_context *ctx;
    const mrp_chunk *const *chunks;
    const mrp_read *const *reads;
    const int64_t *n_reads;
    const mrp_params *params;
    mrp_phase_result **out;
    int64_t next;
    int rc;
    char err[256];
    int threads;
    int64_t n;
} hashing_ctl;
typedef struct { hashing_ctl *ctl; mrp_context *ctx; } hashing_arg;
static void *hashing_main(void *p) {
    hashing_arg *a = p;
    hashing_ctl *hc = a->ctl;
    for (;;) {
        const int64_t c = __atomic_fetch_add(&hc->next, 1, __ATOMIC_RELAXED);
        if (c >= hc->n || __atomic_load_n(&hc->rc, __ATOMIC_RELAXED) != MRP_OK) return NULL;
        const int rc = mrp_phase_reads(a->ctx, hc->chunks[c], hc->reads[c], hc->n_reads[c], hc->params, NULL, &hc->out[c]);
        if (rc != MRP_OK) {
            int expect = MRP_OK;
            if (__atomic_compare_exchange_n(&hc->rc, &expect, rc, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) snprintf(hc->err, sizeof(hc->err), "%s", mrp_last_error());
            return NULL;
        }
    }
}

/* the concurrent batches a call over n_chunks chunks is split into (chunk i goes to batch i % G) */
int mrp_phase_groups_for(const mrp_context *ctx, int64_t n_chunks) {
    int G = mrp_context_phase_groups(ctx); /* mrp_context_set_phase_groups; 0 (default): by batch size */
    if (G <= 0) G = n_chunks < 192 ? (int) (n_chunks / 12 > 4 ? 4 : n_chunks / 12) : 8;
    if (G < 1) G = 1;
    if (G > 16) G = 16;
    if (n_chunks < 4 * G) G = 1;
    return G;
}

static int phase_many_once(mrp_context *ctx, int64_t n_chunks, const mrp_chunk *const *chunks, const mrp_read *const *reads,
                           const int64_t *n_reads, const mrp_params *params, mrp_phase_result **out, mrp_phase_many_stats *stats);

static void stats_add(mrp_phase_many_stats *stats, const mrp_phase_many_stats *st, int first) {
    if (first) { *stats = *st; return; }
    stats->resident = stats->resident && st->resident; stats->fallback_chunks += st->fallback_chunks;
    if (st->levels > stats->levels) stats->levels = st->levels;
    stats->hmms += st->hmms; stats->columns += st->columns; stats->cells += st->cells; stats->merge_cells += st->merge_cells;
    stats->device_ms += st->device_ms; stats->cross_ms += st->cross_ms; stats->sweep_ms += st->sweep_ms; stats->prune_ms += st->prune_ms;
    stats->pack_ms += st->pack_ms; stats->cross_emit_ms += st->cross_emit_ms; stats->recursion_ms += st->recursion_ms; stats->prune_kernel_ms += st->prune_kernel_ms; stats->compact_ms += st->compact_ms;
    if (st->note[0] && !stats->note[0]) memcpy(stats->note, st->note, sizeof(stats->note));
}

/* A call of at most `cap` (read, site) units at a time: what a call keeps on the device grows with its units (measured 214 GB
 * for 1 152 chunks of 60 000 units, ~3.1 KB per unit: the cells of the widest merge level of every concurrent batch), so a call
 * beyond the device's budget runs as consecutive slices that fit.  The estimate is only that: when the driver still refuses
 * an allocation (memory held by another process, a pool grown by best-fit reuse) the slice is redone as two halves after
 * every cache of the context has been given back -- down to single chunks -- instead of failing the call. */
static int phase_many_capped(mrp_context *ctx, int64_t n_chunks, const mrp_chunk *const *chunks, const mrp_read *const *reads,
                             const int64_t *n_reads, const mrp_params *params, mrp_phase_result **out, mrp_phase_many_stats *stats,
                             int64_t cap, int depth) {
    int64_t total = 0;
    for (int64_t c = 0; c < n_chunks; c++)
        for (int64_t r = 0; r < n_reads[c]; r++) total += reads[c][r].length;
    if (total <= cap || n_chunks <= 1) {
        const uint64_t oom0 = mrp_context_oom_events(ctx);
        int rc = phase_many_once(ctx, n_chunks, chunks, reads, n_reads, params, out, stats);
        if (rc == MRP_ERR_HIP && n_chunks > 1 && depth < 8 && mrp_context_oom_events(ctx) != oom0) {
            static int warned;
            if (!__atomic_exchange_n(&warned, 1, __ATOMIC_RELAXED) && !getenv("MRP_QUIET"))
                fprintf(stderr, "margin_rphmm: the device refused memory for a call of %lld chunks (%lld units): redone in two halves\n",
                        (long long) n_chunks, (long long) total);
            for (int64_t c = 0; c < n_chunks; c++) { mrp_phase_result_destroy(out[c]); out[c] = NULL; }
            mrp_context_trim(ctx);
            return phase_many_capped(ctx, n_chunks, chunks, reads, n_reads, params, out, stats, total / 2 + 1, depth + 1);
        }
        return rc;
    }
    int rc = MRP_OK;
    int64_t c0 = 0;
    if (stats) memset(stats, 0, sizeof(*stats));
    while (c0 < n_chunks && rc == MRP_OK) {
        int64_t c1 = c0, u = 0;
        while (c1 < n_chunks) {
            int64_t uc = 0;
            for (int64_t r = 0; r < n_reads[c1]; r++) uc += reads[c1][r].length;
            if (c1 > c0 && u + uc > cap) break;
            u += uc; c1++;
        }
        mrp_phase_many_stats st;
        rc = phase_many_capped(ctx, c1 - c0, chunks + c0, reads + c0, n_reads + c0, params, out + c0, &st, cap, depth);
        if (stats && rc == MRP_OK) stats_add(stats, &st, c0 == 0);
        c0 = c1;
    }
    if (rc != MRP_OK)
        for (int64_t c = 0; c < n_chunks; c++) { mrp_phase_result_destroy(out[c]); out[c] = NULL; }
    return rc;
}

int mrp_phase_reads_many(mrp_context *ctx, int64_t n_chunks, const mrp_chunk *const *chunks, const mrp_read *const *reads,
                         const int64_t *n_reads, const mrp_params *params, mrp_phase_result **out,
                         mrp_phase_many_stats *stats) {
    if (!ctx || n_chunks < 0 || !params || (n_chunks > 0 && (!chunks || !reads || !n_reads || !out)))
        return mrp_set_error(MRP_ERR_ARG, "mrp_phase_reads_many: bad arguments");
    if (params->reserved != 0) return mrp_set_error(MRP_ERR_ARG, "mrp_params.reserved must be 0");
    if (stats) memset(stats, 0, sizeof(*stats));
    for (int64_t c = 0; c < n_chunks; c++) out[c] = NULL;
    /* the slice size from the device's budget (free memory when the process's first pool asked, mrp_internal.h); MRP_CALL_UNITS
     * overrides it (tests) */
    const int64_t budget = mrp_context_device_budget(ctx);
    int64_t cap = budget > 0 ? budget / 3400 : (int64_t) 7e7;
    const char *ce = getenv("MRP_CALL_UNITS");
    if (ce && atoll(ce) > 0) cap = atoll(ce);
    return phase_many_capped(ctx, n_chunks, chunks, reads, n_reads, params, out, stats, cap, 0);
}

static int phase_many_once(mrp_context *ctx, int64_t n_chunks, const mrp_chunk *const *chunks, const mrp_read *const *reads,
                           const int64_t *n_reads, const mrp_params *params, mrp_phase_result **out, mrp_phase_many_stats *stats) {
    if (stats) memset(stats, 0, sizeof(*stats));
    for (int64_t c = 0; c < n_chunks; c++) out[c] = NULL;
    /* the levels of a batch alternate host work (structure, descriptors) and device work; two interleaved halves of the
     * chunks, each with its own context and host thread, keep both busy */
    int G = mrp_phase_groups_for(ctx, n_chunks);
    /* measured on MI355X (bench.py --chunks N --phase-groups G, two streams a batch): 48 chunks 45.3 ms with 2 batches, 42.5
     * with 4; 96: 52.3 with 4, 54.8 with 8; 144: 66.4 / 68.0; 192: 80.2 / 78.6; 288: 104.5 / 96.9; 432: 139.9 with 6, 130.5
     * with 8; 576 with 8: 169.5 (2.04e8 units/s, the best rate; 768: 243 ms).  More than 8 would share hardware queues. */
    int rc = MRP_OK;
    if (G == 1) {
        rc = phase_many_resident(ctx, n_chunks, chunks, reads, n_reads, params, out, stats);
    } else {
        phase_group *grp = xcalloc((size_t) G, sizeof(*grp));
        pthread_t th[16];
        int started[16] = {0};
        /* which batch a chunk goes to (mrp_phase_group_assign: graded shares for calls of large chunks) */
        uint8_t *group_of = xmalloc((size_t) n_chunks + 1);
        {
            int64_t sites = 0;
            for (int64_t i = 0; i < n_chunks; i++) { mrp_chunk_host hv; mrp_chunk_host_view(chunks[i], &hv); sites += hv.n_sites; }
            mrp_phase_group_assign(n_chunks, G, sites, group_of);
        }
        for (int g = 0; g < G; g++) {
            phase_group *q = &grp[g];
            q->index = g;
            q->pool = mrp_pool_current();
            q->ctx = g == 0 ? ctx : mrp_context_sibling(ctx, g - 1);
            q->params = params;
            q->n = 0;
            for (int64_t i = 0; i < n_chunks; i++) if (group_of[i] == g) q->n++;
            q->chunks = xmalloc(sizeof(*q->chunks) * (size_t) (q->n + 1));
            q->reads = xmalloc(sizeof(*q->reads) * (size_t) (q->n + 1));
            q->n_reads = xmalloc(sizeof(*q->n_reads) * (size_t) (q->n + 1));
            q->out = xcalloc((size_t) q->n + 1, sizeof(*q->out));
            q->n = 0;
            for (int64_t i = 0; i < n_chunks; i++)
                if (group_of[i] == g) { q->chunks[q->n] = chunks[i]; q->reads[q->n] = reads[i]; q->n_reads[q->n] = n_reads[i]; q->n++; }
            if (!q->ctx) { q->rc = MRP_ERR_HIP; snprintf(q->err, sizeof(q->err), "%s", mrp_last_error()); }
        }
        const int was_grouped = mrp_context_set_grouped(ctx, 1); /* (the siblings always are) */
        for (int g = 0; g < G; g++) if (grp[g].ctx) mrp_context_set_concurrent_batches(grp[g].ctx, G * mrp_context_calls_sharing_device(ctx));
        mrp_warn_hw_queues_once(G);
        for (int g = 1; g < G; g++)
            if (grp[g].ctx && pthread_create(&th[g], NULL, phase_group_main, &grp[g]) == 0) started[g] = 1;
        if (grp[0].ctx) phase_group_main(&grp[0]);
        for (int g = 1; g < G; g++) {
            if (started[g]) pthread_join(th[g], NULL);
            else if (grp[g].ctx) phase_group_main(&grp[g]); /* thread creation failed: run it here */
        }
        mrp_context_set_grouped(ctx, was_grouped);
        for (int g = 0; g < G; g++) if (grp[g].ctx) mrp_context_set_concurrent_batches(grp[g].ctx, 1);
        for (int g = 0; g < G; g++) {
            phase_group *q = &grp[g];
            if (q->rc != MRP_OK && (rc == MRP_OK || rc == MRP_ERR_UNSUPPORTED)) rc = mrp_set_error(q->rc, "%s", q->err);
            { int64_t k = 0; for (int64_t i = 0; i < n_chunks; i++) if (group_of[i] == g) out[i] = q->out[k++]; }
            if (stats && q->rc == MRP_OK) {
                stats->resident = 1;
                stats->fallback_chunks += q->stats.fallback_chunks;
                stats->levels = q->stats.levels > stats->levels ? q->stats.levels : stats->levels;
                stats->hmms += q->stats.hmms; stats->columns += q->stats.columns; stats->cells += q->stats.cells;
                stats->merge_cells += q->stats.merge_cells;
                stats->device_ms += q->stats.device_ms; stats->cross_ms += q->stats.cross_ms; stats->sweep_ms += q->stats.sweep_ms;
                stats->prune_ms += q->stats.prune_ms;
                stats->pack_ms += q->stats.pack_ms; stats->cross_emit_ms += q->stats.cross_emit_ms; stats->recursion_ms += q->stats.recursion_ms;
                stats->prune_kernel_ms += q->stats.prune_kernel_ms; stats->compact_ms += q->stats.compact_ms;
            }
            free(q->chunks); free(q->reads); free(q->n_reads); free(q->out);
        }
        free(grp);
        free(group_of);
    }
    if (rc == MRP_ERR_UNSUPPORTED) {
        /* Parameters or hmm shapes outside the resident path: the hashing path (mrp_phase_reads), one chunk per host thread,
         * each thread with a context of its own.  Said loudly: this is two orders of magnitude slower than the resident path. */
        char why[160];
        snprintf(why, sizeof(why), "%s", mrp_last_error());
        if (stats) { memset(stats, 0, sizeof(*stats)); snprintf(stats->note, sizeof(stats->note), "%s", why); }
        static int warned;
        if (!__atomic_exchange_n(&warned, 1, __ATOMIC_RELAXED) && !getenv("MRP_QUIET"))
            fprintf(stderr, "margin_rphmm: mrp_phase_reads_many leaves the device-resident path (%s): %lld chunk(s) take the per-chunk hashing path, "
                            "about 100x slower per chunk\n", why, (long long) n_chunks);
        rc = MRP_OK;
        for (int64_t c = 0; c < n_chunks; c++) { mrp_phase_result_destroy(out[c]); out[c] = NULL; }
        int T = mrp_host_threads();
        if (T > 8) T = 8;
        if (T > n_chunks) T = (int) n_chunks;
        if (T < 1) T = 1;
        hashing_ctl hc = {ctx, chunks, reads, n_reads, params, out, 0, MRP_OK, {0}, T, 0};
        for (int t = 1; t < T && rc == MRP_OK; t++)
            if (!mrp_context_sibling(ctx, t - 1)) rc = MRP_ERR_HIP; /* all contexts before the first thread (allocator peers) */
        if (rc == MRP_OK) {
            hc.n = n_chunks;
            pthread_t th[8];
            int started[8] = {0};
            hashing_arg ha[8];
            for (int t = 0; t < T; t++) { ha[t].ctl = &hc; ha[t].ctx = t == 0 ? ctx : mrp_context_sibling(ctx, t - 1); }
            for (int t = 1; t < T; t++) started[t] = pthread_create(&th[t], NULL, hashing_main, &ha[t]) == 0;
            hashing_main(&ha[0]);
            for (int t = 1; t < T; t++) if (started[t]) pthread_join(th[t], NULL);
            rc = hc.rc;
            if (rc != MRP_OK) mrp_set_error(rc, "%s", hc.err);
        }
    }
    if (rc != MRP_OK)
        for (int64_t c = 0; c < n_chunks; c++) { mrp_phase_result_destroy(out[c]); out[c] = NULL; }
    return rc;
}
