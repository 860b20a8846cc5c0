/* hostbench.c -- CPU-only profile of the HOST side of the device-resident pipeline (rphmm_host.c): the engine is replaced by
 * stubs that only hand out segment numbers, so that tiling paths, overlap components, column boundaries, final shadows and
 * genome fragments can be timed (gprof) without a GPU.  Development tool; build: make -C tools/hostbench. */
#define _GNU_SOURCE
#include <stdarg.h>
#include <pthread.h>
#include "../../margin_amd/csrc/rphmm_host.c"

struct mrp_chunk { mrp_chunk_host h; };
struct mrp_engine { int n_segs; int64_t cols; uint64_t hash; };
static uint64_t g_hash; /* of everything the engine is told: a structural change that alters a level shows here */
static inline uint64_t hmix(uint64_t h, uint64_t v) { h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2); return h * 0xff51afd7ed558ccdULL; }
static __thread char g_err[512];
void mrp_chunk_host_view(const mrp_chunk *chunk, mrp_chunk_host *out) { *out = chunk->h; }
mrp_context *mrp_chunk_context(const mrp_chunk *chunk) { (void) chunk; return (mrp_context *) 8; }
int mrp_context_device(const mrp_context *ctx) { (void) ctx; return 0; }
int mrp_context_set_grouped(mrp_context *ctx, int grouped) { (void) ctx; (void) grouped; return 0; }
void mrp_context_set_concurrent_batches(mrp_context *ctx, int n) { (void) ctx; (void) n; }
int mrp_context_calls_sharing_device(const mrp_context *ctx) { (void) ctx; return 1; }
int64_t mrp_context_device_budget(mrp_context *ctx) { (void) ctx; return 0; }
uint64_t mrp_context_oom_events(mrp_context *ctx) { (void) ctx; return 0; }
void mrp_warn_hw_queues_once(int n) { (void) n; }
int mrp_context_trim(mrp_context *ctx) { (void) ctx; return 0; }
void mrp_context_pool_bytes(mrp_context *ctx, int64_t *cached, int64_t *device_held) { (void) ctx; *cached = 0; *device_held = 0; }
mrp_context *mrp_context_sibling(mrp_context *ctx, int i) { (void) i; return ctx; }
int mrp_set_error(int code, const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); return code; }
const char *mrp_last_error(void) { return g_err; }
static int g_threads = 1;
int mrp_host_threads(void) { return g_threads; }
int mrp_context_phase_groups(const mrp_context *ctx) { (void) ctx; return 1; }
void mrp_pool_set_priority(int p) { (void) p; }
void mrp_pool_set_tag(int t) { (void) t; }
void mrp_pool_set_weight(int ns) { (void) ns; }
long long mrp_pool_tag_cpu_ns(int tag) { (void) tag; return 0; }
long long mrp_pool_task_cpu_ns(void) { return 0; }
long long mrp_pool_task_cpu_ns_this_thread(void) { return 0; }
void mrp_pool_adopt(void *p) { (void) p; }
void *mrp_pool_current(void) { return NULL; }
typedef struct { int64_t n, grain; void (*fn)(int64_t, void *); void *arg; int64_t next; } pjob;
static void *pworker(void *a) { pjob *j = a; for (;;) { int64_t lo = __atomic_fetch_add(&j->next, j->grain, __ATOMIC_RELAXED); if (lo >= j->n) return NULL;
    int64_t hi = lo + j->grain < j->n ? lo + j->grain : j->n; for (int64_t i = lo; i < hi; i++) j->fn(i, j->arg); } }
void mrp_pool_run(int64_t n, int64_t grain, void (*fn)(int64_t, void *), void *arg) {
    if (g_threads <= 1) { for (int64_t i = 0; i < n; i++) fn(i, arg); return; }
    pjob j = {n, grain < 1 ? 1 : grain, fn, arg, 0}; if (getenv("HB_DEBUG")) fprintf(stderr, "pool_run n=%lld grain=%lld threads=%d\n", (long long) n, (long long) grain, g_threads);
    pthread_t th[64];
    for (int t = 1; t < g_threads; t++) pthread_create(&th[t], NULL, pworker, &j);
    pworker(&j);
    for (int t = 1; t < g_threads; t++) pthread_join(th[t], NULL);
}
int mrp_fb_run(mrp_context *ctx, int64_t n, const mrp_hmm_job *jobs) { (void) ctx; (void) n; (void) jobs; return MRP_ERR_NO_DEVICE; }
int mrp_batch_add(mrp_batch *b, const mrp_hmm_job *job) { (void) b; (void) job; return MRP_ERR_NO_DEVICE; }
int mrp_engine_create(mrp_context *ctx, const mrp_params *params, mrp_engine **out) { (void) ctx; (void) params; *out = calloc(1, sizeof(mrp_engine)); return MRP_OK; }
void mrp_engine_destroy(mrp_engine *e) { g_hash = e->hash; (free)(e); }
int32_t mrp_engine_stride(const mrp_engine *e) { (void) e; return 100; }
int mrp_engine_locate(const mrp_engine *e, int32_t seg, int64_t col0, const uint64_t **a, const uint32_t **b, const int32_t **c, const int32_t **d) { (void) e; (void) seg; (void) col0; *a = NULL; *b = NULL; *c = NULL; *d = NULL; return MRP_OK; }
static int stage(mrp_engine *e, int64_t n, mrp_xhmm *x, int final) {
    int64_t col = 0;
    for (int64_t i = 0; i < n; i++) {
        uint64_t h = e->hash;
        h = hmix(h, (uint64_t) x[i].ref_start); h = hmix(h, (uint64_t) x[i].ref_end); h = hmix(h, (uint64_t) x[i].n_cols); h = hmix(h, (uint64_t) x[i].n_a); h = hmix(h, (uint64_t) x[i].n_b);
        h = hmix(h, (uint64_t) x[i].bound_cells); h = hmix(h, (uint64_t) x[i].bound_merge); h = hmix(h, (uint64_t) x[i].depth_sites); h = hmix(h, (uint64_t) x[i].bound_max_cells); h = hmix(h, (uint64_t) x[i].bound_max_merge);
        for (int k = 0; k < x[i].n_cols; k++) { h = hmix(h, (uint64_t) x[i].col_start[k]); h = hmix(h, (uint64_t) x[i].col_read_off[k + 1]); }
        for (int k = 0; k < x[i].n_a + x[i].n_b; k++) { const mrp_xpar *q = &x[i].par[k]; h = hmix(h, (uint64_t) q->start); h = hmix(h, (uint64_t) q->end); h = hmix(h, (uint64_t) q->n_cols); h = hmix(h, (uint64_t) q->seg); h = hmix(h, (uint64_t) q->col0); }
        e->hash = h;
        x[i].seg = e->n_segs; x[i].col0 = col; col += x[i].n_cols; x[i].err = 0;
        if (final) { for (int k = 0; k < x[i].n_cols; k++) { x[i].n_cells[k] = 0; x[i].path_part[k] = 0; }
            if (x[i].frag_sites) { memset(x[i].frag_sites, 0, 20 * (size_t) (x[i].ref_end - x[i].ref_start)); x[i].frag_n1 = x[i].frag_n2 = 0; x[i].frag_done = 1; } } } /* (as the device leaves it) */
    e->n_segs++; e->cols += col;
    return MRP_OK;
}
int mrp_engine_level_stage(mrp_engine *e, int64_t n, mrp_xhmm *x) { return stage(e, n, x, 0); }
int mrp_engine_final_stage(mrp_engine *e, int64_t n, mrp_xhmm *x) { return stage(e, n, x, 1); }
int mrp_engine_level_launch(mrp_engine *e) { (void) e; return MRP_OK; }
int mrp_engine_level_end(mrp_engine *e) { (void) e; return MRP_OK; }
int64_t mrp_engine_levels_ended(const mrp_engine *e) { return e->n_segs; } /* (the stub's levels end as they are launched) */
int mrp_engine_fetch(mrp_engine *e, void *dst, const void *src, int64_t bytes) { (void) e; (void) dst; (void) src; (void) bytes; return MRP_OK; }
int mrp_engine_sync(mrp_engine *e) { (void) e; return MRP_OK; }
void mrp_engine_get_stats(const mrp_engine *e, mrp_engine_stats *out) { memset(out, 0, sizeof *out); out->columns = e->cols; out->levels = e->n_segs; }

/* --selftest: r_tiling_paths_sorted (one first-fit pass) against the walk of getTilingPaths as the reference does it -- path after
 * path, each time the first unused hmm that starts at or behind the end of the path's last one (coordination.c:19-55, 186-222) -- on
 * random interval sets in stRPHmm_cmpFn order: nested, touching, equal, sparse and 60 deep. */
static int selftest(void) {
    uint64_t rng = 12345;
    #define RND() (rng = rng * 6364136223846793005ull + 1442695040888963407ull, (uint32_t) (rng >> 33))
    for (int round = 0; round < 400; round++) {
        const int n = 1 + (int) (RND() % (round < 200 ? 40 : 2500));
        const int span = 1 + (int) (RND() % 3000), maxlen = 1 + (int) (RND() % (round % 3 ? 400 : 30));
        rhmm *h = calloc((size_t) n, sizeof *h);
        rhmm **sorted = calloc((size_t) n, sizeof *sorted);
        for (int i = 0; i < n; i++) { h[i].ref_start = (int32_t) (RND() % (uint32_t) span); h[i].ref_length = 1 + (int32_t) (RND() % (uint32_t) maxlen); h[i].first_read = -1; sorted[i] = &h[i]; }
        world w; memset(&w, 0, sizeof w);
        r_sort_hmms(&w, sorted, n);
        for (int i = 1; i < n; i++) if (r_hmm_cmp(&w, sorted[i - 1], sorted[i]) > 0) { printf("selftest: sort order broken in round %d\n", round); return 1; }
        r_path_vec got = r_tiling_paths_sorted(sorted, n);
        /* the reference's walk */
        uint8_t *used = calloc((size_t) n, 1);
        int remaining = n, first = 0, path = 0, bad = 0;
        while (remaining > 0 && !bad) {
            while (used[first]) first++;
            int cur = first, k = 0;
            if (path >= got.n) { bad = 1; break; }
            for (;;) {
                if (k >= got.a[path]->n || got.a[path]->a[k] != sorted[cur]) { bad = 1; break; }
                used[cur] = 1; remaining--; k++;
                int nxt = -1;
                for (int j = cur + 1; j < n; j++) if (!used[j] && sorted[cur]->ref_start + sorted[cur]->ref_length <= sorted[j]->ref_start) { nxt = j; break; }
                if (nxt < 0) break;
                cur = nxt;
            }
            if (!bad && k != got.a[path]->n) bad = 1;
            path++;
        }
        if (!bad && path != got.n) bad = 1;
        if (bad) { printf("selftest: tiling paths differ from the reference's walk in round %d (n = %d)\n", round, n); return 1; }
        for (int64_t i = 0; i < got.n; i++) r_free_path(got.a[i], 0);
        free(got.a); free(used); free(sorted); free(h);
    }
    printf("selftest ok\n");
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 2 && strcmp(argv[1], "--selftest") == 0) return selftest();
    if (argc < 2) { fprintf(stderr, "usage: hostbench chunks.bin [repeat] [threads] | hostbench --selftest\n"); return 2; }
    const int repeat = argc > 2 ? atoi(argv[2]) : 3;
    g_threads = argc > 3 ? atoi(argv[3]) : 1;
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 1; }
    int64_t n;
    if (fread(&n, 8, 1, f) != 1) return 1;
    mrp_chunk *chunks = calloc((size_t) n, sizeof *chunks);
    const mrp_chunk **cp = calloc((size_t) n, sizeof *cp);
    const mrp_read **rp = calloc((size_t) n, sizeof *rp);
    int64_t *nr = calloc((size_t) n, sizeof *nr), units = 0;
    for (int64_t c = 0; c < n; c++) {
        int64_t hd[3];
        if (fread(hd, 8, 3, f) != 3) return 1;
        const int64_t ns = hd[0], nreads = hd[1], pb = hd[2];
        uint32_t *an = malloc(4 * (size_t) ns), *ao = malloc(4 * (size_t) (ns + 1)), *so = malloc(4 * (size_t) (ns + 1));
        if (fread(an, 4, (size_t) ns, f) != (size_t) ns) return 1;
        ao[0] = 0; so[0] = 0;
        for (int64_t i = 0; i < ns; i++) { ao[i + 1] = ao[i] + an[i]; so[i + 1] = so[i] + an[i] * an[i]; }
        int32_t *tab = malloc(16 * (size_t) nreads); int64_t *po = malloc(8 * (size_t) nreads);
        if (fread(tab, 16, (size_t) nreads, f) != (size_t) nreads || fread(po, 8, (size_t) nreads, f) != (size_t) nreads) return 1;
        uint8_t *pool = malloc((size_t) pb + 1);
        if (fread(pool, 1, (size_t) pb, f) != (size_t) pb) return 1;
        mrp_read *rd = calloc((size_t) nreads, sizeof *rd);
        for (int64_t i = 0; i < nreads; i++) {
            char *nm = malloc(24); snprintf(nm, 24, "read_%06lld", (long long) i);
            rd[i].name = nm; rd[i].ref_start = tab[4 * i]; rd[i].length = tab[4 * i + 1]; rd[i].forward_strand = tab[4 * i + 2]; rd[i].pool_offset = po[i];
            units += rd[i].length;
        }
        chunks[c].h.n_sites = ns; chunks[c].h.allele_number = an; chunks[c].h.allele_offset = ao; chunks[c].h.sub_offset = so;
        chunks[c].h.sub = calloc(so[ns] + 1, 2); chunks[c].h.prior = calloc(ao[ns] + 1, 2); chunks[c].h.pool = pool; chunks[c].h.pool_bytes = pb;
        cp[c] = &chunks[c]; rp[c] = rd; nr[c] = nreads;
    }
    fclose(f);
    mrp_params P; memset(&P, 0, sizeof P);
    P.max_not_sum_transitions = 1; P.include_inverted_partitions = 1; P.include_ancestor_sub_prob = 1;
    P.min_partitions_in_a_column = 100; P.max_partitions_in_a_column = 100; P.min_posterior_probability_for_partition = 0.0;
    P.max_coverage_depth = 64; P.min_read_coverage_to_support_phasing_between_heterozygous_sites = 2; P.rounds_of_iterative_refinement = 10;
    mrp_phase_result **out = calloc((size_t) n, sizeof *out);
    for (int r = 0; r < repeat; r++) {
        mrp_phase_many_stats st;
        struct timespec a, b, ca, cb;
        clock_gettime(CLOCK_MONOTONIC, &a); clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ca);
        int rc = phase_many_resident((mrp_context *) 8, n, cp, rp, nr, &P, out, &st);
        clock_gettime(CLOCK_MONOTONIC, &b); clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &cb);
        printf("run %d: rc %d (%s) %lld chunks, %lld units, wall %.1f ms, cpu %.1f ms, columns %lld, levels %lld, structure hash %016llx\n", r, rc, rc ? g_err : "ok", (long long) n, (long long) units,
               1e3 * (b.tv_sec - a.tv_sec) + 1e-6 * (b.tv_nsec - a.tv_nsec), 1e3 * (cb.tv_sec - ca.tv_sec) + 1e-6 * (cb.tv_nsec - ca.tv_nsec), (long long) st.columns, (long long) st.levels, (unsigned long long) g_hash);
        for (int64_t c = 0; c < n; c++) { mrp_phase_result_destroy(out[c]); out[c] = NULL; }
    }
    return 0;
}
