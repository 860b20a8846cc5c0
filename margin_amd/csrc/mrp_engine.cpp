/*
 * mrp_engine.cpp -- host side of the device-resident merge level.
 *
 * One level performs, for a set of independent overlap components (coordination.c:285-328), what the reference does
 * with
 *     stRPHmm_createCrossProductOfTwoAlignedHmm (hmm.c:534)  ->  mrp_cross_kernel
 *     stRPHmm_forwardBackward (hmm.c:931)                    ->  packing / emission / recursion kernels
 *     stRPHmm_prune (hmm.c:1160)                             ->  mrp_prune_kernel + mrp_compact_kernel
 * without the hmm leaving HBM: the parents are read from, and the pruned result is written to, the fixed-stride
 * resident layout of mrp_engine.h.
 *
 * What the host contributes to a level does not depend on any result of the level before: the column structure of the
 * cross products (sites, reads, connector kinds) and the ADDRESSES at which the parents' per-column counts will be
 * found.  A level therefore goes through three steps:
 *     stage   host only: the static description (PlanCol / PlanHmm, read offsets, launch classes from static bounds) is
 *             built by the worker pool and uploaded on the copy stream -- while the level before is still running;
 *     launch  the layout kernels turn the parents' counts into sizes, offsets and every kernel descriptor; four totals
 *             come back (the only host wait of a level, it also ends the level before), the cell arrays are allocated
 *             and the level's kernels queued;
 *     end     per-hmm error flags (and, at the final level, the traced-back path) are read.
 * The structural decisions (tiling paths, overlap components, column alignment) are made by rphmm_host.c from read
 * intervals alone.
 */
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <memory>
#include <new>
#include <vector>

#include "mrp_engine.h"
extern "C" void mrp_pool_set_tag(int t);
extern "C" void mrp_pool_set_weight(int ns_per_index);
#include "mrp_internal.h"

#define ENG_TRY(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) return mrp_set_error(MRP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace {
struct Segment { /* the pruned hmms produced by one level, and their column structure (read by the levels above) */
    DevBuf<uint64_t> part;
    DevBuf<uint32_t> np;
    DevBuf<int32_t> n_cells, n_merge;
    DevBuf<ResCol> cols;
    DevBuf<int64_t> rbo;
};

}  // namespace

/* everything one level keeps between its staging and its completion */
struct mrp_engine_level_state {
    mrp_batch *b = nullptr;
    std::unique_ptr<Segment> seg;
    /* static description, device side */
    DevBuf<PlanCol> d_plan;
    DevBuf<XDesc> d_xd;
    DevBuf<mrp_xpar> d_par;
    DevBuf<int32_t> d_cstart, d_croff;
    DevBuf<PlanHmm> d_phmm;
    DevBuf<uint16_t> d_dims;
    DevBuf<LayoutTot> d_tot;
    DevBuf<LayoutBase> d_base;
    DevBuf<int64_t> d_totals, d_tile_sums;
    DevBuf<CrossCol> d_cc;
    DevBuf<PruneHmm> d_ph;
    DevBuf<int32_t> d_col_hmm, d_nkept, d_nkeptm, d_err, d_err_hmm;
    DevBuf<uint16_t> d_kept, d_keptm;
    DevBuf<uint32_t> d_kept_np;
    PinnedBuf stage;        /* host side of the uploads */
    PinnedBuf results;      /* totals, error flags, final level: path and totals of the sweep */
    int64_t *totals = nullptr;
    int32_t *err = nullptr, *err_hmm = nullptr, *path_cell = nullptr;
    uint64_t *path_part = nullptr;
    double *fb = nullptr;
    std::vector<int32_t> perm; /* position in the (sorted) PruneHmm array -> index into x */
    bool final_level = false;
    bool any_pack = true, any_planes = true; /* columns for the byte packing kernel / the bit plane kernel */
    int seg_id = -1;
    bool fused = false; /* cross product and emission in one kernel, no partition array (merge levels, no ancestor model) */
    bool units = false; /* the level's cell / merge cell arrays hold one entry per complement pair (MRP_XF_UNITS) */
    int64_t n_mini = 0; /* hmms of the single-wave kernel: the last n_mini records of the level's PruneHmm array */
    /* genome fragments of a final level on the device (mrp_fragment_kernel): inputs staged, results fetched with the level's */
    bool frag = false;
    PinnedBuf frag_stage, frag_results;
    DevBuf<FragHmm> d_frag_hmms; DevBuf<FragRead> d_frag_reads; DevBuf<int32_t> d_frag_by_pool, d_frag_disc, d_frag_lists, d_frag_work, d_frag_counts, d_frag_col_read, d_frag_col_cnt;
    DevBuf<FragSite> d_frag_sites; DevBuf<uint64_t> d_frag_col_part; DevBuf<uint32_t> d_frag_read_key;
    std::vector<FragHmm> frag_hmms;
    int64_t frag_reads_total = 0, frag_disc_total = 0, frag_sites_total = 0, frag_list_total = 0;
    FragSite *h_frag_sites = nullptr; int32_t *h_frag_lists = nullptr, *h_frag_counts = nullptr;
    unsigned long long clk[12] = {0};
    mrp_xhmm *x = nullptr;
    int64_t n = 0, total_cols = 0, n_slots = 0, n_reads = 0;
    PruneParams pp{};
    double t_begin = 0, t_staged = 0, t_launch_ms = 0, t_react_ms = 0, t_launched = 0; /* t_react_ms: from the totals' arrival to the last kernel queued */
    hipEvent_t lay0 = nullptr; /* in front of the layout kernels (timing) */
    hipEvent_t uploaded = nullptr; /* end of the uploads on the copy stream */
    hipEvent_t done = nullptr;     /* behind everything the level queued on the main stream (its results' copies included) */
    bool deferred = false;         /* launched without waiting for its totals: arrays sized by the static bounds (level_launch_impl) */
    int64_t bound_cells = 0, bound_merge = 0; /* sums of the hmms' static bounds (cells padded to a multiple of 4 per hmm) */
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; /* before / after the cross product, after the sweeps, after the compaction, [4] after the prune */
    ~mrp_engine_level_state() {
        if (uploaded) (void) hipEventDestroy(uploaded);
        if (done) (void) hipEventDestroy(done);
        if (lay0) (void) hipEventDestroy(lay0);
        for (auto &e_ : ev)
            if (e_) (void) hipEventDestroy(e_);
        if (b) mrp_batch_destroy(b);
    }
};

struct mrp_engine {
    mrp_context *ctx = nullptr;
    mrp_params params{};
    PruneParams pp{};
    DevBuf<uint64_t> leaf_part;
    DevBuf<uint32_t> leaf_np;
    DevBuf<int32_t> leaf_count;
    std::vector<std::unique_ptr<Segment>> segments;
    /* every staged level gets a segment number; its arrays are listed here (host copy + device table) for the levels above */
    static constexpr int MAX_SEGS = 64;
    SegDev segtab[MAX_SEGS] = {};
    int n_segs = 0;
    DevBuf<SegDev> d_segs;
    mrp_engine_stats stats{};
    mrp_engine_level_state *staged = nullptr;  /* staged, not launched */
    /* launched, not ended, oldest first.  A level is normally ended by the launch of the next one (the one host wait of a level: the
     * exact sizes of the next level's arrays come back with it).  Small levels -- a call of a few chunks, the first merge levels --
     * are launched WITHOUT that wait (deferred: arrays sized by the static bounds) and several of them are in flight at a time;
     * they are ended, in order, as their events complete. */
    std::vector<mrp_engine_level_state *> inflight;
    double diag_layout_gap_ms = 0, diag_react_ms = 0;
    std::vector<std::pair<long long, double>> timeline; /* MRP_TIMELINE: (hmms of the level, host time of its launch) */
    double t_created = 0;
    int64_t n_ended = 0;         /* levels ended so far (mrp_engine_levels_ended: the caller settles what it keeps per level) */
    int64_t inflight_bytes = 0;  /* device bytes of the deferred levels in flight (by their bounds) */
    std::vector<mrp_batch *> spare;            /* emptied batch objects (host arrays keep their capacity) */
    std::vector<mrp_engine_level_state *> spare_levels; /* level states whose pinned buffers are reused */
};

static double eng_now() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return 1e3 * ts.tv_sec + 1e-6 * ts.tv_nsec;
}

/* back to the empty state; the batch object and the pinned buffers stay with the engine for the next level */
static void level_retire(mrp_engine *e, mrp_engine_level_state *L, bool complete = false) {
    if (!L) return;
    mrp_context *ctx = e->ctx;
    (void) hipSetDevice(ctx->device);
    if (!complete) { /* (complete: the level's `done` event has been seen -- everything that touches its buffers is over, while later
                      *  levels may still be queued on the stream) */
        (void) ctx->wait_stream(ctx->stream); /* before the buffers go back to the pool */
        if (ctx->pre) (void) ctx->wait_stream(ctx->pre);
    }
    if (L->b) {
        L->b->recycle();
        e->spare.push_back(L->b);
        L->b = nullptr;
    }
    L->seg.reset();
    L->d_plan.release(); L->d_xd.release(); L->d_par.release(); L->d_cstart.release(); L->d_croff.release(); L->d_phmm.release(); L->d_dims.release(); L->d_tot.release(); L->d_base.release(); L->d_totals.release(); L->d_tile_sums.release();
    L->d_frag_hmms.release(); L->d_frag_reads.release(); L->d_frag_by_pool.release(); L->d_frag_disc.release(); L->d_frag_lists.release(); L->d_frag_work.release();
    L->d_frag_counts.release(); L->d_frag_col_read.release(); L->d_frag_col_cnt.release(); L->d_frag_sites.release(); L->d_frag_col_part.release(); L->d_frag_read_key.release();
    L->frag = false;
    L->d_cc.release(); L->d_ph.release(); L->d_col_hmm.release(); L->d_nkept.release(); L->d_nkeptm.release(); L->d_err.release();
    L->d_err_hmm.release(); L->d_kept.release(); L->d_keptm.release(); L->d_kept_np.release();
    L->perm.clear();
    L->x = nullptr; L->n = 0;
    if (L->deferred) { e->inflight_bytes -= 16 * L->bound_cells + 8 * L->bound_merge; L->deferred = false; }
    e->spare_levels.push_back(L);
    ctx->pool.reclaim();
}

extern "C" {

int mrp_engine_create(mrp_context *ctx, const mrp_params *params, mrp_engine **out) {
    if (!ctx || !params || !out) return mrp_set_error(MRP_ERR_ARG, "mrp_engine_create: NULL argument");
    *out = nullptr;
    if (params->reserved != 0) return mrp_set_error(MRP_ERR_ARG, "mrp_params.reserved must be 0");
    if (!params->max_not_sum_transitions)
        return mrp_set_error(MRP_ERR_UNSUPPORTED, "the device-resident merge needs maxNotSumTransitions (integer posteriors)");
    const int64_t lim = std::max<int64_t>(params->min_partitions_in_a_column, params->max_partitions_in_a_column);
    if (lim < 1 || lim > MRP_PRUNE_MAX_S || params->min_partitions_in_a_column < 0)
        return mrp_set_error(MRP_ERR_UNSUPPORTED, "the device-resident merge keeps at most %d partitions per column", MRP_PRUNE_MAX_S);
    ENG_TRY(hipSetDevice(ctx->device));
    mrp_engine *e = new (std::nothrow) mrp_engine();
    if (!e) return mrp_set_error(MRP_ERR_NOMEM, "out of host memory");
    e->ctx = ctx;
    e->params = *params;
    e->t_created = eng_now();
    PruneParams &pp = e->pp;
    pp.S = (int32_t) ((lim + 3) & ~3ll);
    pp.min_p = (int32_t) params->min_partitions_in_a_column;
    pp.max_p = (int32_t) std::min<int64_t>(params->max_partitions_in_a_column, 1 << 20);
    if (pp.max_p < 0) pp.max_p = 0;
    /* the prune chain on complement pairs (mrp_prune_kernel<..., PAIRS>): every cell has its twin and the limits never cut
     * a pair.  MRP_PRUNE_PAIRS=0 (development) and test hook bit 3 keep the general chain for A/B parity. */
    pp.pairs = params->include_inverted_partitions && (pp.min_p & 1) == 0 && (pp.max_p & 1) == 0 ? 1 : 0;
    if (const char *pe = getenv("MRP_PRUNE_PAIRS")) { if (pe[0] == '0') pp.pairs = 0; }
    /* posterior = min(1, exp(s)), s = f + b - total an integer <= 0 (column.c:177-193).  Ranking by the
     * integer is ranking by the double as long as consecutive integers give distinct doubles; from the
     * underflow point of exp down every posterior is 0.0 and they all tie: that is the last bin. */
    int zero_bin = 0;
    while (zero_bin < 4096 && exp(-(double) zero_bin) > 0.0) zero_bin++;
    for (int b2 = 1; b2 <= zero_bin; b2++)
        if (!(exp(-(double) b2) < exp(-(double) (b2 - 1)))) {
            delete e;
            return mrp_set_error(MRP_ERR_UNSUPPORTED, "exp() is not strictly monotone on the integers at %d", -b2);
        }
    pp.n_bins = zero_bin + 1;
    pp.thr_bin = -1;
    for (int b2 = 0; b2 < pp.n_bins; b2++) {
        const double post = std::min(1.0, exp(-(double) b2)); /* exactly 0.0 in the last bin */
        if (!(post < params->min_posterior_probability_for_partition)) pp.thr_bin = b2;
    }
    e->leaf_part.pool = &ctx->pool;
    e->leaf_np.pool = &ctx->pool;
    e->leaf_count.pool = &ctx->pool;
    e->d_segs.pool = &ctx->pool;
    {   /* what the last engine of this context left behind */
        std::lock_guard<std::mutex> lock(ctx->sibling_mu);
        if (ctx->spare_batch) { e->spare.push_back(ctx->spare_batch); ctx->spare_batch = nullptr; }
        e->spare.insert(e->spare.end(), ctx->spare_batches.begin(), ctx->spare_batches.end());
        ctx->spare_batches.clear();
        e->spare_levels.swap(ctx->spare_levels);
    }
    hipError_t he = e->leaf_part.alloc(4);
    if (he == hipSuccess) he = e->leaf_np.alloc(4);
    if (he == hipSuccess) he = e->leaf_count.alloc(4);
    if (he == hipSuccess) he = e->d_segs.alloc(mrp_engine::MAX_SEGS);
    const uint64_t lp[4] = {1, 0, 0, 0}; /* stRPHmm_construct hmm.c:97-133 */
    const uint32_t ln[4] = {0, 0, 0, 0};
    const int32_t lc[4] = {2, 0, 0, 0};  /* its single column has two cells */
    if (he == hipSuccess) he = hipMemcpy(e->leaf_part.p, lp, sizeof(lp), hipMemcpyHostToDevice);
    if (he == hipSuccess) he = hipMemcpy(e->leaf_np.p, ln, sizeof(ln), hipMemcpyHostToDevice);
    if (he == hipSuccess) he = hipMemcpy(e->leaf_count.p, lc, sizeof(lc), hipMemcpyHostToDevice);
    if (he != hipSuccess) {
        mrp_engine_destroy(e);
        return mrp_set_error(MRP_ERR_HIP, "engine setup failed: %s", hipGetErrorString(he));
    }
    *out = e;
    return MRP_OK;
}

void mrp_engine_destroy(mrp_engine *e) {
    if (!e) return;
    mrp_context *ctx = e->ctx;
    if (getenv("MRP_TIMELINE") && !e->timeline.empty()) { /* one line per engine (= per concurrent batch of a call), times on the process clock */
        char buf[4096]; int o = snprintf(buf, sizeof(buf), "timeline: engine made at %.1f, ends at %.1f; levels (hmms @ launch):", fmod(e->t_created, 1e5), fmod(eng_now(), 1e5));
        for (auto &t : e->timeline) if (o < (int) sizeof(buf) - 40) o += snprintf(buf + o, sizeof(buf) - (size_t) o, " %lld@%.1f", t.first, fmod(t.second, 1e5));
        fprintf(stderr, "%s\n", buf);
    }
    if (getenv("MRP_TIMING")) {
        fprintf(stderr, "  levels: %.1f ms of the stream between the start of the layout kernels and the first kernel of the level proper, %.1f ms of it the host reacting to the totals (allocations, launches)\n", e->diag_layout_gap_ms, e->diag_react_ms);
        fprintf(stderr, "  context waits so far: %.1f ms wall, %.1f ms of thread CPU inside them\n", ctx->wait_wall_ms, ctx->wait_cpu_ms);
        ctx->wait_wall_ms = ctx->wait_cpu_ms = 0;
    }
    (void) hipSetDevice(ctx->device);
    (void) hipStreamSynchronize(ctx->stream);
    if (ctx->pre) (void) hipStreamSynchronize(ctx->pre);
    if (e->staged) { level_retire(e, e->staged); e->staged = nullptr; }
    for (auto *L : e->inflight) level_retire(e, L);
    e->inflight.clear();
    {   /* kept for the next engine of this context */
        std::lock_guard<std::mutex> lock(ctx->sibling_mu);
        for (auto *L : e->spare_levels) ctx->spare_levels.push_back(L);
        for (mrp_batch *b : e->spare) ctx->spare_batches.push_back(b);
    }
    e->spare_levels.clear();
    e->spare.clear();
    delete e;
    ctx->pool.reclaim();
}

}  /* extern "C" */

void mrp_engine_release_context_cache(mrp_context *ctx) {
    for (auto *L : ctx->spare_levels) delete L;
    ctx->spare_levels.clear();
    for (mrp_batch *b : ctx->spare_batches) mrp_batch_destroy(b);
    ctx->spare_batches.clear();
}

extern "C" {

int32_t mrp_engine_stride(const mrp_engine *e) { return e->pp.S; }

int mrp_engine_locate(const mrp_engine *e, int32_t seg, int64_t col0, const uint64_t **part, const uint32_t **np,
                      const int32_t **n_cells, const int32_t **n_merge) {
    if (!e || seg >= e->n_segs) return mrp_set_error(MRP_ERR_ARG, "mrp_engine_locate: no segment %d", seg);
    if (seg < 0) { *part = e->leaf_part.p; *np = e->leaf_np.p; *n_cells = e->leaf_count.p; *n_merge = nullptr; return MRP_OK; }
    const SegDev &sg = e->segtab[seg];
    *part = sg.part + col0 * e->pp.S; *np = sg.np + col0 * e->pp.S; *n_cells = sg.n_cells + col0; *n_merge = sg.n_merge + col0;
    return MRP_OK;
}

void mrp_engine_get_stats(const mrp_engine *e, mrp_engine_stats *out) { *out = e->stats; }

int mrp_engine_fetch(mrp_engine *e, void *dst, const void *src_dev, int64_t bytes) {
    if (bytes <= 0) return MRP_OK;
    ENG_TRY(hipSetDevice(e->ctx->device));
    ENG_TRY(hipMemcpyAsync(dst, src_dev, (size_t) bytes, hipMemcpyDeviceToHost, e->ctx->stream));
    return MRP_OK;
}

int mrp_engine_sync(mrp_engine *e) {
    ENG_TRY(hipSetDevice(e->ctx->device));
    ENG_TRY(e->ctx->wait_stream(e->ctx->stream));
    return MRP_OK;
}

}  /* extern "C" */

/* ---- stage: the static description of a level, built and uploaded while the level before runs ---- */
static int level_stage(mrp_engine *e, int64_t n, mrp_xhmm *x, bool final_level) {
    if (!e || n < 0 || (n > 0 && !x)) return mrp_set_error(MRP_ERR_ARG, "mrp_engine_level: bad arguments");
    if (e->staged) return mrp_set_error(MRP_ERR_ARG, "mrp_engine_level_stage: a staged level was not launched");
    if (n == 0) return MRP_OK;
    mrp_context *ctx = e->ctx;
    ENG_TRY(hipSetDevice(ctx->device));
    hipStream_t cs = nullptr; /* copy stream: nothing here depends on the kernels in flight */
    ENG_TRY(ctx->copy_stream(&cs));
    const int S = e->pp.S;
    if (e->n_segs >= mrp_engine::MAX_SEGS) return mrp_set_error(MRP_ERR_UNSUPPORTED, "more than %d levels", mrp_engine::MAX_SEGS);
    std::unique_ptr<mrp_engine_level_state> L;
    if (!e->spare_levels.empty()) {
        /* the parked level object whose page-locked staging block fits best: the levels of a call differ a hundredfold in size, and a
         * block that has to grow is freed and allocated again (milliseconds each, and both synchronize with the device) */
        size_t want = 4096 + (size_t) n * (sizeof(XDesc) + sizeof(PlanHmm) + sizeof(PruneHmm) + 12 + 4 * 64);
        for (int64_t i = 0; i < n; i++) want += 8 * (size_t) x[i].n_cols + sizeof(mrp_xpar) * (size_t) (x[i].n_a + x[i].n_b);
        size_t best = 0;
        long long best_score = -1;
        for (size_t q = 0; q < e->spare_levels.size(); q++) {
            const mrp_engine_level_state *c = e->spare_levels[q];
            long long score = c->stage.bytes >= want ? (long long) (c->stage.bytes - want) : (1ll << 40) + (long long) (want - c->stage.bytes);
            if (final_level != (c->frag_stage.bytes > 0)) score += 1ll << 36; /* (the final level alone stages the genome fragments' reads) */
            if (best_score < 0 || score < best_score) { best_score = score; best = q; }
        }
        L.reset(e->spare_levels[best]);
        e->spare_levels.erase(e->spare_levels.begin() + (long) best);
    }
    else L.reset(new (std::nothrow) mrp_engine_level_state());
    if (!L) return mrp_set_error(MRP_ERR_NOMEM, "out of host memory");
    L->t_begin = eng_now();
    L->x = x;
    L->n = n;
    L->final_level = final_level;
    L->fused = !final_level && !(ctx->test_hooks & 2);
    for (int64_t i = 0; i < n && L->fused; i++)
        if (x[i].flags & MRP_FLAG_INCLUDE_ANCESTOR_SUB_PROB) L->fused = false;
    const bool fused = L->fused;
    if (!L->done) ENG_TRY(hipEventCreateWithFlags(&L->done, hipEventDisableTiming));
    if (!L->lay0) ENG_TRY(hipEventCreate(&L->lay0));
    L->deferred = false; L->bound_cells = 0; L->bound_merge = 0;
    for (int64_t i = 0; i < n; i++) { L->bound_cells += (x[i].bound_cells + 3) & ~3ll; L->bound_merge += x[i].bound_merge; }
    if (!L->uploaded) {
        ENG_TRY(hipEventCreateWithFlags(&L->uploaded, hipEventDisableTiming));
        for (auto &ev : L->ev) ENG_TRY(hipEventCreate(&ev));
    }

    double tm[8]; int tmi = 0;
    tm[tmi++] = eng_now();
    if (e->spare.empty()) {
        mrp_batch *nb = nullptr;
        int rc = mrp_batch_create(ctx, &nb);
        if (rc != MRP_OK) return rc;
        L->b = nb;
    } else {
        L->b = e->spare.back();
        e->spare.pop_back();
    }
    mrp_batch *b = L->b;
    b->resident = true;
    /* per hmm: where its columns, reads, allele slots and parents start; chunk table; range checks against the static bounds */
    std::vector<int64_t> col0((size_t) n + 1), read0((size_t) n + 1), slot0((size_t) n + 1), par0((size_t) n + 1), cost((size_t) n);
    std::vector<int32_t> chunk_index((size_t) n);
    int64_t total_cols = 0, total_reads = 0, total_slots = 0, total_par = 0;
    bool all_planes = true, no_planes = fused;
    for (int64_t i = 0; i < n; i++) {
        const mrp_xhmm &h = x[i];
        if (h.n_cols < 1 || !h.col_start || !h.col_read_off || !h.chunk || h.n_a < 0 || h.n_b < 0 || (h.n_a + h.n_b > 0 && !h.par) ||
            h.ref_start < 0 || h.ref_end <= h.ref_start || h.ref_end > h.chunk->n_sites)
            return mrp_set_error(MRP_ERR_ARG, "mrp_engine_level: bad hmm %lld", (long long) i);
        const mrp_chunk *ch = h.chunk;
        if (ch->ctx->device != ctx->device) return mrp_set_error(MRP_ERR_ARG, "chunk missing or on a different device");
        int idx = -1;
        if (!b->chunks.empty() && b->chunks.back() == ch) idx = (int) b->chunks.size() - 1;
        for (size_t c = 0; idx < 0 && c < b->chunks.size(); c++)
            if (b->chunks[c] == ch) idx = (int) c;
        if (idx < 0) {
            idx = (int) b->chunks.size(); b->chunks.push_back(ch);
            /* a chunk still being uploaded (work queue): the level's structure kernel and, through L->uploaded, its other
             * kernels are ordered behind the end of the upload */
            if (ch->ready_pending.load()) {
                if (hipEventQuery(ch->ready) == hipSuccess) ch->ready_pending.store(false); /* over: never asked again */
                else ENG_TRY(hipStreamWaitEvent(cs, ch->ready, 0));
            }
        }
        chunk_index[(size_t) i] = idx;
        const bool anc = (h.flags & MRP_FLAG_INCLUDE_ANCESTOR_SUB_PROB) != 0;
        if (!anc) all_planes = false; /* (a column without the ancestor model needs bit planes only if its allele counts differ) */
        int64_t cb = 255ll * h.depth_sites;
        if (anc) cb += (2ll * ch->max_sub + ch->max_prior) * (int64_t) (h.ref_end - h.ref_start);
        cost[(size_t) i] = cb;
        if ((anc && ch->max_alleles > MRP_MAX_ALLELES) || h.bound_max_cells > MRP_PRUNE_MAX_CELLS || h.bound_max_merge > MRP_PRUNE_MAX_CELLS ||
            h.bound_cells >= (1ll << 30) || cb >= (1ll << 30))
            return mrp_set_error(MRP_ERR_UNSUPPORTED, "device-resident hmm %lld is outside the kernels' range", (long long) i);
        col0[(size_t) i] = total_cols; read0[(size_t) i] = total_reads; slot0[(size_t) i] = total_slots; par0[(size_t) i] = total_par;
        total_cols += h.n_cols;
        total_reads += h.n_col_reads; /* (= col_read_off[n_cols] and the allele slots of the interval, as the caller counted them: this loop */
        total_slots += h.n_slots;     /*  runs on the thread that feeds the device and stays inside the descriptions) */
        total_par += h.n_a + h.n_b;
    }
    col0[(size_t) n] = total_cols;
    if (total_cols > 0x7FFFFFFFll) return mrp_set_error(MRP_ERR_UNSUPPORTED, "level with %lld columns", (long long) total_cols);
    L->total_cols = total_cols;
    L->n_reads = total_reads;
    L->n_slots = total_slots;

    tm[tmi++] = eng_now();
    /* the level's output: the pruned hmms, fixed stride (the final level keeps one traced-back cell per column), and the
     * column structure the levels above will look up */
    L->seg.reset(new (std::nothrow) Segment());
    if (!L->seg) return mrp_set_error(MRP_ERR_NOMEM, "out of host memory");
    Segment *seg = L->seg.get();
    DevPool *pl = &ctx->pool;
    seg->part.pool = pl; seg->np.pool = pl; seg->n_cells.pool = seg->n_merge.pool = pl; seg->cols.pool = pl; seg->rbo.pool = pl;
    const int64_t out_stride = final_level ? 1 : S;
    ENG_TRY(seg->part.alloc((size_t) (total_cols * out_stride)));
    ENG_TRY(seg->np.alloc((size_t) (final_level ? 1 : total_cols * S)));
    ENG_TRY(seg->n_cells.alloc((size_t) total_cols));
    ENG_TRY(seg->n_merge.alloc((size_t) total_cols));
    ENG_TRY(seg->cols.alloc((size_t) total_cols));
    ENG_TRY(seg->rbo.alloc((size_t) total_reads));
    const int seg_id = e->n_segs++;
    L->seg_id = seg_id;
    SegDev &sd = e->segtab[seg_id];
    sd.part = seg->part.p; sd.np = seg->np.p; sd.n_cells = seg->n_cells.p; sd.n_merge = seg->n_merge.p; sd.cols = seg->cols.p; sd.rbo = seg->rbo.p;

    /* host staging: one page-locked block holding every array that is uploaded */
    auto al = [](size_t v) { return (v + 63) & ~(size_t) 63; };
    const size_t o_xd = 0, o_par = o_xd + al(sizeof(XDesc) * (size_t) n), o_cstart = o_par + al(sizeof(mrp_xpar) * (size_t) total_par),
                 o_croff = o_cstart + al(4 * ((size_t) total_cols + 1)), o_phmm = o_croff + al(4 * (size_t) total_cols),
                 o_ph = o_phmm + al(sizeof(PlanHmm) * (size_t) n), o_ow = o_ph + al(sizeof(PruneHmm) * (size_t) n), o_om = o_ow + al(4 * (size_t) n),
                 o_on = o_om + al(4 * (size_t) n), o_chunks = o_on + al(4 * (size_t) n), o_seg = o_chunks + al(sizeof(DevChunk) * b->chunks.size()),
                 o_end = o_seg + al(sizeof(SegDev));
    ENG_TRY(L->stage.reserve(o_end));
    char *hb = (char *) L->stage.p;
    XDesc *xd = (XDesc *) (hb + o_xd);
    mrp_xpar *par = (mrp_xpar *) (hb + o_par);
    int32_t *cstart = (int32_t *) (hb + o_cstart), *croff = (int32_t *) (hb + o_croff);
    PlanHmm *phmm = (PlanHmm *) (hb + o_phmm);
    PruneHmm *ph = (PruneHmm *) (hb + o_ph);
    int32_t *ord_w = (int32_t *) (hb + o_ow), *ord_m = (int32_t *) (hb + o_om), *ord_n = (int32_t *) (hb + o_on);
    DevChunk *hchunks = (DevChunk *) (hb + o_chunks);
    for (size_t c = 0; c < b->chunks.size(); c++) hchunks[c] = b->chunks[c]->dev;
    *(SegDev *) (hb + o_seg) = sd;

    tm[tmi++] = eng_now();
    /* one array entry per complement pair (MRP_XF_UNITS) where the one-pass cross product + emission kernel writes the level and
     * the prune reads it by units; MRP_UNITS=0 (development) keeps one entry per cell.  (test hook bit 3: the general prune chain) */
    L->units = L->fused && e->pp.pairs != 0 && !(ctx->test_hooks & 8) && !(getenv("MRP_UNITS") && getenv("MRP_UNITS")[0] == '0');
    /* hmms whose columns hold at most 64 units and 64 merge units (the static bounds count cells) go through recursion, prune
     * and compaction on ONE wave each (mrp_mini_kernel): the first merge levels, tens of thousands of hmms of a few cells.
     * They sit at the end of the level's PruneHmm array.  MRP_MINI=0 (development) switches the class off. */
    const bool mini_on = L->units && !(getenv("MRP_MINI") && getenv("MRP_MINI")[0] == '0');
    auto is_mini = [&](int64_t i) { return mini_on && x[i].bound_max_cells <= 2 * MRP_MINI_MAX_UNITS && x[i].bound_max_merge <= 2 * MRP_MINI_MAX_UNITS; };
    /* the prune kernel walks one hmm per workgroup, its columns one after the other: longest hmms first */
    L->perm.resize((size_t) n);
    std::vector<int32_t> pos((size_t) n);
    L->n_mini = 0;
    {   /* stable counting sort by descending number of columns, the single-wave class behind the others */
        int32_t max_cols = 1;
        for (int64_t i = 0; i < n; i++) max_cols = std::max(max_cols, x[i].n_cols);
        const size_t half = (size_t) max_cols + 1;
        std::vector<int64_t> at(2 * half + 1, 0);
        auto slot_of = [&](int64_t i) { return (size_t) (max_cols - x[i].n_cols) + (is_mini(i) ? half : 0); };
        for (int64_t i = 0; i < n; i++) { at[slot_of(i) + 1]++; if (is_mini(i)) L->n_mini++; }
        for (size_t q = 1; q < at.size(); q++) at[q] += at[q - 1];
        for (int64_t i = 0; i < n; i++) L->perm[(size_t) at[slot_of(i)]++] = (int32_t) i;
        for (int64_t j = 0; j < n; j++) pos[(size_t) L->perm[(size_t) j]] = (int32_t) j;
    }
    /* per hmm records and the 8 bytes per column the host contributes (parallel) */
    mrp_pool_set_tag(8); mrp_pool_set_weight(400); mrp_parallel_for(n, std::max<int64_t>(1, n / 256), [&](int64_t i) {
        mrp_xhmm &h = x[i];
        if (i + 3 < n) { __builtin_prefetch(x[i + 3].col_start); __builtin_prefetch(x[i + 3].col_read_off); __builtin_prefetch(x[i + 3].par); }
        const int K = h.n_cols;
        const int64_t colbase = col0[(size_t) i];
        XDesc &d = xd[i];
        d.col0 = colbase; d.read0 = read0[(size_t) i]; d.slot0 = slot0[(size_t) i]; d.par0 = par0[(size_t) i];
        d.ref_start = h.ref_start; d.ref_end = h.ref_end; d.n_cols = K; d.n_a = h.n_a; d.n_b = h.n_b;
        d.chunk = chunk_index[(size_t) i]; d.flags = h.flags; d.prune_pos = pos[(size_t) i];
        if (h.n_a + h.n_b > 0) memcpy(par + par0[(size_t) i], h.par, sizeof(mrp_xpar) * (size_t) (h.n_a + h.n_b));
        memcpy(cstart + colbase, h.col_start, sizeof(int32_t) * (size_t) K);
        memcpy(croff + colbase, h.col_read_off, sizeof(int32_t) * (size_t) K);
        PlanHmm &p = phmm[i];
        p.col0 = colbase; p.n_cols = K; p.flags = h.flags; p.cost_bound = cost[(size_t) i];
        PruneHmm &q = ph[pos[(size_t) i]];
        q.col0 = colbase; q.n_cols = K; q.hmm_index = (int32_t) i;
        q.out_part = seg->part.p + colbase * out_stride;
        q.out_np = final_level ? seg->np.p : seg->np.p + colbase * S;
        q.out_n_cells = seg->n_cells.p + colbase;
        q.out_n_merge = seg->n_merge.p + colbase;
        h.seg = seg_id; h.col0 = colbase;
        h.err = 0;
    });
    mrp_pool_set_weight(0);
    cstart[total_cols] = 0;
    tm[tmi++] = eng_now();
    /* launch classes of the recursion kernel, from the static bounds; largest first inside a class */
    b->order_wide.clear(); b->order_mid.clear(); b->order_narrow.clear(); b->order_f64.clear(); b->order_lse.clear(); b->order_lse_big.clear(); b->order_gen.clear();
    b->max_merge_wide = b->max_merge_mid = b->max_merge_narrow = 1;
    {
        std::vector<std::pair<int64_t, int32_t>> wide, mid, narrow;
        for (int64_t i = 0; i < n; i++) {
            const mrp_xhmm &q = x[i];
            if (is_mini(i)) continue; /* swept by the single-wave kernel */
            if (q.bound_max_cells <= 256) narrow.push_back({-q.bound_cells, (int32_t) i});
            else if (q.bound_max_merge <= 4096) mid.push_back({-q.bound_cells, (int32_t) i});
            else wide.push_back({-q.bound_cells, (int32_t) i});
        }
        auto plan_class = [&](std::vector<std::pair<int64_t, int32_t>> &v, std::vector<int32_t> &order, int32_t *dst, int *mm) {
            /* largest first, so that the long chains start early; a class of many thousand hmms (the first merge levels: a few
             * cells each, a level of 25 000) has no tail worth 1.5 ms of sorting on the thread that feeds the device */
            if (v.size() <= 4096) std::sort(v.begin(), v.end());
            int m = 1;
            for (size_t j = 0; j < v.size(); j++) {
                order.push_back(v[j].second);
                dst[j] = v[j].second;
                m = std::max(m, std::max(1, x[v[j].second].bound_max_merge));
            }
            *mm = m;
        };
        plan_class(wide, b->order_wide, ord_w, &b->max_merge_wide);
        plan_class(mid, b->order_mid, ord_m, &b->max_merge_mid);
        plan_class(narrow, b->order_narrow, ord_n, &b->max_merge_narrow);
        if (L->units) { /* the merge columns of a unit level hold one entry per pair (the bounds count cells): half the LDS per workgroup */
            b->max_merge_wide = (b->max_merge_wide + 1) / 2 + 1; b->max_merge_mid = (b->max_merge_mid + 1) / 2 + 1; b->max_merge_narrow = (b->max_merge_narrow + 1) / 2 + 1;
        }
    }
    PruneParams &pp = L->pp;
    pp = e->pp;
    pp.max_cells = 1; pp.max_merge = 1;
    for (int64_t i = 0; i < n; i++) {
        pp.max_cells = std::max(pp.max_cells, x[i].bound_max_cells);
        pp.max_merge = std::max(pp.max_merge, x[i].bound_max_merge);
    }
    if (ctx->test_hooks & 8) pp.pairs = 0; /* test hook: the general prune chain, for A/B parity with the chain on complement pairs */
    if (L->units) pp.pairs = 2; /* (decided above: the level's arrays hold one entry per complement pair) */
    pp.pad = (ctx->test_hooks & 1) && e->stats.levels + (int64_t) e->inflight.size() == 1 ? 1 : 0; /* test hook, see mrp_context_set_test_hooks */

    tm[tmi++] = eng_now();
    /* device side of the description + the descriptor arrays the structure and layout kernels fill */
    b->bind_pool(pl);
    L->d_plan.pool = pl; L->d_xd.pool = pl; L->d_par.pool = pl; L->d_cstart.pool = L->d_croff.pool = pl;
    L->d_phmm.pool = pl; L->d_dims.pool = pl; L->d_tot.pool = pl; L->d_base.pool = pl; L->d_totals.pool = pl; L->d_tile_sums.pool = pl;
    L->d_cc.pool = pl; L->d_ph.pool = pl; L->d_col_hmm.pool = L->d_nkept.pool = L->d_nkeptm.pool = L->d_err.pool = L->d_err_hmm.pool = pl;
    L->d_kept.pool = L->d_keptm.pool = pl; L->d_kept_np.pool = pl;
    ENG_TRY(L->d_plan.alloc((size_t) total_cols)); ENG_TRY(L->d_xd.alloc((size_t) n)); ENG_TRY(L->d_par.alloc((size_t) total_par));
    ENG_TRY(L->d_cstart.alloc((size_t) total_cols + 1)); ENG_TRY(L->d_croff.alloc((size_t) total_cols));
    ENG_TRY(L->d_phmm.alloc((size_t) n)); ENG_TRY(L->d_dims.alloc(4 * (size_t) total_cols));
    ENG_TRY(L->d_tot.alloc((size_t) n)); ENG_TRY(L->d_base.alloc((size_t) n)); ENG_TRY(L->d_totals.alloc(8));
    ENG_TRY(L->d_tile_sums.alloc(6 * (((size_t) n + 255) / 256)));
    ENG_TRY(L->d_cc.alloc((size_t) total_cols)); ENG_TRY(L->d_ph.alloc((size_t) n)); ENG_TRY(L->d_col_hmm.alloc((size_t) total_cols));
    ENG_TRY(L->d_err.alloc(64)); ENG_TRY(L->d_err_hmm.alloc((size_t) n));
    ENG_TRY(b->d_hmms.alloc((size_t) n)); ENG_TRY(b->d_cols.alloc((size_t) total_cols)); ENG_TRY(b->d_scols.alloc((size_t) total_cols));
    ENG_TRY(b->d_pcols.alloc((size_t) total_cols)); ENG_TRY(b->d_tilecols.alloc((size_t) total_cols));
    ENG_TRY(b->d_chunks.alloc(b->chunks.size()));
    ENG_TRY(b->d_order_wide.alloc(b->order_wide.size())); ENG_TRY(b->d_order_mid.alloc(b->order_mid.size()));
    ENG_TRY(b->d_order_narrow.alloc(b->order_narrow.size())); ENG_TRY(b->d_order_f64.alloc(1));
    auto up = [&](void *dst, const void *src, size_t bytes) -> hipError_t {
        return bytes ? hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, cs) : hipSuccess;
    };
    ENG_TRY(up(L->d_xd.p, xd, sizeof(XDesc) * (size_t) n));
    ENG_TRY(up(L->d_par.p, par, sizeof(mrp_xpar) * (size_t) total_par));
    ENG_TRY(up(L->d_cstart.p, cstart, 4 * ((size_t) total_cols + 1)));
    ENG_TRY(up(L->d_croff.p, croff, 4 * (size_t) total_cols));
    ENG_TRY(up(L->d_phmm.p, phmm, sizeof(PlanHmm) * (size_t) n));
    ENG_TRY(up(L->d_ph.p, ph, sizeof(PruneHmm) * (size_t) n));
    ENG_TRY(up(b->d_order_wide.p, ord_w, 4 * b->order_wide.size()));
    ENG_TRY(up(b->d_order_mid.p, ord_m, 4 * b->order_mid.size()));
    ENG_TRY(up(b->d_order_narrow.p, ord_n, 4 * b->order_narrow.size()));
    ENG_TRY(up(b->d_chunks.p, hchunks, sizeof(DevChunk) * b->chunks.size()));
    ENG_TRY(up(e->d_segs.p + seg_id, hb + o_seg, sizeof(SegDev)));
    ENG_TRY(hipMemsetAsync(L->d_err.p, 0, 256, cs));
    ENG_TRY(hipMemsetAsync(L->d_err_hmm.p, 0, sizeof(int32_t) * (size_t) n, cs));
    {   /* the columns of the level, one thread each: parents, connectors, reads, allele slots (also on the copy stream: the
         * tables it reads were written by the structure kernels of the levels below, on the same stream) */
        StructureIn si{};
        si.xd = L->d_xd.p; si.n_hmms = n; si.n_cols = total_cols; si.par = L->d_par.p; si.col_start = L->d_cstart.p; si.col_roff = L->d_croff.p;
        si.segs = e->d_segs.p; si.chunks = b->d_chunks.p;
        si.leaf_part = e->leaf_part.p; si.leaf_np = e->leaf_np.p; si.leaf_count = e->leaf_count.p;
        si.stride = S; si.fused = fused ? 1 : 0;
        si.plan = L->d_plan.p; si.cols = seg->cols.p; si.rbo = seg->rbo.p; si.col_hmm = L->d_col_hmm.p;
        si.err = L->d_err.p; si.err_hmm = L->d_err_hmm.p;
        ENG_TRY(mrp_launch_structure(si, cs));
        if (mrp_dup('t')) ENG_TRY(mrp_launch_structure(si, cs));
    }
    L->frag = false;
    if (final_level && n > 0) { /* genome fragments on the device: every hmm of the stage brings its chunk's reads */
        bool all = !(getenv("MRP_FRAGMENTS") && getenv("MRP_FRAGMENTS")[0] == '0');
        for (int64_t i = 0; i < n && all; i++) all = x[i].frag_reads && x[i].frag_by_pool && x[i].frag_sites && x[i].frag_reads1 && x[i].frag_reads2 && x[i].frag_n_reads > 0;
        if (all) {
            L->frag_hmms.resize((size_t) n);
            int64_t r0 = 0, d0 = 0, s0 = 0, l0 = 0;
            for (int64_t i = 0; i < n; i++) { /* (in the order of the level's PruneHmm array: the kernel indexes both alike) */
                const mrp_xhmm &h = x[(size_t) L->perm[(size_t) i]];
                FragHmm &f = L->frag_hmms[(size_t) i];
                f.reads0 = r0; f.disc0 = d0; f.site0 = s0; f.list0 = l0; f.slot0 = read0[(size_t) L->perm[(size_t) i]];
                f.n_reads = h.frag_n_reads; f.n_discarded = h.frag_n_discarded; f.ref_start = h.ref_start; f.length = h.ref_end - h.ref_start;
                f.max_iterations = h.frag_iterations; f.pad = 0;
                r0 += h.frag_n_reads; d0 += h.frag_n_discarded; s0 += f.length; l0 += 2 * (int64_t) h.frag_n_reads + 2;
            }
            L->frag_reads_total = r0; L->frag_disc_total = d0; L->frag_sites_total = s0; L->frag_list_total = l0;
            auto al64 = [](size_t v) { return (v + 63) & ~(size_t) 63; };
            const size_t o_fh = 0, o_fr = o_fh + al64(sizeof(FragHmm) * (size_t) n), o_fp = o_fr + al64(sizeof(FragRead) * (size_t) r0), o_fd = o_fp + al64(4 * (size_t) r0),
                         o_fe = o_fd + al64(4 * (size_t) d0 + 4);
            ENG_TRY(L->frag_stage.reserve(o_fe));
            char *fb_ = (char *) L->frag_stage.p;
            memcpy(fb_ + o_fh, L->frag_hmms.data(), sizeof(FragHmm) * (size_t) n);
            FragRead *fr = (FragRead *) (fb_ + o_fr);
            int32_t *fp = (int32_t *) (fb_ + o_fp), *fd = (int32_t *) (fb_ + o_fd);
            mrp_parallel_for(n, 1, [&](int64_t i) {
                const mrp_xhmm &h = x[(size_t) L->perm[(size_t) i]];
                const FragHmm &f = L->frag_hmms[(size_t) i];
                for (int32_t r = 0; r < h.frag_n_reads; r++) {
                    FragRead &q = fr[f.reads0 + r];
                    q.ref_start = h.frag_reads[r].ref_start; q.length = h.frag_reads[r].length; q.pool_offset = h.frag_reads[r].pool_offset;
                }
                memcpy(fp + f.reads0, h.frag_by_pool, 4 * (size_t) h.frag_n_reads);
                if (h.frag_n_discarded > 0) memcpy(fd + f.disc0, h.frag_discarded, 4 * (size_t) h.frag_n_discarded);
            });
            L->d_frag_hmms.pool = pl; L->d_frag_reads.pool = pl; L->d_frag_by_pool.pool = pl; L->d_frag_disc.pool = pl; L->d_frag_lists.pool = pl; L->d_frag_work.pool = pl;
            L->d_frag_counts.pool = pl; L->d_frag_col_read.pool = pl; L->d_frag_col_cnt.pool = pl; L->d_frag_sites.pool = pl; L->d_frag_col_part.pool = pl; L->d_frag_read_key.pool = pl;
            ENG_TRY(L->d_frag_hmms.alloc((size_t) n)); ENG_TRY(L->d_frag_reads.alloc((size_t) r0)); ENG_TRY(L->d_frag_by_pool.alloc((size_t) r0));
            ENG_TRY(L->d_frag_disc.alloc((size_t) d0 + 1)); ENG_TRY(L->d_frag_lists.alloc(2 * (size_t) l0)); ENG_TRY(L->d_frag_work.alloc(2 * (size_t) l0));
            ENG_TRY(L->d_frag_counts.alloc(2 * (size_t) n)); ENG_TRY(L->d_frag_col_read.alloc((size_t) total_reads + 1)); ENG_TRY(L->d_frag_col_cnt.alloc(2 * (size_t) total_cols));
            ENG_TRY(L->d_frag_sites.alloc((size_t) s0)); ENG_TRY(L->d_frag_col_part.alloc((size_t) total_cols)); ENG_TRY(L->d_frag_read_key.alloc(2 * (size_t) r0));
            ENG_TRY(up(L->d_frag_hmms.p, fb_ + o_fh, sizeof(FragHmm) * (size_t) n));
            ENG_TRY(up(L->d_frag_reads.p, fr, sizeof(FragRead) * (size_t) r0));
            ENG_TRY(up(L->d_frag_by_pool.p, fp, 4 * (size_t) r0));
            ENG_TRY(up(L->d_frag_disc.p, fd, 4 * (size_t) d0));
            ENG_TRY(L->frag_results.reserve(al64(sizeof(FragSite) * (size_t) s0) + al64(8 * (size_t) l0) + al64(8 * (size_t) n)));
            L->h_frag_sites = (FragSite *) L->frag_results.p;
            L->h_frag_lists = (int32_t *) ((char *) L->frag_results.p + al64(sizeof(FragSite) * (size_t) s0));
            L->h_frag_counts = (int32_t *) ((char *) L->h_frag_lists + al64(8 * (size_t) l0));
            L->frag = true;
        }
    }
    ENG_TRY(hipEventRecord(L->uploaded, cs));
    /* which of the two packing kernels has columns to look at (they filter by PlaneCol.need_planes) */
    L->any_pack = !all_planes;
    L->any_planes = !no_planes;
    /* results come back into a second page-locked block */
    {
        const size_t cols8 = ((size_t) total_cols + 1) & ~(size_t) 1, n8 = ((size_t) n + 1) & ~(size_t) 1;
        ENG_TRY(L->results.reserve(128 + n8 * 4 + (final_level ? cols8 * 8 + cols8 * 4 + (size_t) n * 16 : 0)));
        char *rb = (char *) L->results.p;
        L->totals = (int64_t *) rb;
        L->err = (int32_t *) (rb + 64);
        L->err_hmm = (int32_t *) (rb + 128);
        L->path_part = (uint64_t *) (rb + 128 + n8 * 4);
        L->fb = (double *) (rb + 128 + n8 * 4 + cols8 * 8);
        L->path_cell = (int32_t *) (rb + 128 + n8 * 4 + cols8 * 8 + (size_t) n * 16);
    }
    b->stats.n_hmms = n;
    b->stats.n_columns = total_cols;
    L->t_staged = eng_now();
    if (getenv("MRP_TIMING"))
        fprintf(stderr, "      stage: offsets+chunks %.2f ms, segment+staging %.2f, records %.2f, classes %.2f, allocs+uploads %.2f\n", tm[1] - tm[0],
                tm[2] - tm[1], tm[3] - tm[2], tm[4] - tm[3], L->t_staged - tm[4]);
    e->staged = L.release();
    return MRP_OK;
}

/* ---- end: the per-hmm error flags of the running level (and the final level's results) ---- */
/* one level whose `done` event is complete (or whose stream has been waited for) */
static int level_finish_one(mrp_engine *e, mrp_engine_level_state *Lp, bool complete) {
    int rc = MRP_OK;
    if (rc == MRP_OK && getenv("MRP_TIMING")) {
        fprintf(stderr, "  level: %lld hmms %lld cols %lld cells: staged in %.1f ms, launch (layout + totals + queue) %.1f ms  [stage began at %.1f ms, launched at %.1f ms on the process clock]\n", (long long) Lp->n,
                (long long) Lp->total_cols, (long long) Lp->totals[0], Lp->t_staged - Lp->t_begin, Lp->t_launch_ms, fmod(Lp->t_begin, 1e5), fmod(Lp->t_launched, 1e5));
#if defined(PRUNE_EXP_CLOCK) || defined(PRUNE_EXP_CLOCK2) || defined(XE_CLOCK)
        fprintf(stderr, "  prune clocks (first hmm; shader cycles):");
        for (int i = 0; i < 12; i++) fprintf(stderr, " %llu", Lp->clk[i]);
        fprintf(stderr, "\n");
#endif
    }
    /* Per hmm: a parent outside the closed-form cross product's pair order, or a merge cell the kept cells lead to that
     * hmm.c:1090-1100 would drop, means "not handled on the device": the caller redoes that hmm's chunk on the hashing path
     * (whatever else the kernels flagged for it came from the discarded arrays).  Posterior / range violations on an hmm
     * that is otherwise fine are the reference's st_errAbort cases. */
    if (rc == MRP_OK && Lp->err[0] != 0) {
        for (int64_t j = 0; j < Lp->n && rc == MRP_OK; j++) {
            const int32_t bits = Lp->err_hmm[j];
            if (bits == 0) continue;
            Lp->x[(size_t) Lp->perm[(size_t) j]].err = bits;
            if (bits & (MRP_ENGINE_ERR_STRUCTURE | MRP_ENGINE_ERR_MERGE)) {
                /* the chunk leaves the resident path: said at once (the caller says the same when it settles the level), so that the
                 * levels in flight behind this one -- they ran on this hmm's discarded arrays -- are not held to what they raise */
                const mrp_xhmm &xq = Lp->x[(size_t) Lp->perm[(size_t) j]];
                if (xq.discarded) *const_cast<int *>(xq.discarded) = 1;
                continue;
            }
            /* an hmm whose chunk already left the resident path at the level before (this level was staged before that was
             * known): it ran on a discarded parent's arrays, whatever it raised is that parent's */
            { const mrp_xhmm &xq = Lp->x[(size_t) Lp->perm[(size_t) j]]; if (xq.discarded && *xq.discarded) continue; }
            if (bits & MRP_ENGINE_ERR_POSTERIOR) rc = mrp_set_error(MRP_ERR_ARG, "ERROR: invalid prob (f + b exceeds the column total)");
            else rc = mrp_set_error(MRP_ERR_LOOKUP, "device-resident merge: transition index out of range");
        }
    }
    if (rc == MRP_OK && Lp->final_level) {
        int64_t colbase = 0;
        for (int64_t i = 0; i < Lp->n; i++) {
            if (Lp->x[i].n_cells) memcpy(Lp->x[i].n_cells, Lp->path_cell + colbase, sizeof(int32_t) * (size_t) Lp->x[i].n_cols);
            if (Lp->x[i].path_part) memcpy(Lp->x[i].path_part, Lp->path_part + colbase, sizeof(uint64_t) * (size_t) Lp->x[i].n_cols);
            Lp->x[i].hmm_forward = Lp->fb[(size_t) (2 * i)];
            Lp->x[i].hmm_backward = Lp->fb[(size_t) (2 * i + 1)];
            colbase += Lp->x[i].n_cols;
        }
        if (Lp->frag) { /* the genome fragments, in the order of the PruneHmm array */
            mrp_parallel_for(Lp->n, 1, [&](int64_t j) {
                mrp_xhmm &h = Lp->x[(size_t) Lp->perm[(size_t) j]];
                const FragHmm &f = Lp->frag_hmms[(size_t) j];
                h.frag_done = 0;
                if (Lp->err_hmm[j] != 0) return;
                const int cap = 2 * f.n_reads + 2;
                const int n1 = Lp->h_frag_counts[2 * j], n2 = Lp->h_frag_counts[2 * j + 1];
                if (n1 < 0 || n2 < 0 || n1 > cap || n2 > cap) return;
                memcpy(h.frag_sites, Lp->h_frag_sites + f.site0, sizeof(FragSite) * (size_t) f.length);
                memcpy(h.frag_reads1, Lp->h_frag_lists + 2 * f.list0, 4 * (size_t) n1);
                memcpy(h.frag_reads2, Lp->h_frag_lists + 2 * f.list0 + cap, 4 * (size_t) n2);
                h.frag_n1 = n1; h.frag_n2 = n2; h.frag_done = 1;
            });
        }
    }
    if (rc == MRP_OK) {
        float t_cross = 0, t_sweep = 0, t_prune = 0;
        (void) hipEventElapsedTime(&t_cross, Lp->ev[0], Lp->ev[1]);
        (void) hipEventElapsedTime(&t_sweep, Lp->ev[1], Lp->ev[2]);
        (void) hipEventElapsedTime(&t_prune, Lp->ev[2], Lp->ev[3]);
        e->stats.levels += 1;
        e->stats.hmms += Lp->n;
        e->stats.columns += Lp->total_cols;
        e->stats.cells += Lp->totals[4];       /* the cross products' cells / merge cells (the arrays may hold units, mrp_engine.h) */
        e->stats.merge_cells += Lp->totals[5];
        e->stats.cross_ms += t_cross;
        e->stats.sweep_ms += t_sweep;
        e->stats.prune_ms += t_prune;
        e->stats.device_ms += t_cross + t_sweep + t_prune;
        {   /* (diagnostics, MRP_TIMING) the stream between the layout kernels' start and the level's first own kernel: layout kernels,
             * totals back, the host's reaction (allocations, launches) */
            float t_lay = 0;
            (void) hipEventElapsedTime(&t_lay, Lp->lay0, Lp->ev[0]);
            e->diag_layout_gap_ms += t_lay; e->diag_react_ms += Lp->t_react_ms > 0 ? Lp->t_react_ms : 0;
        }
        {   /* by kernel family: packing, cross product + emission, recursion (the batch's own events), prune, compaction */
            float t_pack = 0, t_emit = 0, t_rec = 0, t_pr = 0;
            mrp_batch_last_launch_ms(Lp->b, &t_pack, &t_emit, &t_rec);
            if (!Lp->final_level) (void) hipEventElapsedTime(&t_pr, Lp->ev[2], Lp->ev[4]);
            e->stats.pack_ms += t_pack;
            e->stats.cross_emit_ms += t_cross + t_emit;
            e->stats.recursion_ms += t_rec;
            e->stats.prune_kernel_ms += Lp->final_level ? 0.0 : t_pr;
            e->stats.compact_ms += Lp->final_level ? t_prune : t_prune - t_pr; /* (final level: the trace back) */
        }
        e->segments.push_back(std::move(Lp->seg));
    }
    e->n_ended++;
    level_retire(e, Lp, complete);
    return rc;
}

/* ends every level in flight, oldest first, after ONE wait for the stream they were queued on */
static int level_finish(mrp_engine *e) {
    if (e->inflight.empty()) return MRP_OK;
    mrp_context *ctx = e->ctx;
    int rc = MRP_OK;
    hipError_t se = hipSetDevice(ctx->device);
    if (se == hipSuccess) se = ctx->wait_stream(ctx->stream);
    if (se != hipSuccess) rc = mrp_set_error(MRP_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(se));
    std::vector<mrp_engine_level_state *> all;
    all.swap(e->inflight);
    for (auto *Lp : all) {
        if (rc == MRP_OK) rc = level_finish_one(e, Lp, false);
        else { e->n_ended++; level_retire(e, Lp); }
    }
    return rc;
}

/* ends the levels in flight whose work is over (oldest first, as far as their events say so); never waits */
static int level_finish_ready(mrp_engine *e) {
    int rc = MRP_OK;
    while (rc == MRP_OK && !e->inflight.empty()) {
        mrp_engine_level_state *Lp = e->inflight.front();
        if (!Lp->done || hipEventQuery(Lp->done) != hipSuccess) { (void) hipGetLastError(); break; }
        e->inflight.erase(e->inflight.begin());
        rc = level_finish_one(e, Lp, true);
    }
    return rc;
}


/* ---- launch: layout on the device, four totals back, allocation, the level's kernels ---- */
static int level_launch_impl(mrp_engine *e, mrp_engine_level_state *L) {
    mrp_context *ctx = e->ctx;
    ENG_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    mrp_batch *b = L->b;
    const int64_t n = L->n, total_cols = L->total_cols;
    const double t0 = eng_now();
    LayoutOut lo{};
    lo.dims = L->d_dims.p; lo.tot = L->d_tot.p; lo.base = L->d_base.p; lo.totals = L->d_totals.p; lo.tile_sums = L->d_tile_sums.p;
    lo.hmms = b->d_hmms.p; lo.cols = b->d_cols.p; lo.scols = b->d_scols.p; lo.pcols = b->d_pcols.p; lo.tilecols = b->d_tilecols.p;
    lo.ccols = L->d_cc.p;
    ENG_TRY(hipStreamWaitEvent(s, L->uploaded, 0));
    ENG_TRY(hipEventRecord(L->lay0, s));
    ENG_TRY(mrp_launch_layout(L->d_plan.p, L->d_phmm.p, n, total_cols, b->d_chunks.p, e->pp.S,
                              (e->params.include_inverted_partitions ? MRP_XF_INVERTED : 0u) | (L->units ? MRP_XF_UNITS : 0u), lo, s));
    if (mrp_dup('l'))
        ENG_TRY(mrp_launch_layout(L->d_plan.p, L->d_phmm.p, n, total_cols, b->d_chunks.p, e->pp.S,
                                  (e->params.include_inverted_partitions ? MRP_XF_INVERTED : 0u) | (L->units ? MRP_XF_UNITS : 0u), lo, s));
    ENG_TRY(hipMemcpyAsync(L->totals, L->d_totals.p, 48, hipMemcpyDeviceToHost, s));
    /* Deferred launch (round 5): a small merge level -- its arrays at most MRP_DEFER_MB (1 GB) by the hmms' STATIC bounds, 4 GB of
     * such levels in flight -- does not wait for its totals: the arrays are sized by the bounds (the layout kernels' offsets stay
     * inside them: every count is clamped to what the bounds assume), its kernels are queued behind the layout kernels at once and the
     * level before is ended whenever its event has completed.  A call of one chunk walks eleven levels whose kernels take a
     * millisecond or two each: the wait (totals back, the host awake again, two dozen allocations, a dozen launches) left the device
     * idle for a third of a millisecond per level.  The large levels keep the wait: bounds are loose there (a pruned column of 100
     * cells times another is the bound, a fifth of it the average) and their arrays are what the device's memory goes to. */
    static const long defer_mb = getenv("MRP_DEFER_MB") ? atol(getenv("MRP_DEFER_MB")) : 1024;
    const int64_t bound_bytes = 16 * L->bound_cells + 8 * L->bound_merge;
    const bool defer = L->fused && !L->final_level && defer_mb > 0 && bound_bytes <= ((int64_t) defer_mb << 20) &&
                       e->inflight_bytes + bound_bytes <= ((int64_t) 4 * defer_mb << 20) && e->inflight.size() < 12;
    int rc;
    int64_t cells, merge, tiles_fast = 0, tiles = 0;
    if (defer) {
        rc = level_finish_ready(e);
        if (rc != MRP_OK) return rc;
        cells = L->bound_cells; merge = L->bound_merge;
        L->deferred = true;
        e->inflight_bytes += bound_bytes;
    } else {
        /* the one host wait of a level: it also ends the levels before (their error flags are in) */
        rc = level_finish(e);
        if (rc != MRP_OK) return rc;
        ENG_TRY(ctx->wait_stream(s));
        L->t_react_ms = -eng_now();
        ctx->pool.reclaim(); /* the blocks of the level before can be reused */
        cells = L->totals[0]; merge = L->totals[1]; tiles_fast = L->totals[2]; tiles = L->totals[2] + L->totals[3];
    }
    b->n_cells_total = cells; b->n_merge = merge; b->n_slots = L->n_slots; b->n_tiles_dev = tiles; b->n_fast_tiles = tiles_fast;
    b->stats.n_cells = cells; b->stats.n_merge_cells = merge;
    b->stats.algorithmic_bytes = 24 * cells + 32 * merge + 8 * total_cols;
    const size_t nC = (size_t) cells;
    if (L->fused) { b->d_partition.release(); b->n_tiles_dev = 0; b->n_fast_tiles = 0; }
    else ENG_TRY(b->d_partition.alloc(nC));
    ENG_TRY(b->d_np.alloc(nC)); ENG_TRY(b->d_cost.alloc(nC));
    ENG_TRY(b->d_f32.alloc(nC)); ENG_TRY(b->d_b32.alloc(nC));
    ENG_TRY(b->d_mf32.alloc((size_t) merge)); ENG_TRY(b->d_mb32.alloc((size_t) merge));
    if (!L->fused) ENG_TRY(b->d_tiles.alloc((size_t) tiles));
    ENG_TRY(b->d_planes.alloc((size_t) L->n_slots * 8)); ENG_TRY(b->d_slot_total.alloc((size_t) L->n_slots));
    ENG_TRY(b->d_slot_bytes.alloc((size_t) L->n_slots * 16));
    ENG_TRY(b->d_total.alloc((size_t) total_cols)); ENG_TRY(b->d_hmm_fb.alloc(2 * (size_t) n));
    MrpBatchDev &d = b->dev;
    d = MrpBatchDev{};
    d.hmms = b->d_hmms.p; d.cols = b->d_cols.p; d.chunks = b->d_chunks.p; d.read_byte_off = L->seg->rbo.p;
    d.partition = b->d_partition.p; d.scols = b->d_scols.p; d.pcols = b->d_pcols.p;
    d.pack_list = nullptr; d.plane_list = nullptr; d.list_filter = 1; /* every column, filtered by PlaneCol.need_planes */
    d.n_pack_list = L->any_pack ? total_cols : 0; d.n_plane_list = L->any_planes ? total_cols : 0;
    d.cell_np = b->d_np.p; d.planes = b->d_planes.p; d.slot_total = b->d_slot_total.p; d.slot_bytes = b->d_slot_bytes.p;
    d.cell_cost = b->d_cost.p; d.cell_f32 = b->d_f32.p; d.cell_b32 = b->d_b32.p; d.merge_f32 = b->d_mf32.p; d.merge_b32 = b->d_mb32.p;
    d.col_total = b->d_total.p; d.hmm_fb = b->d_hmm_fb.p;
    d.n_hmms = n; d.n_cols = total_cols; d.n_cells = cells; d.n_merge = merge; d.n_slots = L->n_slots;
    b->uploaded = true;
    b->outs.clear(); /* device-only */

    const int S = e->pp.S;
    if (!L->final_level) {
        ENG_TRY(L->d_kept.alloc((size_t) total_cols * S));
        ENG_TRY(L->d_keptm.alloc((size_t) total_cols * S));
        ENG_TRY(L->d_kept_np.alloc((size_t) total_cols * S));
        ENG_TRY(L->d_nkept.alloc((size_t) total_cols));
        ENG_TRY(L->d_nkeptm.alloc((size_t) total_cols));
    }
    PruneScratch sc{};
    sc.kept = L->d_kept.p; sc.kept_np = L->d_kept_np.p; sc.keptm = L->d_keptm.p; sc.n_kept = L->d_nkept.p; sc.n_keptm = L->d_nkeptm.p;
    sc.err = L->d_err.p;
    sc.err_hmm = L->d_err_hmm.p;

    Segment *seg = L->seg.get();
    if (L->fused) {
        /* the packed profile bytes first (mrp_batch_launch), then cross product + emission in one pass; no tiles, no
         * partitions, the batch's own emission launch finds nothing to do */
        ENG_TRY(hipEventRecord(L->ev[0], s));
        ENG_TRY(hipEventRecord(L->ev[1], s));
        b->pre_sweep = [L](hipStream_t st) -> hipError_t {
            return mrp_launch_cross_emit(L->d_cc.p, L->b->dev, L->d_err.p, L->d_col_hmm.p, L->d_err_hmm.p, L->pp.max_cells, 2 * L->n_mini > L->n, st);
        };
    } else {
        b->pre_sweep = nullptr;
        ENG_TRY(mrp_launch_tiles(b->d_cols.p, b->d_tilecols.p, total_cols, b->d_tiles.p, s));
        ENG_TRY(hipEventRecord(L->ev[0], s));
        ENG_TRY(mrp_launch_cross(L->d_cc.p, total_cols, b->d_partition.p, b->d_np.p, L->d_err.p, L->d_col_hmm.p, L->d_err_hmm.p, s));
        ENG_TRY(hipEventRecord(L->ev[1], s));
    }
    rc = mrp_batch_launch(b);
    b->pre_sweep = nullptr;
    if (rc != MRP_OK) return rc;
    ENG_TRY(hipEventRecord(L->ev[2], s));
    if (L->final_level) {
        ENG_TRY(mrp_launch_traceback(b->dev, L->d_ph.p, n, L->d_err.p, L->d_err_hmm.p, s));
        if (mrp_dup('b')) ENG_TRY(mrp_launch_traceback(b->dev, L->d_ph.p, n, L->d_err.p, L->d_err_hmm.p, s));
        if (L->frag) {
            FragArrays fa{};
            fa.hmms = L->d_frag_hmms.p; fa.reads = L->d_frag_reads.p; fa.by_pool = L->d_frag_by_pool.p; fa.discarded = L->d_frag_disc.p;
            fa.sites = L->d_frag_sites.p; fa.lists = L->d_frag_lists.p; fa.work = L->d_frag_work.p; fa.counts = L->d_frag_counts.p;
            fa.col_read = L->d_frag_col_read.p; fa.col_part = L->d_frag_col_part.p; fa.read_key = L->d_frag_read_key.p; fa.col_cnt = L->d_frag_col_cnt.p;
            ENG_TRY(mrp_launch_fragments(b->dev, L->d_ph.p, n, fa, L->d_err.p, L->d_err_hmm.p, s));
        }
    } else {
        const int64_t n_reg = n - L->n_mini;
        ENG_TRY(mrp_launch_mini(b->dev, L->d_cc.p, L->d_ph.p + n_reg, L->n_mini, n_reg, L->pp, sc, s));
        if (mrp_dup('m')) ENG_TRY(mrp_launch_mini(b->dev, L->d_cc.p, L->d_ph.p + n_reg, L->n_mini, n_reg, L->pp, sc, s));
        ENG_TRY(mrp_launch_prune(b->dev, L->d_cc.p, L->d_ph.p, n_reg, L->pp, sc, s));
        if (mrp_dup('r')) ENG_TRY(mrp_launch_prune(b->dev, L->d_cc.p, L->d_ph.p, n_reg, L->pp, sc, s));
        ENG_TRY(hipEventRecord(L->ev[4], s));
        ENG_TRY(mrp_launch_compact(b->dev, L->d_cc.p, L->d_ph.p, L->d_col_hmm.p, total_cols, n_reg, L->pp, sc, s));
        if (mrp_dup('c')) ENG_TRY(mrp_launch_compact(b->dev, L->d_cc.p, L->d_ph.p, L->d_col_hmm.p, total_cols, n_reg, L->pp, sc, s));
    }
    ENG_TRY(hipEventRecord(L->ev[3], s));
    if (L->final_level) {
        ENG_TRY(hipMemcpyAsync(L->path_cell, seg->n_cells.p, sizeof(int32_t) * (size_t) total_cols, hipMemcpyDeviceToHost, s));
        ENG_TRY(hipMemcpyAsync(L->path_part, seg->part.p, sizeof(uint64_t) * (size_t) total_cols, hipMemcpyDeviceToHost, s));
        ENG_TRY(hipMemcpyAsync(L->fb, b->dev.hmm_fb, sizeof(double) * (size_t) (2 * n), hipMemcpyDeviceToHost, s));
        if (L->frag) {
            ENG_TRY(hipMemcpyAsync(L->h_frag_sites, L->d_frag_sites.p, sizeof(FragSite) * (size_t) L->frag_sites_total, hipMemcpyDeviceToHost, s));
            ENG_TRY(hipMemcpyAsync(L->h_frag_lists, L->d_frag_lists.p, 8 * (size_t) L->frag_list_total, hipMemcpyDeviceToHost, s));
            ENG_TRY(hipMemcpyAsync(L->h_frag_counts, L->d_frag_counts.p, 8 * (size_t) n, hipMemcpyDeviceToHost, s));
        }
    }
    ENG_TRY(hipMemcpyAsync(L->err, L->d_err.p, 16, hipMemcpyDeviceToHost, s));
    ENG_TRY(hipMemcpyAsync(L->err_hmm, L->d_err_hmm.p, sizeof(int32_t) * (size_t) n, hipMemcpyDeviceToHost, s));
#if defined(PRUNE_EXP_CLOCK) || defined(PRUNE_EXP_CLOCK2) || defined(XE_CLOCK)
    ENG_TRY(hipMemcpyAsync(L->clk, L->d_err.p + 4, 96, hipMemcpyDeviceToHost, s));
#endif
    ENG_TRY(hipEventRecord(L->done, s));
    if (L->t_react_ms < 0) L->t_react_ms += eng_now();
    L->t_launched = eng_now();
    if (getenv("MRP_TIMELINE")) e->timeline.emplace_back((long long) L->n, L->t_launched);
    L->t_launch_ms = L->t_launched - t0;
    return MRP_OK;
}

static int level_launch(mrp_engine *e) {
    if (!e->staged) return level_finish(e); /* an empty level still ends the ones before */
    mrp_engine_level_state *L = e->staged;
    e->staged = nullptr;
    int rc = level_launch_impl(e, L);
    if (rc != MRP_OK) { level_retire(e, L); return rc; }
    e->inflight.push_back(L);
    return MRP_OK;
}

extern "C" {

int mrp_engine_level_stage(mrp_engine *e, int64_t n, mrp_xhmm *x) { return level_stage(e, n, x, false); }
int mrp_engine_final_stage(mrp_engine *e, int64_t n, mrp_xhmm *x) { return level_stage(e, n, x, true); }
int mrp_engine_level_launch(mrp_engine *e) { return level_launch(e); }
int mrp_engine_level_end(mrp_engine *e) {
    if (!e) return mrp_set_error(MRP_ERR_ARG, "mrp_engine_level_end: NULL engine");
    return level_finish(e);
}
int64_t mrp_engine_levels_ended(const mrp_engine *e) { return e ? e->n_ended : 0; }

int mrp_engine_level_begin(mrp_engine *e, int64_t n, mrp_xhmm *x) {
    int rc = level_stage(e, n, x, false);
    if (rc == MRP_OK) rc = level_launch(e);
    return rc;
}

int mrp_engine_level(mrp_engine *e, int64_t n, mrp_xhmm *x) {
    int rc = mrp_engine_level_begin(e, n, x);
    if (rc == MRP_OK) rc = mrp_engine_level_end(e);
    return rc;
}

int mrp_engine_final(mrp_engine *e, int64_t n, mrp_xhmm *x) {
    int rc = level_stage(e, n, x, true);
    if (rc == MRP_OK) rc = level_launch(e);
    if (rc == MRP_OK) rc = mrp_engine_level_end(e);
    return rc;
}

}  /* extern "C" */
