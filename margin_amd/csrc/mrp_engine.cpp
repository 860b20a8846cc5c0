/*
 * mrp_engine.cpp -- host side of the device-resident merge level.
 *
 * One call of mrp_engine_level() performs, for a set of independent overlap components
 * (coordination.c:285-328), what the reference does with
 *     stRPHmm_createCrossProductOfTwoAlignedHmm (hmm.c:534)  ->  mrp_cross_kernel
 *     stRPHmm_forwardBackward (hmm.c:931)                    ->  plane / emission / recursion kernels
 *     stRPHmm_prune (hmm.c:1160)                             ->  mrp_prune_kernel + mrp_compact_kernel
 * without the hmm leaving HBM: the parents are read from, and the pruned result is written to, the
 * fixed-stride resident layout of mrp_engine.h; only the per-column cell counts (4 B per column) come
 * back to the host, which needs them to lay out the next level.  The structural decisions (tiling
 * paths, overlap components, column alignment) are made by rphmm_host.c from read intervals alone.
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
#include "mrp_internal.h"

#define ENG_TRY(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) return mrp_set_error(MRP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace {
struct Segment { /* the pruned hmms produced by one level */
    DevBuf<uint64_t> part;
    DevBuf<uint32_t> np;
    DevBuf<int32_t> n_cells, n_merge;
};
}  // namespace

struct mrp_engine {
    mrp_context *ctx = nullptr;
    mrp_params params{};
    PruneParams pp{};
    DevBuf<uint64_t> leaf_part;
    DevBuf<uint32_t> leaf_np;
    std::vector<std::unique_ptr<Segment>> segments;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    mrp_engine_stats stats{};
    struct mrp_engine_level_state *cur = nullptr; /* a level between begin and end */
    mrp_batch *spare = nullptr;                   /* the batch object of the previous level, emptied */
};

static double eng_now() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return 1e3 * ts.tv_sec + 1e-6 * ts.tv_nsec;
}

static void mrp_engine_level_abandon(mrp_engine *e);

extern "C" {

int mrp_engine_create(mrp_context *ctx, const mrp_params *params, mrp_engine **out) {
    if (!ctx || !params || !out) return mrp_set_error(MRP_ERR_ARG, "mrp_engine_create: NULL argument");
    *out = nullptr;
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
    PruneParams &pp = e->pp;
    pp.S = (int32_t) ((lim + 3) & ~3ll);
    pp.min_p = (int32_t) params->min_partitions_in_a_column;
    pp.max_p = (int32_t) std::min<int64_t>(params->max_partitions_in_a_column, 1 << 20);
    if (pp.max_p < 0) pp.max_p = 0;
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
    {
        std::lock_guard<std::mutex> lock(ctx->sibling_mu);
        e->spare = ctx->spare_batch;
        ctx->spare_batch = nullptr;
    }
    hipError_t he = e->leaf_part.alloc(4);
    if (he == hipSuccess) he = e->leaf_np.alloc(4);
    const uint64_t lp[4] = {1, 0, 0, 0}; /* stRPHmm_construct hmm.c:97-133 */
    const uint32_t ln[4] = {0, 0, 0, 0};
    if (he == hipSuccess) he = hipMemcpy(e->leaf_part.p, lp, sizeof(lp), hipMemcpyHostToDevice);
    if (he == hipSuccess) he = hipMemcpy(e->leaf_np.p, ln, sizeof(ln), hipMemcpyHostToDevice);
    for (int i = 0; i < 4 && he == hipSuccess; i++) he = hipEventCreate(&e->ev[i]);
    if (he != hipSuccess) {
        mrp_engine_destroy(e);
        return mrp_set_error(MRP_ERR_HIP, "engine setup failed: %s", hipGetErrorString(he));
    }
    *out = e;
    return MRP_OK;
}

void mrp_engine_destroy(mrp_engine *e) {
    if (!e) return;
    (void) hipSetDevice(e->ctx->device);
    const double t0 = eng_now();
    (void) hipStreamSynchronize(e->ctx->stream);
    const double t1 = eng_now();
    for (auto &ev : e->ev)
        if (ev) (void) hipEventDestroy(ev);
    mrp_context *ctx = e->ctx;
    mrp_engine_level_abandon(e);
    const double t2 = eng_now();
    if (e->spare) { /* kept for the next engine of this context */
        std::lock_guard<std::mutex> lock(ctx->sibling_mu);
        if (!ctx->spare_batch) { ctx->spare_batch = e->spare; e->spare = nullptr; }
    }
    if (e->spare) mrp_batch_destroy(e->spare);
    const double t3 = eng_now();
    delete e;
    const double t4 = eng_now();
    ctx->pool.reclaim();
    if (getenv("MRP_TIMING"))
        fprintf(stderr, "  engine destroy: sync %.1f ms, events %.1f, spare batch %.1f, segments %.1f, reclaim %.1f\n", t1 - t0, t2 - t1, t3 - t2, t4 - t3,
                eng_now() - t4);
}

int32_t mrp_engine_stride(const mrp_engine *e) { return e->pp.S; }

void mrp_engine_leaf(const mrp_engine *e, const uint64_t **part, const uint32_t **np) {
    *part = e->leaf_part.p;
    *np = e->leaf_np.p;
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
    ENG_TRY(hipStreamSynchronize(e->ctx->stream));
    return MRP_OK;
}

}  /* extern "C" */

/* everything one level keeps between its launch and its completion */
struct mrp_engine_level_state {
    mrp_batch *b = nullptr;
    std::unique_ptr<Segment> seg;
    DevBuf<CrossCol> d_cc;
    DevBuf<PruneHmm> d_ph;
    DevBuf<int32_t> d_col_hmm, d_nkept, d_nkeptm, d_err, d_err_hmm;
    DevBuf<uint16_t> d_kept, d_keptm;
    DevBuf<uint32_t> d_kept_np;
    /* results, in the context's page-locked staging buffer (so the copies are asynchronous) */
    int32_t *nc = nullptr, *nm = nullptr, *err = nullptr, *err_hmm = nullptr;
    std::vector<int32_t> perm; /* position in the (sorted) PruneHmm array -> index into x */
    uint64_t *path_part = nullptr; /* final level */
    double *fb = nullptr;
    bool final_level = false;
    unsigned long long clk[12] = {0};
    mrp_xhmm *x = nullptr;
    int64_t n = 0, total_cols = 0, level_cells = 0, level_merge = 0;
    double t_begin = 0, t_launched = 0;
    mrp_engine *owner = nullptr;
    ~mrp_engine_level_state() {
        if (!b) return;
        mrp_context *ctx = b->ctx;
        (void) hipSetDevice(ctx->device);
        (void) hipStreamSynchronize(ctx->stream); /* before the buffers go back to the pool */
        if (owner && !owner->spare) {
            b->recycle();
            owner->spare = b;
        } else {
            mrp_batch_destroy(b);
        }
        ctx->pool.reclaim();
    }
};

static void mrp_engine_level_abandon(mrp_engine *e) {
    delete e->cur;
    e->cur = nullptr;
}

extern "C" {


static int level_begin(mrp_engine *e, int64_t n, mrp_xhmm *x, bool final_level) {
    if (!e || n < 0 || (n > 0 && !x)) return mrp_set_error(MRP_ERR_ARG, "mrp_engine_level: bad arguments");
    if (e->cur) return mrp_set_error(MRP_ERR_ARG, "mrp_engine_level_begin: the previous level was not completed");
    if (n == 0) return MRP_OK;
    mrp_context *ctx = e->ctx;
    ENG_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int S = e->pp.S;
    const bool inv = e->params.include_inverted_partitions != 0;
    std::unique_ptr<mrp_engine_level_state> L(new (std::nothrow) mrp_engine_level_state());
    if (!L) return mrp_set_error(MRP_ERR_NOMEM, "out of host memory");
    L->t_begin = eng_now();
    L->x = x;
    L->n = n;

    int64_t total_cols = 0;
    for (int64_t i = 0; i < n; i++) {
        if (x[i].n_cols < 1 || !x[i].cols || !x[i].n_cells || (!final_level && !x[i].n_merge)) return mrp_set_error(MRP_ERR_ARG, "mrp_engine_level: bad hmm %lld", (long long) i);
        total_cols += x[i].n_cols;
        for (int k = 0; k < x[i].n_cols; k++) { /* range checks the kernels rely on */
            const mrp_xcol &c = x[i].cols[k];
            const int64_t C = (int64_t) c.C1 * c.C2, M = (int64_t) c.Ma * c.Mb;
            if (C < 1 || C > MRP_PRUNE_MAX_CELLS) return mrp_set_error(MRP_ERR_UNSUPPORTED, "cross product column with %lld cells", (long long) C);
            if (k + 1 < x[i].n_cols && (M < 1 || M > MRP_PRUNE_MAX_CELLS)) return mrp_set_error(MRP_ERR_UNSUPPORTED, "cross product merge column with %lld cells", (long long) M);
        }
    }
    L->total_cols = total_cols;
    ENG_TRY(hipStreamSynchronize(s)); /* nothing of an earlier level or sweep is in flight: */
    ctx->pool.reclaim();              /* blocks released since then can be reused */
    L->seg.reset(new (std::nothrow) Segment());
    if (!L->seg) return mrp_set_error(MRP_ERR_NOMEM, "out of host memory");
    Segment *seg = L->seg.get();
    seg->part.pool = &ctx->pool;
    seg->np.pool = &ctx->pool;
    seg->n_cells.pool = seg->n_merge.pool = &ctx->pool;
    const int64_t out_stride = final_level ? 1 : S; /* the final level keeps one traced-back cell per column */
    ENG_TRY(seg->part.alloc((size_t) (total_cols * out_stride)));
    ENG_TRY(seg->np.alloc((size_t) (final_level ? 1 : total_cols * S)));
    ENG_TRY(seg->n_cells.alloc((size_t) total_cols));
    ENG_TRY(seg->n_merge.alloc((size_t) total_cols));

    const double tA = eng_now();
    int rc = MRP_OK;
    L->owner = e;
    if (e->spare) { L->b = e->spare; e->spare = nullptr; }
    else rc = mrp_batch_create(ctx, &L->b);
    if (rc != MRP_OK) return rc;
    mrp_batch *b = L->b;
    std::vector<int64_t> cell0((size_t) n), col0((size_t) n);
    rc = mrp_batch_add_resident_bulk(b, n, x, cell0.data(), col0.data());
    if (rc != MRP_OK) return rc;
    const double tB = eng_now();
    HostVec<CrossCol> cc((size_t) total_cols); /* filled entirely below */
    HostVec<PruneHmm> ph((size_t) n);
    HostVec<int32_t> col_hmm((size_t) total_cols);
    mrp_parallel_for(n, std::max<int64_t>(1, n / 256), [&](int64_t i) {
        mrp_xhmm &h = x[i];
        const int K = h.n_cols;
        const int64_t colbase = col0[(size_t) i];
        int64_t c_off = cell0[(size_t) i];
        for (int k = 0; k < K; k++) {
            const mrp_xcol &c = h.cols[k];
            CrossCol &o = cc[(size_t) (colbase + k)];
            memset(&o, 0, sizeof(o));
            o.a_part = c.a_part; o.b_part = c.b_part; o.a_np = c.a_np; o.b_np = c.b_np;
            o.x_cell_off = c_off;
            c_off += (int64_t) c.C1 * c.C2;
            o.C1 = c.C1; o.C2 = c.C2; o.d1 = c.d1; o.d2 = c.d2;
            uint8_t fl = inv ? MRP_XF_INVERTED : 0;
            if (k + 1 < K) {
                o.Ma = c.Ma; o.Mb = c.Mb; o.out_a = c.out_a; o.out_b = c.out_b;
                if (c.out_a_paired) fl |= MRP_XF_OUT_A_PAIRED;
                if (c.out_b_paired) fl |= MRP_XF_OUT_B_PAIRED;
            }
            if (k > 0) {
                const mrp_xcol &q = h.cols[k - 1];
                o.Pa = q.Ma; o.Pb = q.Mb; o.in_a = q.out_a; o.in_b = q.out_b;
                if (q.out_a_paired) fl |= MRP_XF_IN_A_PAIRED;
                if (q.out_b_paired) fl |= MRP_XF_IN_B_PAIRED;
            }
            o.flags = fl;
            col_hmm[(size_t) (colbase + k)] = (int32_t) i;
        }
        PruneHmm &p = ph[(size_t) i];
        p.col0 = colbase;
        p.n_cols = K;
        p.hmm_index = (int32_t) i;
        p.out_part = seg->part.p + colbase * out_stride;
        p.out_np = final_level ? seg->np.p : seg->np.p + colbase * S;
        p.out_n_cells = seg->n_cells.p + colbase;
        p.out_n_merge = seg->n_merge.p + colbase;
        h.d_part = p.out_part; h.d_np = p.out_np;
    });
    L->level_cells = b->stats.n_cells;
    L->level_merge = b->stats.n_merge_cells;
    /* the prune kernel walks one hmm per workgroup and its columns one after the other: longest hmms first, so that the
     * launch does not end with a long chain that started late */
    {
        std::vector<int32_t> perm((size_t) n), pos((size_t) n);
        for (int64_t i = 0; i < n; i++) perm[(size_t) i] = (int32_t) i;
        std::stable_sort(perm.begin(), perm.end(), [&](int32_t a, int32_t c) { return ph[(size_t) a].n_cols > ph[(size_t) c].n_cols; });
        HostVec<PruneHmm> sorted((size_t) n);
        for (int64_t j = 0; j < n; j++) { sorted[(size_t) j] = ph[(size_t) perm[(size_t) j]]; pos[(size_t) perm[(size_t) j]] = (int32_t) j; }
        ph.swap(sorted);
        L->perm.swap(perm);
        mrp_parallel_for(total_cols, 65536, [&](int64_t c) { col_hmm[(size_t) c] = pos[(size_t) col_hmm[(size_t) c]]; });
    }

    const double tC = eng_now();
    rc = mrp_batch_upload(b);
    if (rc != MRP_OK) return rc;
    const double tD = eng_now();
    PruneParams pp = e->pp;
    pp.max_cells = 1; pp.max_merge = 1;
    pp.pad = (e->params.reserved & 1) && e->stats.levels == 1 ? 1 : 0; /* test hook, see mrp_params.reserved */
    for (size_t i = 0; i < b->hmms.size(); i++) {
        if (!b->outs[i].int_path) return mrp_set_error(MRP_ERR_UNSUPPORTED, "cross product hmm outside the int32 recursion kernel's range");
        pp.max_cells = std::max(pp.max_cells, b->hmms[i].max_cells);
        pp.max_merge = std::max(pp.max_merge, b->hmms[i].max_merge);
    }
    if (getenv("MRP_TIMING")) {
        int64_t ns = 0, cs = 0, nb2 = 0, cb = 0, longest_s = 0, longest_b = 0;
        for (size_t i = 0; i < b->hmms.size(); i++) {
            if (b->hmms[i].max_cells <= 4096) { ns++; cs += b->hmms[i].n_cols; longest_s = std::max<int64_t>(longest_s, b->hmms[i].n_cols); }
            else { nb2++; cb += b->hmms[i].n_cols; longest_b = std::max<int64_t>(longest_b, b->hmms[i].n_cols); }
        }
        fprintf(stderr, "    prune classes: <=4096 cells/column: %lld hmms %lld cols (longest %lld); larger: %lld hmms %lld cols (longest %lld)\n",
                (long long) ns, (long long) cs, (long long) longest_s, (long long) nb2, (long long) cb, (long long) longest_b);
    }
    DevPool *pl = &ctx->pool;
    L->d_cc.pool = pl; L->d_ph.pool = pl; L->d_col_hmm.pool = L->d_nkept.pool = L->d_nkeptm.pool = L->d_err.pool = L->d_err_hmm.pool = pl;
    L->d_kept.pool = L->d_keptm.pool = pl; L->d_kept_np.pool = pl;
    ENG_TRY(L->d_cc.upload(cc, s));
    ENG_TRY(L->d_ph.upload(ph, s));
    ENG_TRY(L->d_col_hmm.upload(col_hmm, s));
    if (!final_level) {
        ENG_TRY(L->d_kept.alloc((size_t) total_cols * S));
        ENG_TRY(L->d_keptm.alloc((size_t) total_cols * S));
        ENG_TRY(L->d_kept_np.alloc((size_t) total_cols * S));
        ENG_TRY(L->d_nkept.alloc((size_t) total_cols));
        ENG_TRY(L->d_nkeptm.alloc((size_t) total_cols));
    }
    ENG_TRY(L->d_err.alloc(64));
    ENG_TRY(hipMemsetAsync(L->d_err.p, 0, 256, s));
    ENG_TRY(L->d_err_hmm.alloc((size_t) n));
    ENG_TRY(hipMemsetAsync(L->d_err_hmm.p, 0, sizeof(int32_t) * (size_t) n, s));
    PruneScratch sc{};
    sc.kept = L->d_kept.p; sc.kept_np = L->d_kept_np.p; sc.keptm = L->d_keptm.p; sc.n_kept = L->d_nkept.p; sc.n_keptm = L->d_nkeptm.p;
    sc.err = L->d_err.p;
    sc.err_hmm = L->d_err_hmm.p;
    /* the pageable host vectors above are read by the queued copies: wait for them before they go out of scope */
    ENG_TRY(hipStreamSynchronize(s));

    const double tE = eng_now();
    ENG_TRY(hipEventRecord(e->ev[0], s));
    ENG_TRY(mrp_launch_cross(L->d_cc.p, total_cols, b->d_partition.p, b->d_np.p, L->d_err.p, L->d_col_hmm.p, L->d_err_hmm.p, s));
    ENG_TRY(hipEventRecord(e->ev[1], s));
    rc = mrp_batch_launch(b);
    if (rc != MRP_OK) return rc;
    ENG_TRY(hipEventRecord(e->ev[2], s));
    if (final_level) {
        ENG_TRY(mrp_launch_traceback(b->dev, L->d_ph.p, n, L->d_err.p, L->d_err_hmm.p, s));
    } else {
        ENG_TRY(mrp_launch_prune(b->dev, L->d_cc.p, L->d_ph.p, n, pp, sc, s));
        ENG_TRY(mrp_launch_compact(b->dev, L->d_ph.p, L->d_col_hmm.p, total_cols, pp, sc, s));
    }
    ENG_TRY(hipEventRecord(e->ev[3], s));
    L->final_level = final_level;
    {
        const size_t cols8 = ((size_t) total_cols + 1) & ~(size_t) 1; /* keep the 8-byte arrays aligned */
        const size_t n8 = ((size_t) n + 1) & ~(size_t) 1;
        ENG_TRY(ctx->pinned_reserve(64 + cols8 * 4 * 2 + cols8 * 8 + (size_t) n * 16 + n8 * 4));
        char *base = (char *) ctx->pinned;
        L->err = (int32_t *) base;
        L->path_part = (uint64_t *) (base + 64);
        L->fb = (double *) (base + 64 + cols8 * 8);
        L->nc = (int32_t *) (base + 64 + cols8 * 8 + (size_t) n * 16);
        L->nm = L->nc + cols8;
        L->err_hmm = L->nm + cols8;
    }
    ENG_TRY(hipMemcpyAsync(L->nc, seg->n_cells.p, sizeof(int32_t) * (size_t) total_cols, hipMemcpyDeviceToHost, s));
    if (final_level) {
        ENG_TRY(hipMemcpyAsync(L->path_part, seg->part.p, sizeof(uint64_t) * (size_t) total_cols, hipMemcpyDeviceToHost, s));
        ENG_TRY(hipMemcpyAsync(L->fb, b->dev.hmm_fb, sizeof(double) * (size_t) (2 * n), hipMemcpyDeviceToHost, s));
    } else {
        ENG_TRY(hipMemcpyAsync(L->nm, seg->n_merge.p, sizeof(int32_t) * (size_t) total_cols, hipMemcpyDeviceToHost, s));
    }
    ENG_TRY(hipMemcpyAsync(L->err, L->d_err.p, 16, hipMemcpyDeviceToHost, s));
    ENG_TRY(hipMemcpyAsync(L->err_hmm, L->d_err_hmm.p, sizeof(int32_t) * (size_t) n, hipMemcpyDeviceToHost, s));
#if defined(PRUNE_EXP_CLOCK) || defined(PRUNE_EXP_CLOCK2)
    ENG_TRY(hipMemcpyAsync(L->clk, L->d_err.p + 4, 96, hipMemcpyDeviceToHost, s));
#endif
    L->t_launched = eng_now();
    if (getenv("MRP_TIMING"))
        fprintf(stderr, "    begin: checks+segment %.1f ms, bulk add %.1f, cross descriptors %.1f, batch upload %.1f, engine upload %.1f, launches %.1f\n",
                tA - L->t_begin, tB - tA, tC - tB, tD - tC, tE - tD, L->t_launched - tE);
    e->cur = L.release();
    return MRP_OK;
}

int mrp_engine_level_end(mrp_engine *e) {
    if (!e) return mrp_set_error(MRP_ERR_ARG, "mrp_engine_level_end: NULL engine");
    if (!e->cur) return MRP_OK; /* an empty level */
    std::unique_ptr<mrp_engine_level_state> L(e->cur);
    e->cur = nullptr;
    mrp_context *ctx = e->ctx;
    ENG_TRY(hipSetDevice(ctx->device));
    ENG_TRY(hipStreamSynchronize(ctx->stream));
    if (getenv("MRP_TIMING"))
        fprintf(stderr, "  level: %lld hmms %lld cols %lld cells: host build + upload %.1f ms, kernels (after launch) %.1f ms\n",
                (long long) L->n, (long long) L->total_cols, (long long) L->level_cells, L->t_launched - L->t_begin, eng_now() - L->t_launched);
#if defined(PRUNE_EXP_CLOCK) || defined(PRUNE_EXP_CLOCK2)
    fprintf(stderr, "  prune clocks (first hmm; work, wait per role: chain, lists 1, lists 2, tables, bins group 0, bins group 1; shader cycles):");
    for (int i = 0; i < 12; i++) fprintf(stderr, " %llu", L->clk[i]);
    fprintf(stderr, "\n");
#endif
    /* Per hmm: a parent outside the closed-form cross product's pair order, or a merge cell the kept cells lead to that
     * hmm.c:1090-1100 would drop, means "not handled on the device": the caller redoes that hmm's chunk on the hashing path
     * (whatever else the kernels flagged for it came from the discarded arrays).  Posterior / range violations on an hmm
     * that is otherwise fine are the reference's st_errAbort cases. */
    for (int64_t i = 0; i < L->n; i++) L->x[i].err = 0;
    if (L->err[0] != 0) {
        for (int64_t j = 0; j < L->n; j++) {
            const int32_t bits = L->err_hmm[j];
            if (bits == 0) continue;
            L->x[(size_t) L->perm[(size_t) j]].err = bits;
            if (bits & (MRP_ENGINE_ERR_STRUCTURE | MRP_ENGINE_ERR_MERGE)) continue;
            if (bits & MRP_ENGINE_ERR_POSTERIOR) return mrp_set_error(MRP_ERR_ARG, "ERROR: invalid prob (f + b exceeds the column total)");
            return mrp_set_error(MRP_ERR_LOOKUP, "device-resident merge: transition index out of range");
        }
    }
    int64_t colbase = 0;
    for (int64_t i = 0; i < L->n; i++) {
        memcpy(L->x[i].n_cells, L->nc + colbase, sizeof(int32_t) * (size_t) L->x[i].n_cols);
        if (L->final_level) {
            if (L->x[i].path_part) memcpy(L->x[i].path_part, L->path_part + colbase, sizeof(uint64_t) * (size_t) L->x[i].n_cols);
            L->x[i].hmm_forward = L->fb[(size_t) (2 * i)];
            L->x[i].hmm_backward = L->fb[(size_t) (2 * i + 1)];
        } else {
            memcpy(L->x[i].n_merge, L->nm + colbase, sizeof(int32_t) * (size_t) L->x[i].n_cols);
        }
        colbase += L->x[i].n_cols;
    }
    float t_cross = 0, t_sweep = 0, t_prune = 0;
    ENG_TRY(hipEventElapsedTime(&t_cross, e->ev[0], e->ev[1]));
    ENG_TRY(hipEventElapsedTime(&t_sweep, e->ev[1], e->ev[2]));
    ENG_TRY(hipEventElapsedTime(&t_prune, e->ev[2], e->ev[3]));
    e->stats.levels += 1;
    e->stats.hmms += L->n;
    e->stats.columns += L->total_cols;
    e->stats.cells += L->level_cells;
    e->stats.merge_cells += L->level_merge;
    e->stats.cross_ms += t_cross;
    e->stats.sweep_ms += t_sweep;
    e->stats.prune_ms += t_prune;
    e->stats.device_ms += t_cross + t_sweep + t_prune;
    e->segments.push_back(std::move(L->seg));
    return MRP_OK;
}

int mrp_engine_level_begin(mrp_engine *e, int64_t n, mrp_xhmm *x) { return level_begin(e, n, x, false); }

int mrp_engine_level(mrp_engine *e, int64_t n, mrp_xhmm *x) {
    int rc = level_begin(e, n, x, false);
    if (rc == MRP_OK) rc = mrp_engine_level_end(e);
    return rc;
}

int mrp_engine_final(mrp_engine *e, int64_t n, mrp_xhmm *x) {
    int rc = level_begin(e, n, x, true);
    if (rc == MRP_OK) rc = mrp_engine_level_end(e);
    return rc;
}

}  /* extern "C" */
