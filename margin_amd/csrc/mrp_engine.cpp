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
    DevBuf<uint64_t> part, mfrom, mto;
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
};

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
    (void) hipStreamSynchronize(e->ctx->stream);
    for (auto &ev : e->ev)
        if (ev) (void) hipEventDestroy(ev);
    delete e;
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

int mrp_engine_level(mrp_engine *e, int64_t n, mrp_xhmm *x) {
    if (!e || n < 0 || (n > 0 && !x)) return mrp_set_error(MRP_ERR_ARG, "mrp_engine_level: bad arguments");
    if (n == 0) return MRP_OK;
    mrp_context *ctx = e->ctx;
    ENG_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int S = e->pp.S;
    const bool inv = e->params.include_inverted_partitions != 0;

    int64_t total_cols = 0;
    for (int64_t i = 0; i < n; i++) {
        if (x[i].n_cols < 1 || !x[i].cols || !x[i].n_cells || !x[i].n_merge) return mrp_set_error(MRP_ERR_ARG, "mrp_engine_level: bad hmm %lld", (long long) i);
        total_cols += x[i].n_cols;
    }
    std::unique_ptr<Segment> seg(new (std::nothrow) Segment());
    if (!seg) return mrp_set_error(MRP_ERR_NOMEM, "out of host memory");
    ENG_TRY(seg->part.alloc((size_t) total_cols * S));
    ENG_TRY(seg->mfrom.alloc((size_t) total_cols * S));
    ENG_TRY(seg->mto.alloc((size_t) total_cols * S));
    ENG_TRY(seg->np.alloc((size_t) total_cols * S));
    ENG_TRY(seg->n_cells.alloc((size_t) total_cols));
    ENG_TRY(seg->n_merge.alloc((size_t) total_cols));

    mrp_batch *b = nullptr;
    int rc = mrp_batch_create(ctx, &b);
    if (rc != MRP_OK) return rc;
    struct BatchGuard { mrp_batch *b; ~BatchGuard() { mrp_batch_destroy(b); } } guard{b};

    std::vector<CrossCol> cc((size_t) total_cols);
    std::vector<PruneHmm> ph((size_t) n);
    std::vector<int32_t> col_hmm((size_t) total_cols);
    std::vector<uint64_t> mask_from((size_t) total_cols, 0), mask_to((size_t) total_cols, 0);
    std::vector<int64_t> cell_off, mcell_off;
    int64_t colbase = 0, level_cells = 0, level_merge = 0;
    for (int64_t i = 0; i < n; i++) {
        mrp_xhmm &h = x[i];
        const int K = h.n_cols;
        cell_off.assign((size_t) K + 1, 0);
        mcell_off.assign((size_t) K, 0);
        for (int k = 0; k < K; k++) {
            const mrp_xcol &c = h.cols[k];
            const int64_t C = (int64_t) c.C1 * c.C2;
            if (C < 1 || C > MRP_PRUNE_MAX_CELLS) return mrp_set_error(MRP_ERR_UNSUPPORTED, "cross product column with %lld cells", (long long) C);
            cell_off[k + 1] = cell_off[k] + C;
            if (k + 1 < K) {
                const int64_t M = (int64_t) c.Ma * c.Mb;
                if (M < 1 || M > 65535) return mrp_set_error(MRP_ERR_UNSUPPORTED, "cross product merge column with %lld cells", (long long) M);
                mcell_off[k + 1] = mcell_off[k] + M;
            }
        }
        mrp_hmm_job job;
        memset(&job, 0, sizeof(job));
        job.chunk = h.chunk;
        job.n_columns = K;
        job.flags = h.flags;
        job.col_ref_start = h.col_ref_start;
        job.col_length = h.col_length;
        job.col_depth = h.col_depth;
        job.col_cell_off = cell_off.data();
        job.col_read_off = h.col_read_off;
        job.read_byte_off = h.read_byte_off;
        job.mcol_cell_off = mcell_off.data();
        int64_t cell0 = 0, mcell0 = 0, col0 = 0;
        rc = mrp_batch_add_impl(b, &job, true, &cell0, &mcell0, &col0);
        if (rc != MRP_OK) return rc;
        if (col0 != colbase) return mrp_set_error(MRP_ERR_ARG, "engine: column bookkeeping out of step");
        for (int k = 0; k < K; k++) {
            const mrp_xcol &c = h.cols[k];
            CrossCol &o = cc[(size_t) (colbase + k)];
            memset(&o, 0, sizeof(o));
            o.a_part = c.a_part; o.b_part = c.b_part; o.a_np = c.a_np; o.b_np = c.b_np;
            o.x_cell_off = cell0 + cell_off[k];
            o.C1 = c.C1; o.C2 = c.C2; o.d1 = c.d1; o.d2 = c.d2;
            uint8_t fl = inv ? MRP_XF_INVERTED : 0;
            if (k + 1 < K) {
                o.Ma = c.Ma; o.Mb = c.Mb; o.out_a = c.out_a; o.out_b = c.out_b;
                if (c.out_a_paired) fl |= MRP_XF_OUT_A_PAIRED;
                if (c.out_b_paired) fl |= MRP_XF_OUT_B_PAIRED;
                mask_from[(size_t) (colbase + k)] = c.mask_from;
                mask_to[(size_t) (colbase + k)] = c.mask_to;
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
        p.out_part = seg->part.p + colbase * S;
        p.out_np = seg->np.p + colbase * S;
        p.out_mfrom = seg->mfrom.p + colbase * S;
        p.out_mto = seg->mto.p + colbase * S;
        p.out_n_cells = seg->n_cells.p + colbase;
        p.out_n_merge = seg->n_merge.p + colbase;
        h.d_part = p.out_part; h.d_np = p.out_np; h.d_mfrom = p.out_mfrom; h.d_mto = p.out_mto;
        colbase += K;
        level_cells += cell_off[K];
        level_merge += mcell_off[K - 1];
    }

    rc = mrp_batch_upload(b);
    if (rc != MRP_OK) return rc;
    PruneParams pp = e->pp;
    pp.max_cells = 1; pp.max_merge = 1;
    for (size_t i = 0; i < b->hmms.size(); i++) {
        if (!b->outs[i].int_path) return mrp_set_error(MRP_ERR_UNSUPPORTED, "cross product hmm outside the int32 recursion kernel's range");
        pp.max_cells = std::max(pp.max_cells, b->hmms[i].max_cells);
        pp.max_merge = std::max(pp.max_merge, b->hmms[i].max_merge);
    }

    DevBuf<CrossCol> d_cc;
    DevBuf<PruneHmm> d_ph;
    DevBuf<int32_t> d_col_hmm, d_nkept, d_nkeptm, d_err;
    DevBuf<uint64_t> d_mask_from, d_mask_to;
    DevBuf<uint16_t> d_kept, d_keptm;
    DevBuf<uint32_t> d_kept_np;
    ENG_TRY(d_cc.upload(cc, s));
    ENG_TRY(d_ph.upload(ph, s));
    ENG_TRY(d_col_hmm.upload(col_hmm, s));
    ENG_TRY(d_mask_from.upload(mask_from, s));
    ENG_TRY(d_mask_to.upload(mask_to, s));
    ENG_TRY(d_kept.alloc((size_t) total_cols * S));
    ENG_TRY(d_keptm.alloc((size_t) total_cols * S));
    ENG_TRY(d_kept_np.alloc((size_t) total_cols * S));
    ENG_TRY(d_nkept.alloc((size_t) total_cols));
    ENG_TRY(d_nkeptm.alloc((size_t) total_cols));
    ENG_TRY(d_err.alloc(4));
    ENG_TRY(hipMemsetAsync(d_err.p, 0, 16, s));
    PruneScratch sc{};
    sc.kept = d_kept.p; sc.kept_np = d_kept_np.p; sc.keptm = d_keptm.p; sc.n_kept = d_nkept.p; sc.n_keptm = d_nkeptm.p;
    sc.mask_from = d_mask_from.p; sc.mask_to = d_mask_to.p; sc.err = d_err.p;

    ENG_TRY(hipEventRecord(e->ev[0], s));
    ENG_TRY(mrp_launch_cross(d_cc.p, total_cols, b->d_partition.p, b->d_np.p, d_err.p, s));
    ENG_TRY(hipEventRecord(e->ev[1], s));
    rc = mrp_batch_launch(b);
    if (rc != MRP_OK) return rc;
    ENG_TRY(hipEventRecord(e->ev[2], s));
    ENG_TRY(mrp_launch_prune(b->dev, d_ph.p, n, pp, sc, s));
    ENG_TRY(mrp_launch_compact(b->dev, d_ph.p, d_col_hmm.p, total_cols, pp, sc, s));
    ENG_TRY(hipEventRecord(e->ev[3], s));

    std::vector<int32_t> nc((size_t) total_cols), nm((size_t) total_cols);
    int32_t err[4] = {0, 0, 0, 0};
    ENG_TRY(hipMemcpyAsync(nc.data(), seg->n_cells.p, sizeof(int32_t) * (size_t) total_cols, hipMemcpyDeviceToHost, s));
    ENG_TRY(hipMemcpyAsync(nm.data(), seg->n_merge.p, sizeof(int32_t) * (size_t) total_cols, hipMemcpyDeviceToHost, s));
    ENG_TRY(hipMemcpyAsync(err, d_err.p, sizeof(err), hipMemcpyDeviceToHost, s));
    ENG_TRY(hipStreamSynchronize(s));
    if (err[0] & MRP_ENGINE_ERR_POSTERIOR) return mrp_set_error(MRP_ERR_ARG, "ERROR: invalid prob (f + b exceeds the column total)");
    if (err[0] & MRP_ENGINE_ERR_RANGE) return mrp_set_error(MRP_ERR_LOOKUP, "device-resident merge: transition index out of range");
    if (err[0] & MRP_ENGINE_ERR_STRUCTURE)
        return mrp_set_error(MRP_ERR_UNSUPPORTED, "device-resident merge: a parent hmm is not in complement-pair order");
    colbase = 0;
    for (int64_t i = 0; i < n; i++) {
        memcpy(x[i].n_cells, nc.data() + colbase, sizeof(int32_t) * (size_t) x[i].n_cols);
        memcpy(x[i].n_merge, nm.data() + colbase, sizeof(int32_t) * (size_t) x[i].n_cols);
        colbase += x[i].n_cols;
    }
    float t_cross = 0, t_sweep = 0, t_prune = 0;
    ENG_TRY(hipEventElapsedTime(&t_cross, e->ev[0], e->ev[1]));
    ENG_TRY(hipEventElapsedTime(&t_sweep, e->ev[1], e->ev[2]));
    ENG_TRY(hipEventElapsedTime(&t_prune, e->ev[2], e->ev[3]));
    e->stats.levels += 1;
    e->stats.hmms += n;
    e->stats.columns += total_cols;
    e->stats.cells += level_cells;
    e->stats.merge_cells += level_merge;
    e->stats.cross_ms += t_cross;
    e->stats.sweep_ms += t_sweep;
    e->stats.prune_ms += t_prune;
    e->stats.device_ms += t_cross + t_sweep + t_prune;
    e->segments.push_back(std::move(seg));
    return MRP_OK;
}

}  /* extern "C" */
