/*
 * mrp_api.cpp -- host side of the C-ABI declared in include/margin_rphmm.h.
 *
 * Validates and concatenates flattened stRPHmm jobs into batch-wide arrays (mrp_device.h),
 * moves them to HBM, launches the plane and sweep kernels on the context's stream and scatters
 * the post-conditions of stRPHmm_forwardBackward (hmm.c:931-942) back into the caller's arrays.
 * There is no CPU implementation of the sweep in this library: without a HIP device every entry
 * point that computes returns MRP_ERR_NO_DEVICE.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "mrp_internal.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(MRP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

}  // namespace

extern "C" {

void mrp_chunk_host_view(const mrp_chunk *chunk, mrp_chunk_host *out) {
    out->n_sites = chunk->n_sites;
    out->allele_number = chunk->allele_number.data();
    out->allele_offset = chunk->allele_offset.data();
    out->sub_offset = chunk->sub_offset.data();
    out->sub = chunk->sub.data();
    out->prior = chunk->prior.data();
    out->pool = chunk->pool_host;
    out->pool_bytes = chunk->pool_bytes;
}
mrp_context *mrp_chunk_context(const mrp_chunk *chunk) { return chunk->ctx; }
int mrp_context_device(const mrp_context *ctx) { return ctx->device; }
/* Said once per process, when a call first runs concurrent batches: the runtime deals HIP streams onto GPU_MAX_HW_QUEUES hardware
 * queues (default 4) and the kernels of streams that share a queue serialize -- eight concurrent batches on two streams each
 * want 16 (mrp_runtime_init() before the first HIP call, or the variable in the environment). */
void mrp_warn_hw_queues_once(int concurrent_batches) {
    static std::atomic<int> said{0};
    if (concurrent_batches <= 2 || getenv("MRP_QUIET")) return;
    const char *q = getenv("GPU_MAX_HW_QUEUES");
    const int have = q ? atoi(q) : 4;
    if (have >= 2 * concurrent_batches || have >= 16) return;
    if (said.exchange(1)) return;
    fprintf(stderr, "margin_rphmm: %d concurrent batches (two streams each) on %d hardware queues (GPU_MAX_HW_QUEUES%s): their kernels share "
                    "queues and serialize; call mrp_runtime_init() before the process's first HIP call, or set GPU_MAX_HW_QUEUES=16\n",
            concurrent_batches, have, q ? "" : " unset, the runtime's default");
}

int mrp_context_set_grouped(mrp_context *ctx, int grouped) { const int was = ctx->grouped ? 1 : 0; ctx->grouped = grouped != 0; return was; }
void mrp_context_set_concurrent_batches(mrp_context *ctx, int n) { ctx->concurrent_batches = n < 1 ? 1 : n; }
int mrp_context_calls_sharing_device(const mrp_context *ctx) { return ctx->calls_sharing_device; }
int64_t mrp_context_device_budget(mrp_context *ctx) { /* bytes the pools of the context's device may hold together */
    if (ctx->pool.device < 0 || hipSetDevice(ctx->device) != hipSuccess) return 0;
    const size_t b = DevPoolRegistry::get().budget_of(ctx->pool.device);
    return b > (size_t) 1 << 62 ? 0 : (int64_t) b;
}
uint64_t mrp_context_oom_events(mrp_context *ctx) { /* device allocations refused for good to this context and to its siblings (the concurrent batches of its calls) so far */
    uint64_t n = ctx->pool.oom_local.load();
    std::lock_guard<std::mutex> lock(ctx->sibling_mu);
    for (mrp_context *s_ : ctx->siblings) n += s_->pool.oom_local.load();
    return n;
}
void mrp_context_pool_bytes(mrp_context *ctx, int64_t *cached, int64_t *device_held) {
    { std::lock_guard<std::mutex> lock(ctx->pool.mu); *cached = (int64_t) ctx->pool.cached_bytes; }
    *device_held = ctx->pool.device >= 0 ? (int64_t) DevPoolRegistry::get().held[ctx->pool.device].load() : 0;
}
mrp_context *mrp_context_sibling(mrp_context *ctx, int i) {
    std::lock_guard<std::mutex> lock(ctx->sibling_mu);
    while ((int) ctx->siblings.size() <= i) {
        mrp_context *s = nullptr;
        if (mrp_context_create(ctx->device, &s) != MRP_OK) return nullptr;
        s->phase_groups = 1;
        s->grouped = true;
        s->test_hooks = ctx->test_hooks;
        ctx->siblings.push_back(s);
    }
    return ctx->siblings[(size_t) i];
}
int mrp_set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

const char *mrp_last_error(void) { return g_err; }
const char *mrp_version(void) { return "margin_rphmm 0.5.0 gfx950"; }
int mrp_abi_version(void) { return MRP_ABI_VERSION; }

/* Concurrent batches of mrp_phase_reads_many launch on 4 streams each; with the ROCm runtime's default of 4 hardware queues
 * their kernels would serialize (include/margin_rphmm.h).  An explicit call, to be made before the process's first HIP call:
 * a library that edits its host's environment when it is loaded is a surprise for whoever links it. */
int mrp_runtime_init(void) {
    if (setenv("GPU_MAX_HW_QUEUES", "16", 0) != 0) return fail(MRP_ERR_ARG, "mrp_runtime_init: setenv failed");
    return MRP_OK;
}

int mrp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mrp_context_create(int device, mrp_context **out) {
    if (!out) return fail(MRP_ERR_ARG, "mrp_context_create: out is NULL");
    *out = nullptr;
    int n = mrp_device_count();
    if (n <= 0) return fail(MRP_ERR_NO_DEVICE, "no HIP device visible: the stRPHmm sweep has no CPU fallback");
    if (device < 0 || device >= n) return fail(MRP_ERR_ARG, "device %d out of range (0..%d)", device, n - 1);
    HIP_TRY(hipSetDevice(device));
    mrp_context *ctx = new (std::nothrow) mrp_context();
    if (!ctx) return fail(MRP_ERR_NOMEM, "out of host memory");
    ctx->device = device;
    ctx->pool.attach(device); /* (the registry of the device's pools: budget, out-of-memory retry) */
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    for (int i = 0; i < 4 && e == hipSuccess; i++) e = hipEventCreate(&ctx->ev[i]);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->fork, hipEventDisableTiming);
    for (int i = 0; i < 2 && e == hipSuccess; i++) e = hipEventCreateWithFlags(&ctx->join[i], hipEventDisableTiming);
    if (e != hipSuccess) {
        mrp_context_destroy(ctx);
        return fail(MRP_ERR_HIP, "context setup failed: %s", hipGetErrorString(e));
    }
    *out = ctx;
    return MRP_OK;
}

void mrp_context_destroy(mrp_context *ctx) {
    if (!ctx) return;
    for (mrp_context *s : ctx->siblings) mrp_context_destroy(s);
    ctx->siblings.clear();
    (void) hipSetDevice(ctx->device);
    mrp_engine_release_context_cache(ctx);
    if (ctx->spare_batch) { mrp_batch_destroy(ctx->spare_batch); ctx->spare_batch = nullptr; }
    (void) hipDeviceSynchronize(); /* the auxiliary and copy streams too */
    ctx->pool.destroy();
    if (ctx->pinned) (void) hipHostFree(ctx->pinned);
    for (auto &e : ctx->ev)
        if (e) (void) hipEventDestroy(e);
    if (ctx->fork) (void) hipEventDestroy(ctx->fork);
    if (ctx->block_ev) (void) hipEventDestroy(ctx->block_ev);
    for (auto &e : ctx->join)
        if (e) (void) hipEventDestroy(e);
    for (auto &st : ctx->aux)
        if (st) (void) hipStreamDestroy(st);
    if (ctx->pre) (void) hipStreamDestroy(ctx->pre);
    if (ctx->stream) (void) hipStreamDestroy(ctx->stream);
    delete ctx;
}

/* the device memory the context (and the contexts it owns) keeps cached for its next call goes back to the driver */
int mrp_context_trim(mrp_context *ctx) {
    if (!ctx) return fail(MRP_ERR_ARG, "context is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lock(ctx->sibling_mu);
    for (mrp_context *s_ : ctx->siblings) { s_->pool.reclaim(); s_->pool.trim(); }
    ctx->pool.reclaim();
    ctx->pool.trim();
    DevPoolRegistry::get().forget_budget(ctx->pool.device); /* (asked for again at the next allocation: a budget shrunk under another tenant's pressure recovers) */
    return MRP_OK;
}

int mrp_context_synchronize(mrp_context *ctx) {
    if (!ctx) return fail(MRP_ERR_ARG, "context is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return MRP_OK;
}

}  /* extern "C" */

/* host half of a chunk: validation, prefix sums, host copies of everything (the caller's arrays may go after the call) */
static int chunk_host_init(mrp_context *ctx, int64_t n_sites, const uint32_t *allele_number, const uint16_t *substitution_log_probs,
                           const uint16_t *allele_prior_log_probs, const uint8_t *profile_pool, int64_t pool_bytes, mrp_chunk **out,
                           bool copy_pool = true) {
    if (!ctx || !out || n_sites < 0 || pool_bytes < 0 || (n_sites > 0 && !allele_number) ||
        (pool_bytes > 0 && !profile_pool))
        return fail(MRP_ERR_ARG, "mrp_chunk_create: bad arguments");
    *out = nullptr;
    mrp_chunk *ch = new (std::nothrow) mrp_chunk();
    if (!ch) return fail(MRP_ERR_NOMEM, "out of host memory");
    ch->ctx = ctx;
    ch->n_sites = n_sites;
    ch->pool_bytes = pool_bytes;
    ch->allele_number.assign(allele_number, allele_number + n_sites);
    ch->allele_offset.resize(n_sites + 1);
    ch->sub_offset.resize(n_sites + 1);
    uint64_t off = 0, soff = 0;
    for (int64_t i = 0; i < n_sites; i++) {
        ch->allele_offset[i] = (uint32_t) off;
        ch->sub_offset[i] = (uint32_t) soff;
        uint64_t A = allele_number[i];
        if (A == 0 || A > 65535) {
            delete ch;
            return fail(MRP_ERR_ARG, "site %lld has %llu alleles", (long long) i, (unsigned long long) A);
        }
        off += A;
        soff += A * A;
        ch->max_alleles = std::max<uint32_t>(ch->max_alleles, (uint32_t) A);
        if (off > 0xFFFFFFFFull || soff > 0xFFFFFFFFull) {
            delete ch;
            return fail(MRP_ERR_ARG, "allele tables exceed 32-bit offsets");
        }
    }
    ch->allele_offset[n_sites] = (uint32_t) off;
    ch->same_until.resize((size_t) n_sites);
    for (int64_t i = n_sites - 1; i >= 0; i--)
        ch->same_until[(size_t) i] = (i + 1 < n_sites && allele_number[i + 1] == allele_number[i]) ? ch->same_until[(size_t) i + 1] : (int32_t) (i + 1);
    ch->sub_offset[n_sites] = (uint32_t) soff;
    std::vector<uint16_t> &sub = ch->sub, &prior = ch->prior;
    sub.assign(soff, 0);
    prior.assign(off, 0);
    if (pool_bytes > 0 && copy_pool) ch->pool.assign(profile_pool, profile_pool + pool_bytes);
    ch->pool_host = ch->pool.data(); /* (copy_pool = false: set by the caller, who keeps the bytes elsewhere) */
    if (substitution_log_probs) sub.assign(substitution_log_probs, substitution_log_probs + soff);
    if (allele_prior_log_probs) prior.assign(allele_prior_log_probs, allele_prior_log_probs + off);
    for (uint16_t v : sub) ch->max_sub = std::max<uint32_t>(ch->max_sub, v);
    for (uint16_t v : prior) ch->max_prior = std::max<uint32_t>(ch->max_prior, v);
    *out = ch;
    return MRP_OK;
}

extern "C" {
int mrp_chunk_create(mrp_context *ctx, int64_t n_sites, const uint32_t *allele_number,
                     const uint16_t *substitution_log_probs, const uint16_t *allele_prior_log_probs,
                     const uint8_t *profile_pool, int64_t pool_bytes, mrp_chunk **out) {
    mrp_chunk *ch = nullptr;
    int rc = chunk_host_init(ctx, n_sites, allele_number, substitution_log_probs, allele_prior_log_probs, profile_pool, pool_bytes, &ch);
    if (rc != MRP_OK) return rc;
    *out = nullptr;
    hipError_t e = hipSetDevice(ctx->device);
    hipStream_t s = ctx->stream;
    /* from the context's caching allocator */
    ch->d_allele_number.pool = ch->d_allele_offset.pool = ch->d_sub_offset.pool = &ctx->pool;
    ch->d_sub.pool = ch->d_prior.pool = &ctx->pool;
    ch->d_same_until.pool = &ctx->pool;
    ch->d_pool.pool = &ctx->pool;
    if (e == hipSuccess) e = ch->d_allele_number.upload(ch->allele_number, s);
    if (e == hipSuccess) e = ch->d_allele_offset.upload(ch->allele_offset, s);
    if (e == hipSuccess) e = ch->d_sub_offset.upload(ch->sub_offset, s);
    if (e == hipSuccess) e = ch->d_same_until.upload(ch->same_until, s);
    if (e == hipSuccess) e = ch->d_sub.upload(ch->sub, s);
    if (e == hipSuccess) e = ch->d_prior.upload(ch->prior, s);
    if (e == hipSuccess) e = ch->d_pool.alloc((size_t) pool_bytes + 16); /* (the packing kernel reads a read's last bytes a dword at a time) */
    if (e == hipSuccess && pool_bytes > 0)
        e = hipMemcpyAsync(ch->d_pool.p, ch->pool.data(), (size_t) pool_bytes, hipMemcpyHostToDevice, s); /* (the chunk's own copy: the caller's may go) */
    if (e == hipSuccess) e = ctx->wait_stream(s);
    if (e != hipSuccess) {
        delete ch;
        return fail(MRP_ERR_HIP, "chunk upload failed: %s", hipGetErrorString(e));
    }
    ch->dev.allele_number = ch->d_allele_number.p;
    ch->dev.allele_offset = ch->d_allele_offset.p;
    ch->dev.sub_offset = ch->d_sub_offset.p;
    ch->dev.sub = ch->d_sub.p;
    ch->dev.prior = ch->d_prior.p;
    ch->dev.pool = ch->d_pool.p;
    ch->dev.same_until = ch->d_same_until.p;
    *out = ch;
    return MRP_OK;
}
}  /* extern "C" */

/* The chunks of one batch of a work queue, uploaded TOGETHER: every array of every chunk is copied (by the calling thread's
 * host pool) into one page-locked block, which goes to one device block with one asynchronous copy on the context's stream;
 * one event ends it.  Nothing is waited for here: the first device work that reads a chunk waits for the event on its stream
 * (mrp_engine.cpp) or on the host (mrp_chunk::host_wait).  Per chunk this replaces seven allocations and seven pageable
 * copies (30 ms for 288 chunks) by a share of one. */
int mrp_chunk_block_create(mrp_context *ctx, int64_t n, const mrp_chunk_desc *const *descs, mrp_chunk **out, mrp_chunk_block *blk, int groups) {
    if (!ctx || n < 0 || !blk || (n > 0 && (!descs || !out))) return fail(MRP_ERR_ARG, "mrp_chunk_block_create: bad arguments");
    for (int64_t i = 0; i < n; i++) out[i] = nullptr;
    HIP_TRY(hipSetDevice(ctx->device));
    std::vector<int> rcs((size_t) n, MRP_OK);
    std::vector<std::string> msgs((size_t) n);
    mrp_parallel_for(n, 4, [&](int64_t i) {
        const mrp_chunk_desc &c = *descs[i];
        /* (the profile bytes -- nine tenths of a chunk -- are copied ONCE, into the page-locked block below, which outlives the chunk) */
        rcs[(size_t) i] = chunk_host_init(ctx, c.n_sites, c.allele_number, c.substitution_log_probs, c.allele_prior_log_probs, c.profile_pool, c.pool_bytes, &out[i], false);
        if (rcs[(size_t) i] != MRP_OK) msgs[(size_t) i] = mrp_last_error();
    });
    int rc = MRP_OK;
    for (int64_t i = 0; i < n && rc == MRP_OK; i++)
        if (rcs[(size_t) i] != MRP_OK) rc = fail(rcs[(size_t) i], "%s", msgs[(size_t) i].c_str());
    auto al = [](size_t v) { return (v + 255) & ~(size_t) 255; };
    if (groups < 1 || n < 4 * (int64_t) groups) groups = 1;
    if (groups > 16) groups = 16;
    /* the order of the chunks in the block: group 0's, then group 1's, ... -- the groups are the concurrent batches mrp_phase_reads_many
     * will deal the chunks to (mrp_phase_group_assign: not always i % groups) */
    std::vector<uint8_t> group_of((size_t) n + 1, 0);
    {
        int64_t sites = 0;
        for (int64_t i = 0; i < n; i++) sites += descs[i]->n_sites;
        mrp_phase_group_assign(n, groups, sites, group_of.data());
    }
    std::vector<int64_t> order; order.reserve((size_t) n);
    std::vector<int64_t> group_first((size_t) groups + 1, 0);
    for (int g = 0; g < groups; g++) { group_first[(size_t) g] = (int64_t) order.size(); for (int64_t i = 0; i < n; i++) if (group_of[(size_t) i] == g) order.push_back(i); }
    group_first[(size_t) groups] = n;
    std::vector<size_t> off((size_t) n + 1, 0), at_of((size_t) n, 0); /* off: by position in the block; at_of: by chunk */
    for (int64_t k = 0; k < n && rc == MRP_OK; k++) {
        const mrp_chunk *ch = out[order[(size_t) k]];
        at_of[(size_t) order[(size_t) k]] = off[(size_t) k];
        off[(size_t) k + 1] = off[(size_t) k] + al(4 * ch->allele_number.size()) + al(4 * ch->allele_offset.size()) + al(4 * ch->sub_offset.size()) +
                              al(4 * ch->same_until.size()) + al(2 * ch->sub.size()) + al(2 * ch->prior.size()) + al((size_t) ch->pool_bytes);
    }
    hipError_t e = hipSuccess;
    if (rc == MRP_OK) {
        blk->dev.pool = &ctx->pool;
        e = blk->host.reserve(off[(size_t) n] + 256);
        if (e == hipSuccess) e = blk->dev.alloc(off[(size_t) n] + 256);
        if (e == hipSuccess && !blk->ready) e = hipEventCreateWithFlags(&blk->ready, hipEventBlockingSync | hipEventDisableTiming);
        while (e == hipSuccess && (int) blk->group_ready.size() < groups) {
            hipEvent_t ev = nullptr;
            e = hipEventCreateWithFlags(&ev, hipEventBlockingSync | hipEventDisableTiming);
            if (e == hipSuccess) blk->group_ready.push_back(ev);
        }
    }
    if (rc == MRP_OK && e == hipSuccess) {
        char *hb = (char *) blk->host.p;
        uint8_t *db = blk->dev.p;
        for (int g = 0; g < groups && e == hipSuccess; g++) { /* group by group: the copy of one runs beside the staging of the next */
            const int64_t g_n = group_first[(size_t) g + 1] - group_first[(size_t) g];
            mrp_parallel_for(g_n, 4, [&](int64_t k) {
                const int64_t i = order[(size_t) (group_first[(size_t) g] + k)];
                mrp_chunk *ch = out[i];
                size_t o = at_of[(size_t) i];
                auto put = [&](const void *src, size_t bytes) { const size_t at = o; if (bytes) memcpy(hb + at, src, bytes); o += al(bytes); return db + at; };
                ch->dev.allele_number = (const uint32_t *) put(ch->allele_number.data(), 4 * ch->allele_number.size());
                ch->dev.allele_offset = (const uint32_t *) put(ch->allele_offset.data(), 4 * ch->allele_offset.size());
                ch->dev.sub_offset = (const uint32_t *) put(ch->sub_offset.data(), 4 * ch->sub_offset.size());
                ch->dev.same_until = (const int32_t *) put(ch->same_until.data(), 4 * ch->same_until.size());
                ch->dev.sub = (const uint16_t *) put(ch->sub.data(), 2 * ch->sub.size());
                ch->dev.prior = (const uint16_t *) put(ch->prior.data(), 2 * ch->prior.size());
                ch->pool_host = (const uint8_t *) (hb + o);
                ch->dev.pool = (const uint8_t *) put(descs[i]->profile_pool, (size_t) ch->pool_bytes);
            });
            const size_t lo = off[(size_t) group_first[(size_t) g]], hi = off[(size_t) group_first[(size_t) g + 1]];
            if (hi > lo) e = hipMemcpyAsync(db + lo, hb + lo, hi - lo, hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) e = hipEventRecord(blk->group_ready[(size_t) g], ctx->stream);
        }
        if (e == hipSuccess) e = hipEventRecord(blk->ready, ctx->stream);
        if (e == hipSuccess && getenv("MRP_TIMING_UPLOAD")) { /* diagnosis only: waits for the copy */
            const auto t0 = std::chrono::steady_clock::now();
            e = hipEventSynchronize(blk->ready);
            fprintf(stderr, "  chunk block: %lld chunks, %.1f MB, copy waited %.2f ms\n", (long long) n, (double) off[(size_t) n] / 1e6,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }
        if (e == hipSuccess)
            for (int64_t i = 0; i < n; i++) { out[i]->ready = blk->group_ready[(size_t) group_of[(size_t) i]]; out[i]->owns_ready = false; out[i]->ready_pending.store(true); }
    }
    if (rc == MRP_OK && e != hipSuccess) rc = fail(MRP_ERR_HIP, "chunk block upload failed: %s", hipGetErrorString(e));
    if (rc != MRP_OK)
        for (int64_t i = 0; i < n; i++) { delete out[i]; out[i] = nullptr; }
    return rc;
}

extern "C" {

void mrp_chunk_destroy(mrp_chunk *chunk) {
    if (!chunk) return;
    (void) hipSetDevice(chunk->ctx->device);
    delete chunk;
}

int mrp_batch_create(mrp_context *ctx, mrp_batch **out) {
    if (!ctx || !out) return fail(MRP_ERR_ARG, "mrp_batch_create: bad arguments");
    mrp_batch *b = new (std::nothrow) mrp_batch();
    if (!b) return fail(MRP_ERR_NOMEM, "out of host memory");
    b->ctx = ctx;
    b->bind_pool(&ctx->pool);
    *out = b;
    return MRP_OK;
}

void mrp_batch_destroy(mrp_batch *batch) {
    if (!batch) return;
    (void) hipSetDevice(batch->ctx->device);
    mrp_context *ctx = batch->ctx;
    (void) hipStreamSynchronize(ctx->stream); /* the auxiliary streams were joined into it */
    for (auto &slot : batch->ev_ring)
        for (hipEvent_t ev : slot) {
            if (ev == ctx->last_emission) ctx->last_emission = nullptr;
            if (ev) (void) hipEventDestroy(ev);
        }
    delete batch;
    ctx->pool.reclaim();
}

/* resolve cell -> merge cell indices from keys: stHash_search of mergeColumn.c:63-79 */
static int resolve_column(const uint64_t *part, int64_t n_cells, uint64_t mask, const uint64_t *keys,
                          int64_t n_keys, uint32_t *out) {
    size_t cap = 16;
    while (cap < (size_t) n_keys * 2) cap *= 2;
    std::vector<uint64_t> hk(cap);
    std::vector<uint32_t> hv(cap, 0xFFFFFFFFu);
    auto mix = [](uint64_t x) {
        x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
        return x;
    };
    for (int64_t i = 0; i < n_keys; i++) {
        size_t s = mix(keys[i]) & (cap - 1);
        while (hv[s] != 0xFFFFFFFFu) {
            if (hk[s] == keys[i]) return MRP_ERR_ARG; /* duplicate key: mergeColumn.c:104-107 asserts */
            s = (s + 1) & (cap - 1);
        }
        hk[s] = keys[i];
        hv[s] = (uint32_t) i;
    }
    for (int64_t c = 0; c < n_cells; c++) {
        const uint64_t key = part[c] & mask;
        size_t s = mix(key) & (cap - 1);
        uint32_t found = 0xFFFFFFFFu;
        while (hv[s] != 0xFFFFFFFFu) {
            if (hk[s] == key) { found = hv[s]; break; }
            s = (s + 1) & (cap - 1);
        }
        if (found == 0xFFFFFFFFu) return MRP_ERR_LOOKUP;
        out[c] = found;
    }
    return MRP_OK;
}

}  /* extern "C" */

int mrp_batch_add_impl(mrp_batch *b, const mrp_hmm_job *job, bool resident, int64_t *cell0_out, int64_t *mcell0_out,
                       int64_t *col0_out) {
    if (!b || !job) return fail(MRP_ERR_ARG, "mrp_batch_add: NULL argument");
    if (b->uploaded) return fail(MRP_ERR_ARG, "mrp_batch_add: batch already uploaded");
    const int K = job->n_columns;
    if (K < 1) return fail(MRP_ERR_ARG, "hmm has %d columns", K);
    if (!job->chunk || !job->col_ref_start || !job->col_length || !job->col_depth || !job->col_cell_off ||
        !job->col_read_off || (!resident && !job->partition))
        return fail(MRP_ERR_ARG, "hmm job is missing required arrays");
    if (K > 1 && (!job->mcol_cell_off || (!resident && (!job->cell_next || !job->cell_prev) &&
                                          (!job->mask_from || !job->mask_to || !job->merge_from || !job->merge_to))))
        return fail(MRP_ERR_ARG, "hmm job is missing its merge column arrays");
    const bool device_only = !job->cell_forward && !job->cell_backward && !job->col_total && !job->hmm_forward &&
                             !job->hmm_backward && !job->merge_forward && !job->merge_backward;
    if (!device_only && (!job->cell_forward || !job->cell_backward || !job->col_total || !job->hmm_forward ||
                         !job->hmm_backward || (K > 1 && (!job->merge_forward || !job->merge_backward))))
        return fail(MRP_ERR_ARG, "hmm job is missing output arrays");
    const mrp_chunk *ch = job->chunk;
    if (ch->host_wait() != hipSuccess) return fail(MRP_ERR_HIP, "chunk upload failed");
    if (ch->ctx->device != b->ctx->device) return fail(MRP_ERR_ARG, "chunk lives on a different device");
    std::lock_guard<std::mutex> lock(b->mu);
    if (b->stats.n_hmms > 0 && b->resident != resident) return fail(MRP_ERR_ARG, "a batch is either host-fed or device-resident");
    b->resident = resident;
    const bool ancestor = (job->flags & MRP_FLAG_INCLUDE_ANCESTOR_SUB_PROB) != 0;

    int chunk_index = -1;
    for (size_t i = 0; i < b->chunks.size(); i++)
        if (b->chunks[i] == ch) chunk_index = (int) i;
    if (chunk_index < 0) {
        chunk_index = (int) b->chunks.size();
        b->chunks.push_back(ch);
    }

    /* validate before touching the batch */
    if (job->col_cell_off[0] != 0 || job->col_read_off[0] != 0 || (K > 1 && job->mcol_cell_off[0] != 0))
        return fail(MRP_ERR_ARG, "prefix-sum arrays must start at 0");
    for (int k = 0; k < K; k++) {
        const int64_t nc = job->col_cell_off[k + 1] - job->col_cell_off[k];
        const int64_t nd = job->col_read_off[k + 1] - job->col_read_off[k];
        if (nc < 1 || nc > 0x7FFFFFFF) return fail(MRP_ERR_ARG, "column %d has %lld cells", k, (long long) nc);
        if (job->col_depth[k] < 0 || job->col_depth[k] > MRP_MAX_READ_PARTITIONING_DEPTH || nd != job->col_depth[k])
            return fail(MRP_ERR_ARG, "column %d: depth %d inconsistent", k, job->col_depth[k]);
        if (job->col_length[k] < 1 || job->col_ref_start[k] < 0 ||
            (int64_t) job->col_ref_start[k] + job->col_length[k] > ch->n_sites)
            return fail(MRP_ERR_ARG, "column %d: site interval [%d,+%d) outside the reference", k,
                        job->col_ref_start[k], job->col_length[k]);
        if (nd > 0 && !job->read_byte_off) return fail(MRP_ERR_ARG, "read_byte_off is NULL");
        const uint32_t slots = ch->allele_offset[job->col_ref_start[k] + job->col_length[k]] -
                               ch->allele_offset[job->col_ref_start[k]];
        for (int64_t i = 0; i < nd; i++) {
            const int64_t o = job->read_byte_off[job->col_read_off[k] + i];
            if (o < 0 || o + (int64_t) slots > ch->pool_bytes)
                return fail(MRP_ERR_ARG, "column %d read %lld: profile bytes outside the pool", k, (long long) i);
        }
        if (ancestor) {
            for (int s = 0; s < job->col_length[k]; s++)
                if (ch->allele_number[job->col_ref_start[k] + s] > MRP_MAX_ALLELES)
                    return fail(MRP_ERR_UNSUPPORTED, "site %d has more than %d alleles (ancestor mode)",
                                job->col_ref_start[k] + s, MRP_MAX_ALLELES);
        }
        if (k + 1 < K) {
            const int64_t nm = job->mcol_cell_off[k + 1] - job->mcol_cell_off[k];
            if (nm < 1 || nm > 0x7FFFFFFF) return fail(MRP_ERR_ARG, "merge column %d has %lld cells", k, (long long) nm);
        }
    }

    const int64_t n_cells = job->col_cell_off[K];
    const int64_t n_merge = K > 1 ? job->mcol_cell_off[K - 1] : 0;
    /* every hmm starts at a multiple of 4 cells: the recursion kernel moves 4 cells per lane (16 B) */
    while (b->n_cells_total % 4 != 0) {
        b->n_cells_total++;
        if (!resident) {
            b->partition.push_back(0);
            if (b->need_wide) { b->cell_next.push_back(0); b->cell_prev.push_back(0); }
            b->cell_np.push_back(0);
        }
    }
    const int64_t cell0 = b->n_cells_total;
    const int64_t mcell0 = b->n_merge;
    const int64_t col0 = (int64_t) b->cols.size();
    const int64_t read0 = (int64_t) b->read_byte_off.size();

    std::vector<uint32_t> nxt((size_t) (resident ? 0 : n_cells), 0), prv((size_t) (resident ? 0 : n_cells), 0);
    for (int k = 0; k < K && !resident; k++) {
        const int64_t c0 = job->col_cell_off[k], nc = job->col_cell_off[k + 1] - c0;
        if (k + 1 < K) {
            const int64_t m0 = job->mcol_cell_off[k], nm = job->mcol_cell_off[k + 1] - m0;
            if (job->cell_next) {
                for (int64_t c = 0; c < nc; c++) {
                    if (job->cell_next[c0 + c] >= (uint64_t) nm) return fail(MRP_ERR_ARG, "cell_next out of range");
                    nxt[c0 + c] = job->cell_next[c0 + c];
                }
            } else {
                int rc = resolve_column(job->partition + c0, nc, job->mask_from[k], job->merge_from + m0, nm, &nxt[c0]);
                if (rc != MRP_OK) return fail(rc, "column %d: a cell has no next merge cell (mergeColumn.c:63)", k);
            }
        }
        if (k > 0) {
            const int64_t m0 = job->mcol_cell_off[k - 1], nm = job->mcol_cell_off[k] - m0;
            if (job->cell_prev) {
                for (int64_t c = 0; c < nc; c++) {
                    if (job->cell_prev[c0 + c] >= (uint64_t) nm) return fail(MRP_ERR_ARG, "cell_prev out of range");
                    prv[c0 + c] = job->cell_prev[c0 + c];
                }
            } else {
                int rc = resolve_column(job->partition + c0, nc, job->mask_to[k - 1], job->merge_to + m0, nm, &prv[c0]);
                if (rc != MRP_OK) return fail(rc, "column %d: a cell has no previous merge cell (mergeColumn.c:72)", k);
            }
        }
    }

    DevHmm h{};
    h.col0 = col0;
    h.n_cols = K;
    h.flags = job->flags;
    h.max_merge = 1;
    h.max_cells = 1;
    h.cost_bound = 0;
    for (int k = 0; k < K; k++) {
        DevCol c{};
        c.cell_off = cell0 + job->col_cell_off[k];
        c.n_cells = (int32_t) (job->col_cell_off[k + 1] - job->col_cell_off[k]);
        c.mcell_off = k + 1 < K ? mcell0 + job->mcol_cell_off[k] : 0;
        c.n_merge = k + 1 < K ? (int32_t) (job->mcol_cell_off[k + 1] - job->mcol_cell_off[k]) : 0;
        c.slot_off = b->n_slots;
        c.read_off = read0 + job->col_read_off[k];
        c.site_start = job->col_ref_start[k];
        c.n_sites = job->col_length[k];
        c.depth = job->col_depth[k];
        c.n_slots = (int32_t) (ch->allele_offset[c.site_start + c.n_sites] - ch->allele_offset[c.site_start]);
        c.chunk = chunk_index;
        c.flags = job->flags;
        b->n_slots += c.n_slots;
        int32_t uniform = (int32_t) ch->allele_number[c.site_start];
        for (int s2 = 1; s2 < c.n_sites; s2++)
            if ((int32_t) ch->allele_number[c.site_start + s2] != uniform) uniform = 0;
        for (int t0 = 0; t0 < c.n_cells; t0 += MRP_EMIT_TILE) {
            EmitTile t{};
            t.cell_off = c.cell_off + t0;
            t.slot_off = c.slot_off;
            t.n = std::min<int32_t>(MRP_EMIT_TILE, c.n_cells - t0);
            t.col = (int32_t) b->cols.size();
            t.n_sites = c.n_sites;
            t.uniform_alleles = uniform;
            t.depth = c.depth;
            t.flags = job->flags;
            b->tiles.push_back(t);
        }
        SweepCol sc{};
        sc.cell_off = c.cell_off;
        sc.mcell_off = c.mcell_off;
        sc.n_cells = c.n_cells;
        sc.n_merge = c.n_merge;
        b->scols.push_back(sc);
        PlaneCol pc{};
        pc.pool = ch->dev.pool;
        pc.read_off = c.read_off;
        pc.slot_off = c.slot_off;
        pc.depth = c.depth;
        pc.n_slots = c.n_slots;
        pc.need_planes = (uniform == 0 || ancestor) ? 1 : 0;
        b->pcols.push_back(pc);
        b->cols.push_back(c);
        h.max_merge = std::max(h.max_merge, c.n_merge);
        h.max_cells = std::max(h.max_cells, c.n_cells);
        int64_t per_site = 255ll * c.depth;
        if (ancestor) per_site += 2ll * ch->max_sub + ch->max_prior;
        h.cost_bound += per_site * c.n_sites;
        /* statistics: SURVEY.md 8(d) algorithmic bytes and the CPU formulation's popcount count */
        b->stats.profile_bytes += (int64_t) c.depth * c.n_slots;
        b->stats.algorithmic_bytes += 24ll * c.n_cells + 32ll * c.n_merge + (int64_t) c.depth * c.n_slots + 8;
        b->stats.popcount_ops += (int64_t) c.n_cells * 2 * c.n_slots * 8;
    }
    h.n_cells = n_cells;
    h.n_merge = n_merge;
    h.wide_idx = h.max_merge > 65535 ? 1 : 0;
    if (h.wide_idx && !b->need_wide) {
        /* the first hmm whose transitions do not fit 16 bits: from now on the full indices are kept too; those of the
         * hmms added so far are recovered from their packed form (they fit) */
        b->need_wide = true;
        if (!resident) {
            const size_t have = b->cell_np.size();
            b->cell_next.resize(have);
            b->cell_prev.resize(have);
            for (size_t c = 0; c < have; c++) { b->cell_next[c] = b->cell_np[c] & 0xFFFFu; b->cell_prev[c] = b->cell_np[c] >> 16; }
        }
    }
    b->hmms.push_back(h);
    b->n_cells_total += n_cells;
    if (resident && h.wide_idx) return fail(MRP_ERR_UNSUPPORTED, "device-resident hmm with more than 65535 merge cells in a column");
    if (!resident) {
        b->partition.insert(b->partition.end(), job->partition, job->partition + n_cells);
        if (b->need_wide) {
            b->cell_next.insert(b->cell_next.end(), nxt.begin(), nxt.end());
            b->cell_prev.insert(b->cell_prev.end(), prv.begin(), prv.end());
        }
        const size_t base = b->cell_np.size();
        b->cell_np.resize(base + (size_t) n_cells);
        for (int64_t c = 0; c < n_cells; c++) b->cell_np[base + c] = (nxt[c] & 0xFFFFu) | (prv[c] << 16);
    }
    if (job->col_read_off[K] > 0)
        b->read_byte_off.insert(b->read_byte_off.end(), job->read_byte_off, job->read_byte_off + job->col_read_off[K]);
    b->n_merge += n_merge;
    JobOut o{job->cell_forward, job->cell_backward, job->merge_forward, job->merge_backward, job->col_total,
             job->hmm_forward, job->hmm_backward, cell0, n_cells, mcell0, n_merge, col0, K};
    b->outs.push_back(o);
    b->stats.n_hmms += 1;
    b->stats.n_columns += K;
    b->stats.n_cells += n_cells;
    b->stats.n_merge_cells += n_merge;
    if (cell0_out) *cell0_out = cell0;
    if (mcell0_out) *mcell0_out = mcell0;
    if (col0_out) *col0_out = col0;
    return MRP_OK;
}

/* ---- persistent host worker pool ----------------------------------------------------------------------------
 * mrp_pool_run(n, grain, fn, arg) calls fn(i, arg) for every i in [0, n) on the calling thread and the pool's workers and
 * returns when all are done.  The resident pipeline issues ~50 short parallel loops per call; creating and joining 15
 * threads for each of them cost more than many of the loops.  Several callers may be inside at once (the concurrent
 * batches of mrp_phase_reads_many): jobs queue up, a worker serves the job of the most urgent caller that still has indices
 * to hand out (mrp_pool_set_priority: batch 0 before batch 1 ...: the batches then leave their host-only phases one after
 * the other instead of all together, and the device has work while the later ones are still being prepared), the oldest
 * among equals. */
std::atomic<long long> g_pool_task_cpu_ns{0};
std::atomic<long long> g_pool_tag_cpu_ns[16];
static thread_local long long t_pool_task_cpu_ns = 0; /* pool tasks executed by the calling thread itself */
extern "C" long long mrp_pool_task_cpu_ns(void) { return g_pool_task_cpu_ns.load(); }
extern "C" long long mrp_pool_task_cpu_ns_this_thread(void) { return t_pool_task_cpu_ns; }
extern "C" long long mrp_pool_tag_cpu_ns(int tag) { return g_pool_tag_cpu_ns[tag & 15].load(); }
namespace {
thread_local int t_pool_priority = 0;
thread_local int t_pool_tag = 0;
thread_local int t_pool_weight_ns = 0; /* mrp_pool_set_weight: what one index of the calling thread's next loops costs, roughly; 0: unknown */
/* The indices of a loop are dealt out from MRP_POOL_RANGES contiguous ranges, and a thread starts with the range of its own
 * number before it helps with the others: the loops of a batch's levels run over the same chunks in the same order, so the
 * thread that built a chunk's hmms at one level mostly meets them again at the next (their blocks are in its cache, or its
 * neighbours') instead of wherever a single shared counter sends it. */
#define MRP_POOL_RANGES 64 /* at most; in use: pool_ranges() */
static inline int pool_ranges() { const char *e = getenv("MRP_POOL_RANGES"); const int v = e ? atoi(e) : 16; return v < 1 ? 1 : (v > MRP_POOL_RANGES ? MRP_POOL_RANGES : v); } /* development knob */
struct alignas(64) PoolRange { std::atomic<int64_t> next{0}; int64_t end = 0; };
struct PoolJob {
    void (*fn)(int64_t, void *);
    void *arg;
    int64_t n, grain;
    int prio = 0, tag = 0;
    PoolRange range[MRP_POOL_RANGES];
    std::atomic<int64_t> done{0};
    std::atomic<int> exhausted{0}; /* ranges that have nothing left to hand out */
    int n_ranges = 16;
    int active = 0; /* workers currently holding the pointer (under Pool::mu) */
    std::condition_variable cv; /* the posting thread waits here: woken by the last worker to let go of the job, not by every worker of every job */
    bool has_work() const { return exhausted.load(std::memory_order_relaxed) < n_ranges; }
};
thread_local int t_pool_slot = -1; /* the calling thread's number in its pool: workers 0 .. threads - 2, a posting thread threads - 1 */
}  // namespace
/* One pool serves the process by default (mrp_set_host_threads); a work queue gives every device its own (mrp_queue.cpp:
 * the reference's axis is "every core works", phase.c:276-279 -- eight devices on one shared pool of sixteen threads would
 * starve each other), optionally bound to the CPUs next to the device.  A thread posts its loops to the pool it has adopted
 * (mrp_pool_adopt; the batch threads of mrp_phase_reads_many inherit their caller's).
 * Wake-ups are counted: a loop wakes as many sleeping workers as it has grains to give away (a call posts some four hundred loops,
 * half of them over a few hundred indices: waking every worker for each of them, and every posting thread whenever any worker
 * finished, was a seventh of the call's host CPU time in futex calls and on the pool's mutex). */
struct mrp_host_pool {
    std::mutex mu;
    std::condition_variable cv_work;
    std::vector<PoolJob *> jobs;
    std::vector<std::thread> workers;
    int idle = 0; /* workers asleep in cv_work (under mu) */
    bool stop = false;
    int fixed_threads = 0; /* 0: the process-wide pool, sized by mrp_host_threads() */
    static void run_chunks(PoolJob *j) {
        struct Acc { /* MRP_TIMING: thread CPU spent inside pool tasks */
            timespec a;
            int tag;
            Acc(int t) : tag(t) { clock_gettime(CLOCK_THREAD_CPUTIME_ID, &a); }
            ~Acc() { timespec b; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &b); const long long d = (b.tv_sec - a.tv_sec) * 1000000000ll + (b.tv_nsec - a.tv_nsec);
                     g_pool_task_cpu_ns.fetch_add(d); g_pool_tag_cpu_ns[tag & 15].fetch_add(d); t_pool_task_cpu_ns += d; }
        } acc(j->tag);
        const int nr = j->n_ranges, home = (t_pool_slot >= 0 ? t_pool_slot : 0) % nr;
        int64_t mine = 0; /* booked once: the counter is one cache line shared by every thread of the loop */
        for (int k = 0; k < nr; k++) {
            PoolRange &r = j->range[(home + k) % nr];
            for (;;) {
                if (r.next.load(std::memory_order_relaxed) >= r.end) break;
                const int64_t lo = r.next.fetch_add(j->grain);
                if (lo >= r.end) break;
                const int64_t hi = std::min(r.end, lo + j->grain);
                if (lo + j->grain >= r.end) j->exhausted.fetch_add(1); /* (took the range's last grain: exactly one thread does) */
                for (int64_t i = lo; i < hi; i++) j->fn(i, j->arg);
                mine += hi - lo;
            }
        }
        if (mine) j->done.fetch_add(mine);
    }
    void worker() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            PoolJob *j = nullptr;
            for (PoolJob *q : jobs)
                if (q->has_work() && (!j || q->prio < j->prio)) j = q;
            if (!j) {
                if (stop) return;
                idle++;
                cv_work.wait(lk);
                idle--;
                continue;
            }
            j->active++;
            lk.unlock();
            run_chunks(j);
            lk.lock();
            if (--j->active == 0 && j->done.load() >= j->n) j->cv.notify_one();
        }
    }
    void ensure(int n_workers) {
        std::lock_guard<std::mutex> lk(mu);
        while ((int) workers.size() < n_workers) { const int slot = (int) workers.size(); workers.emplace_back([this, slot] { t_pool_slot = slot; worker(); }); }
    }
    ~mrp_host_pool() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv_work.notify_all();
        for (auto &t : workers) t.join();
    }
};
namespace {
typedef mrp_host_pool Pool;
thread_local Pool *t_pool_current = nullptr;
Pool &pool() {
    static Pool *p = new Pool(); /* never destroyed: worker threads must not be joined from a static destructor at exit */
    return *p;
}
}  // namespace

extern "C" void mrp_pool_run(int64_t n, int64_t grain, void (*fn)(int64_t, void *), void *arg) {
    if (n <= 0) return;
    if (grain < 1) grain = 1;
    Pool &P = t_pool_current ? *t_pool_current : pool();
    const int threads = P.fixed_threads > 0 ? P.fixed_threads : mrp_host_threads();
    const int nt = (int) std::min<int64_t>(threads, (n + grain - 1) / grain);
    if (nt <= 1) {
        for (int64_t i = 0; i < n; i++) fn(i, arg);
        return;
    }
    /* a loop whose whole work is a few dozen microseconds is run here: posting it costs the poster and every woken worker a futex
     * call and a turn on the pool's mutex each (the levels of a call of small chunks post hundreds of such loops) */
    const int64_t est_ns = t_pool_weight_ns > 0 ? n * (int64_t) t_pool_weight_ns : -1;
    if (est_ns >= 0 && est_ns < 60000) {
        for (int64_t i = 0; i < n; i++) fn(i, arg);
        return;
    }
    P.ensure(threads - 1);
    PoolJob j;
    j.fn = fn; j.arg = arg; j.n = n; j.grain = grain; j.prio = t_pool_priority; j.tag = t_pool_tag;
    {   /* ranges of whole grains; the empty ones (a short loop) count as exhausted from the start */
        const int64_t grains = (n + grain - 1) / grain;
        int empty = 0;
        const int nr = j.n_ranges = pool_ranges();
        for (int r = 0; r < nr; r++) {
            const int64_t lo = std::min(n, grains * r / nr * grain), hi = std::min(n, grains * (r + 1) / nr * grain);
            j.range[r].next.store(lo); j.range[r].end = hi;
            if (lo >= hi) empty++;
        }
        j.exhausted.store(empty);
    }
    if (t_pool_slot < 0) t_pool_slot = threads - 1;
    int wake;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        P.jobs.push_back(&j);
        /* one sleeper per grain beyond the poster's own (a worker that is busy looks at the job list when it is done: nothing is lost) */
        wake = (int) std::min<int64_t>(P.idle, std::min<int64_t>(threads - 1, (n + grain - 1) / grain - 1));
        if (est_ns >= 0) wake = (int) std::min<int64_t>(wake, est_ns / 100000); /* ... that has some 100 us of work to find */
    }
    for (int k = 0; k < wake; k++) P.cv_work.notify_one();
    Pool::run_chunks(&j);
    std::unique_lock<std::mutex> lk(P.mu);
    j.cv.wait(lk, [&] { return j.done.load() >= j.n && j.active == 0; });
    P.jobs.erase(std::find(P.jobs.begin(), P.jobs.end(), &j));
}

/* a pool of its own with `threads` threads (the posting thread counts as one); its workers are created by the first loop
 * posted to it and inherit the CPU affinity of the thread that posts it */
mrp_host_pool *mrp_host_pool_create(int threads) {
    mrp_host_pool *p = new (std::nothrow) mrp_host_pool();
    if (p) p->fixed_threads = threads < 1 ? 1 : threads;
    return p;
}
void mrp_host_pool_destroy(mrp_host_pool *p) { delete p; }
extern "C" void *mrp_pool_current(void) { return t_pool_current; }
extern "C" void mrp_pool_adopt(void *p) { t_pool_current = static_cast<mrp_host_pool *>(p); }

extern "C" void mrp_pool_set_priority(int p) { t_pool_priority = p; }
extern "C" void mrp_pool_set_weight(int ns_per_index) { t_pool_weight_ns = ns_per_index; }
extern "C" void mrp_pool_set_tag(int t) { t_pool_tag = t; } /* MRP_TIMING: which loop the CPU time of the pool tasks is booked to */

static std::atomic<int> g_host_threads{0};
static std::atomic<bool> g_host_threads_set{false};
int mrp_host_threads(void) {
    int n = g_host_threads.load();
    if (n <= 0) {
        n = (int) std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
        g_host_threads.store(n);
    }
    return n;
}
int mrp_host_threads_setting(void) { return g_host_threads_set.load() ? g_host_threads.load() : 0; } /* 0: never set */
int mrp_set_host_threads(int n) {
    if (n < 1 || n > 256) return fail(MRP_ERR_ARG, "mrp_set_host_threads: %d outside 1..256", n);
    g_host_threads.store(n);
    g_host_threads_set.store(true);
    return MRP_OK;
}
int mrp_context_set_phase_groups(mrp_context *ctx, int groups) {
    if (!ctx || groups < 0 || groups > 16) return fail(MRP_ERR_ARG, "mrp_context_set_phase_groups: bad arguments");
    ctx->phase_groups = groups;
    return MRP_OK;
}
int mrp_context_phase_groups(const mrp_context *ctx) { return ctx->phase_groups; }
int mrp_context_set_test_hooks(mrp_context *ctx, int hooks) {
    if (!ctx || hooks < 0 || hooks > 15) return fail(MRP_ERR_ARG, "mrp_context_set_test_hooks: bad arguments");
    if ((hooks & 4) && ctx->pool.device >= 0) DevPoolRegistry::get().inject_oom[ctx->pool.device].store(1);
    hooks &= ~4;
    ctx->test_hooks = hooks;
    for (mrp_context *s_ : ctx->siblings) s_->test_hooks = hooks;
    return MRP_OK;
}
int mrp_context_test_hooks(const mrp_context *ctx) { return ctx->test_hooks; }

extern "C" {

int mrp_batch_add(mrp_batch *b, const mrp_hmm_job *job) { return mrp_batch_add_impl(b, job, false, nullptr, nullptr, nullptr); }

int mrp_batch_upload(mrp_batch *b) {
    if (!b) return fail(MRP_ERR_ARG, "batch is NULL");
    if (b->uploaded) return MRP_OK;
    mrp_context *ctx = b->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;

    const bool timing_ = getenv("MRP_TIMING") != nullptr;
    auto now_ = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return 1e3 * ts.tv_sec + 1e-6 * ts.tv_nsec; };
    const double u0 = now_();
    /* launch plan: int32/LDS path for max-plus HMMs that fit, fp64 path otherwise.  The int32 path is
     * split into size classes (LDS per workgroup = 2 * largest merge column * 4 B) that are launched
     * on separate streams so that small hmms do not inherit the residency of the largest one. */
    std::vector<std::pair<int64_t, int32_t>> wide, mid, narrow, generic, lse, lse_big;
    for (size_t i = 0; i < b->hmms.size(); i++) {
        const DevHmm &h = b->hmms[i];
        const int64_t work = b->outs[i].n_cells;
        const bool max_mode = (h.flags & MRP_FLAG_MAX_NOT_SUM) != 0;
        const size_t lds = (size_t) (2 * (int64_t) std::max(h.max_merge, 64) + 4) * sizeof(int32_t) + 128 * 8;
        if (max_mode && !h.wide_idx && b->outs[i].n_cells < (1ll << 30) && h.cost_bound < (1ll << 30) && lds <= (size_t) MRP_LDS_BUDGET) {
            b->outs[i].int_path = true;
            if (h.max_cells <= 256) narrow.push_back({-work, (int32_t) i});
            else if (h.max_merge <= 4096) mid.push_back({-work, (int32_t) i});
            else wide.push_back({-work, (int32_t) i});
        } else {
            generic.push_back({-work, (int32_t) i});
            /* sum mode with merge columns that fit LDS: the reproducible log-sum-exp kernel.  Its sums are 64-bit fixed point
             * in units of 2^-50 with every term <= 1: fewer than 2^14 terms per sum (a column's cells also sum into the
             * hmm's / column's total), and its reference points are floats: |log p| has to stay below 2^27, where a float
             * still resolves 8 -- beyond either bound the generic fp64 kernel takes the hmm */
            const bool lse_ok = !max_mode && !h.wide_idx && h.max_cells < MRP_LSE_MAX_TERMS && h.cost_bound < MRP_LSE_MAX_COST;
            if (lse_ok && h.max_merge <= MRP_LSE_CUR_LDS_MAX_MERGE) lse.push_back({-work, (int32_t) i});
            else if (lse_ok && h.max_merge <= MRP_LSE_MAX_MERGE) lse_big.push_back({-work, (int32_t) i});
        }
    }
    auto plan = [&](std::vector<std::pair<int64_t, int32_t>> &v, std::vector<int32_t> &order, int *max_merge) {
        std::sort(v.begin(), v.end()); /* largest first */
        order.clear();
        int mm = 1;
        for (auto &p : v) {
            order.push_back(p.second);
            mm = std::max(mm, b->hmms[p.second].max_merge);
        }
        if (max_merge) *max_merge = mm;
    };
    plan(wide, b->order_wide, &b->max_merge_wide);
    plan(mid, b->order_mid, &b->max_merge_mid);
    plan(narrow, b->order_narrow, &b->max_merge_narrow);
    plan(generic, b->order_f64, nullptr); /* every hmm with fp64 results */
    plan(lse, b->order_lse, &b->max_merge_lse);
    plan(lse_big, b->order_lse_big, &b->max_merge_lse_big);
    b->order_gen.clear();                  /* ... of which the generic kernel takes what the LDS ones do not */
    {
        std::vector<char> in_lse(b->hmms.size(), 0);
        for (int32_t i : b->order_lse) in_lse[(size_t) i] = 1;
        for (int32_t i : b->order_lse_big) in_lse[(size_t) i] = 1;
        for (int32_t i : b->order_f64)
            if (!in_lse[(size_t) i]) b->order_gen.push_back(i);
    }

    const double u1 = now_();
    std::vector<DevChunk> chunks;
    for (auto *c : b->chunks) chunks.push_back(c->dev);
    std::vector<int32_t> pack_list, plane_list;
    /* declared after the staging vectors: an early return drains the stream before they are destroyed */
    struct Drain { hipStream_t s; ~Drain() { (void) hipStreamSynchronize(s); } } drain{s};

    HIP_TRY(b->d_hmms.upload(b->hmms, s));
    HIP_TRY(b->d_cols.upload(b->cols, s));
    HIP_TRY(b->d_chunks.upload(chunks, s));
    HIP_TRY(b->d_read_byte_off.upload(b->read_byte_off, s));
    if (b->resident) { /* filled by the cross product kernel (mrp_engine.cpp) */
        HIP_TRY(b->d_partition.alloc((size_t) b->n_cells_total));
        HIP_TRY(b->d_np.alloc((size_t) b->n_cells_total));
    } else {
        HIP_TRY(b->d_partition.upload(b->partition, s));
        HIP_TRY(b->d_np.upload(b->cell_np, s));
    }
    HIP_TRY(b->d_scols.upload(b->scols, s));
    HIP_TRY(b->d_pcols.upload(b->pcols, s));
    if (b->need_wide) {
        HIP_TRY(b->d_next.upload(b->cell_next, s));
        HIP_TRY(b->d_prev.upload(b->cell_prev, s));
    }
    HIP_TRY(b->d_order_wide.upload(b->order_wide, s));
    HIP_TRY(b->d_order_mid.upload(b->order_mid, s));
    HIP_TRY(b->d_order_narrow.upload(b->order_narrow, s));
    HIP_TRY(b->d_order_f64.upload(b->order_gen, s));
    HIP_TRY(b->d_order_lse.upload(b->order_lse, s));
    HIP_TRY(b->d_order_lse_big.upload(b->order_lse_big, s));
    const double u2 = now_();
    if (!b->tilecols.empty()) { /* resident batch: the tiles are written on the device */
        HIP_TRY(b->d_tilecols.upload(b->tilecols, s));
        HIP_TRY(b->d_tiles.alloc((size_t) b->n_tiles_dev));
        HIP_TRY(mrp_launch_tiles(b->d_cols.p, b->d_tilecols.p, (int64_t) b->cols.size(), b->d_tiles.p, s));
    } else {
        {   /* fast tiles first */
            auto is_fast = [](const EmitTile &t) { return t.uniform_alleles != 0 && !(t.flags & MRP_FLAG_INCLUDE_ANCESTOR_SUB_PROB); };
            auto mid_it = std::stable_partition(b->tiles.begin(), b->tiles.end(), is_fast);
            b->n_fast_tiles = mid_it - b->tiles.begin();
        }
        b->n_tiles_dev = (int64_t) b->tiles.size();
        HIP_TRY(b->d_tiles.upload(b->tiles, s));
    }
    pack_list.reserve(b->pcols.size());
    for (size_t i = 0; i < b->pcols.size(); i++) (b->pcols[i].need_planes ? plane_list : pack_list).push_back((int32_t) i);
    HIP_TRY(b->d_pack_list.upload(pack_list, s));
    HIP_TRY(b->d_plane_list.upload(plane_list, s));
    const double u3 = now_();
    const size_t nC = (size_t) b->n_cells_total;
    HIP_TRY(b->d_planes.alloc((size_t) b->n_slots * 8));
    HIP_TRY(b->d_slot_total.alloc((size_t) b->n_slots));
    HIP_TRY(b->d_slot_bytes.alloc((size_t) b->n_slots * 16));
    HIP_TRY(b->d_cost.alloc(nC));
    HIP_TRY(b->d_f32.alloc(nC));
    HIP_TRY(b->d_b32.alloc(nC));
    HIP_TRY(b->d_mf32.alloc((size_t) b->n_merge));
    HIP_TRY(b->d_mb32.alloc((size_t) b->n_merge));
    if (!b->order_f64.empty()) { /* fp64 result arrays only when some hmm takes the fp64 path */
        HIP_TRY(b->d_f.alloc(nC));
        HIP_TRY(b->d_b.alloc(nC));
        HIP_TRY(b->d_mf.alloc((size_t) b->n_merge));
        HIP_TRY(b->d_mb.alloc((size_t) b->n_merge));
    }
    HIP_TRY(b->d_total.alloc(b->cols.size()));
    HIP_TRY(b->d_hmm_fb.alloc(2 * b->hmms.size()));
    const double u4 = now_();
    HIP_TRY(hipStreamSynchronize(s));
    if (timing_)
        fprintf(stderr, "      upload: plan %.2f ms, arrays %.2f, tiles+lists %.2f, allocs %.2f, sync %.2f\n", u1 - u0, u2 - u1, u3 - u2, u4 - u3, now_() - u4);

    MrpBatchDev &d = b->dev;
    d.hmms = b->d_hmms.p;
    d.cols = b->d_cols.p;
    d.chunks = b->d_chunks.p;
    d.read_byte_off = b->d_read_byte_off.p;
    d.partition = b->d_partition.p;
    d.scols = b->d_scols.p;
    d.pcols = b->d_pcols.p;
    d.pack_list = b->d_pack_list.p;
    d.plane_list = b->d_plane_list.p;
    d.n_pack_list = (int64_t) b->d_pack_list.n;
    d.n_plane_list = (int64_t) b->d_plane_list.n;
    d.cell_np = b->d_np.p;
    d.cell_next = b->d_next.p;
    d.cell_prev = b->d_prev.p;
    d.planes = b->d_planes.p;
    d.slot_total = b->d_slot_total.p;
    d.slot_bytes = b->d_slot_bytes.p;
    d.cell_cost = b->d_cost.p;
    d.cell_f32 = b->d_f32.p;
    d.cell_b32 = b->d_b32.p;
    d.merge_f32 = b->d_mf32.p;
    d.merge_b32 = b->d_mb32.p;
    d.cell_f = b->d_f.p;
    d.cell_b = b->d_b.p;
    d.merge_f = b->d_mf.p;
    d.merge_b = b->d_mb.p;
    d.col_total = b->d_total.p;
    d.hmm_fb = b->d_hmm_fb.p;
    d.n_hmms = (int64_t) b->hmms.size();
    d.n_cols = (int64_t) b->cols.size();
    d.n_cells = (int64_t) nC;
    d.n_merge = b->n_merge;
    d.n_slots = b->n_slots;
    /* host copies of the bulky inputs are no longer needed (a resident batch keeps its arrays' capacity for the next level) */
    if (!b->resident) HostVec<int64_t>().swap(b->read_byte_off);
    HostVec<uint64_t>().swap(b->partition);
    HostVec<uint32_t>().swap(b->cell_next);
    HostVec<uint32_t>().swap(b->cell_prev);
    HostVec<uint32_t>().swap(b->cell_np);
    b->uploaded = true;
    return MRP_OK;
}

int mrp_batch_launch(mrp_batch *b) {
    if (!b) return fail(MRP_ERR_ARG, "batch is NULL");
    if (!b->uploaded) {
        int rc = mrp_batch_upload(b);
        if (rc != MRP_OK) return rc;
    }
    mrp_context *ctx = b->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const MrpBatchDev &d = b->dev;
    hipStream_t ps = s; /* byte packing / bit planes ahead of the emission kernel, on the same stream */
    const size_t slot = (size_t) (b->n_launches % mrp_batch::EV_RING);
    if (slot >= b->ev_ring.size()) {
        std::array<hipEvent_t, 5> fresh{};
        for (auto &ev : fresh) HIP_TRY(hipEventCreate(&ev));
        b->ev_ring.push_back(fresh);
    }
    const std::array<hipEvent_t, 5> &ev = b->ev_ring[slot];
    if (b->n_launches >= mrp_batch::EV_RING) HIP_TRY(hipEventSynchronize(ev[2])); /* the launch that used these events has to be over */
    if (ps != s && ctx->last_emission) HIP_TRY(hipStreamWaitEvent(ps, ctx->last_emission, 0));
    HIP_TRY(hipEventRecord(ev[0], ps));
    HIP_TRY(mrp_launch_planes(d, ps));
    if (b->resident && mrp_dup('p')) HIP_TRY(mrp_launch_planes(d, ps));
    HIP_TRY(hipEventRecord(ev[1], ps));
    if (ps != s) HIP_TRY(hipStreamWaitEvent(s, ev[1], 0));
    HIP_TRY(hipEventRecord(ev[4], s));
    if (b->pre_sweep) HIP_TRY(b->pre_sweep(s)); /* resident merge levels: cross product + emission in one pass */
    if (b->pre_sweep && mrp_dup('x')) HIP_TRY(b->pre_sweep(s));
    HIP_TRY(mrp_launch_emission(d, b->d_tiles.p, b->n_fast_tiles, b->n_tiles_dev - b->n_fast_tiles, s));
    HIP_TRY(hipEventRecord(ev[3], s));
    ctx->last_emission = ev[3];
    if (!b->order_f64.empty()) {
        /* stRPHmm_initialiseProbs (hmm.c:752-789) for the accumulate-in-place fp64 path */
        const double neg = -__builtin_inf();
        HIP_TRY(mrp_launch_fill_f64(b->d_mf.p, d.n_merge, neg, s));
        HIP_TRY(mrp_launch_fill_f64(b->d_mb.p, d.n_merge, neg, s));
        HIP_TRY(mrp_launch_fill_f64(b->d_total.p, d.n_cols, neg, s));
        HIP_TRY(mrp_launch_fill_f64(b->d_hmm_fb.p, 2 * d.n_hmms, neg, s));
    }
    /* size classes side by side: wide on the main stream, mid and narrow on the auxiliary streams */
    hipStream_t a0 = s, a1 = s;
    if (s == ctx->stream) HIP_TRY(ctx->side_streams(&a0, &a1));
    const bool side = a0 != s;
    if (side) {
        HIP_TRY(hipEventRecord(ctx->fork, s));
        HIP_TRY(hipStreamWaitEvent(a0, ctx->fork, 0));
        HIP_TRY(hipStreamWaitEvent(a1, ctx->fork, 0));
    }
    /* Wide and mid hmms: 512 threads walk an hmm fastest, but the kernel's 128 registers then allow two workgroups to a CU.  When the
     * concurrent batches of a call together bring more such workgroups than the device has slots for (the top levels of a
     * 1 152-chunk call: 2 304 on 512 slots), 256 threads -- four to a CU -- get them through sooner: -3 % per call, A/B on one box;
     * a single batch, whose hmms all find a slot, stays at 512 (+4 % with 256). */
    const int64_t chains = 2 * (int64_t) (b->order_wide.size() + b->order_mid.size()) * (int64_t) ctx->concurrent_batches;
    int t_chain = ctx->concurrent_batches > 1 && chains > 2 * 256 ? 256 : 512; /* (a batch on its own -- the kernel replay of bench.py too: 512, as measured in rounds 1-3) */
    if (const char *ts = getenv("MRP_SWEEP_T")) { const int v = atoi(ts); if (v == 64 || v == 128 || v == 256 || v == 512) t_chain = v; } /* development */
    const int t_wide = t_chain, t_mid = t_chain, t_narrow = 64;
    /* workgroup sizes of the recursion kernel's classes (measured; in the
                                                         * concurrent batches of a call 64 to 512 threads for the mid class make no difference) */
    HIP_TRY(mrp_launch_sweep_i32(d, b->d_order_wide.p, (int64_t) b->order_wide.size(), t_wide, b->max_merge_wide, s));
    HIP_TRY(mrp_launch_sweep_i32(d, b->d_order_mid.p, (int64_t) b->order_mid.size(), t_mid, b->max_merge_mid, a0));
    HIP_TRY(mrp_launch_sweep_i32(d, b->d_order_narrow.p, (int64_t) b->order_narrow.size(), t_narrow, b->max_merge_narrow, a1));
    if (b->resident && mrp_dup('s')) {
        HIP_TRY(mrp_launch_sweep_i32(d, b->d_order_wide.p, (int64_t) b->order_wide.size(), t_wide, b->max_merge_wide, s));
        HIP_TRY(mrp_launch_sweep_i32(d, b->d_order_mid.p, (int64_t) b->order_mid.size(), t_mid, b->max_merge_mid, a0));
        HIP_TRY(mrp_launch_sweep_i32(d, b->d_order_narrow.p, (int64_t) b->order_narrow.size(), t_narrow, b->max_merge_narrow, a1));
    }
    HIP_TRY(mrp_launch_sweep_f64(d, b->d_order_f64.p, (int64_t) b->order_gen.size(), 256, s));
    HIP_TRY(mrp_launch_sweep_lse(d, b->d_order_lse.p, (int64_t) b->order_lse.size(), b->max_merge_lse, s));
    HIP_TRY(mrp_launch_sweep_lse(d, b->d_order_lse_big.p, (int64_t) b->order_lse_big.size(), std::max(b->max_merge_lse_big, MRP_LSE_CUR_LDS_MAX_MERGE + 2), s));
    if (side) {
        HIP_TRY(hipEventRecord(ctx->join[0], a0));
        HIP_TRY(hipEventRecord(ctx->join[1], a1));
        HIP_TRY(hipStreamWaitEvent(s, ctx->join[0], 0));
        HIP_TRY(hipStreamWaitEvent(s, ctx->join[1], 0));
    }
    HIP_TRY(hipEventRecord(ev[2], s));
    b->n_launches++;
    b->launched = true;
    return MRP_OK;
}

int mrp_batch_stats(mrp_batch *b, mrp_launch_stats *out) {
    if (!b || !out) return fail(MRP_ERR_ARG, "mrp_batch_stats: NULL argument");
    *out = b->stats;
    out->n_hmms_lse = (int64_t) (b->order_lse.size() + b->order_lse_big.size());
    out->n_hmms_generic = (int64_t) b->order_gen.size();
    out->n_hmms_int32 = (int64_t) (b->order_wide.size() + b->order_mid.size() + b->order_narrow.size());
    if (b->launched) {
        mrp_context *ctx = b->ctx;
        HIP_TRY(hipSetDevice(ctx->device));
        /* the launches since the previous call (the most recent one at least), as far back as the ring reaches */
        const int64_t k = std::min<int64_t>(std::max<int64_t>(b->n_launches - b->stats_mark, 1), std::min<int64_t>(b->n_launches, mrp_batch::EV_RING));
        b->stats_mark = b->n_launches;
        double sa = 0, se = 0, sc = 0;
        float a = 0, e = 0, c = 0;
        for (int64_t j = 0; j < k; j++) { /* oldest first; the last one is the most recent launch */
            const auto &ev = b->ev_ring[(size_t) ((b->n_launches - k + j) % mrp_batch::EV_RING)];
            HIP_TRY(hipEventSynchronize(ev[2]));
            HIP_TRY(hipEventElapsedTime(&a, ev[0], ev[1]));
            HIP_TRY(hipEventElapsedTime(&e, ev[4], ev[3]));
            HIP_TRY(hipEventElapsedTime(&c, ev[3], ev[2]));
            sa += a; se += e; sc += c;
        }
        out->planes_ms = a;
        out->emission_ms = e;
        out->sweep_ms = c;
        out->avg_planes_ms = k ? sa / (double) k : 0.0;
        out->avg_emission_ms = k ? se / (double) k : 0.0;
        out->avg_sweep_ms = k ? sc / (double) k : 0.0;
        out->launches_averaged = k;
    }
    return MRP_OK;
}

/* the most recent launch by kernel family (ms): packing / bit planes, cross product + emission, recursion.  Called after the
 * stream the launch ran on has been waited for. */
void mrp_batch_last_launch_ms(mrp_batch *b, float *pack, float *emission, float *recursion) {
    *pack = *emission = *recursion = 0.f;
    if (!b || !b->launched || b->n_launches < 1) return;
    const auto &ev = b->ev_ring[(size_t) ((b->n_launches - 1) % mrp_batch::EV_RING)];
    (void) hipEventElapsedTime(pack, ev[0], ev[1]);
    (void) hipEventElapsedTime(emission, ev[4], ev[3]);
    (void) hipEventElapsedTime(recursion, ev[3], ev[2]);
}

int mrp_batch_download(mrp_batch *b) {
    if (!b) return fail(MRP_ERR_ARG, "batch is NULL");
    if (!b->launched) return fail(MRP_ERR_ARG, "mrp_batch_download before mrp_batch_launch");
    mrp_context *ctx = b->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const MrpBatchDev &d = b->dev;
    const bool any_out = std::any_of(b->outs.begin(), b->outs.end(), [](const JobOut &o) { return o.cell_f != nullptr; });
    if (!any_out) {
        HIP_TRY(hipStreamSynchronize(s));
        return MRP_OK;
    }
    const bool has_f64 = !b->order_f64.empty();
    std::vector<double> f, bb, mf, mb, tot((size_t) d.n_cols), fb((size_t) (2 * d.n_hmms));
    std::vector<int32_t> f32((size_t) d.n_cells), b32((size_t) d.n_cells), mf32((size_t) d.n_merge), mb32((size_t) d.n_merge);
    if (has_f64) { f.resize((size_t) d.n_cells); bb.resize((size_t) d.n_cells); mf.resize((size_t) d.n_merge); mb.resize((size_t) d.n_merge); }
    auto pull = [&](void *h, const void *p, size_t bytes) -> hipError_t {
        if (bytes == 0) return hipSuccess;
        return hipMemcpyAsync(h, p, bytes, hipMemcpyDeviceToHost, s);
    };
    HIP_TRY(pull(f32.data(), d.cell_f32, f32.size() * 4));
    HIP_TRY(pull(b32.data(), d.cell_b32, b32.size() * 4));
    HIP_TRY(pull(mf32.data(), d.merge_f32, mf32.size() * 4));
    HIP_TRY(pull(mb32.data(), d.merge_b32, mb32.size() * 4));
    if (has_f64) {
        HIP_TRY(pull(f.data(), d.cell_f, f.size() * 8));
        HIP_TRY(pull(bb.data(), d.cell_b, bb.size() * 8));
        HIP_TRY(pull(mf.data(), d.merge_f, mf.size() * 8));
        HIP_TRY(pull(mb.data(), d.merge_b, mb.size() * 8));
    }
    HIP_TRY(pull(tot.data(), d.col_total, tot.size() * 8));
    HIP_TRY(pull(fb.data(), d.hmm_fb, fb.size() * 8));
    HIP_TRY(hipStreamSynchronize(s));
    /* widen the max-plus integers to the reference's doubles (exact); MRP_NEG_I32 is log(0) */
    auto widen = [](double *dst, const int32_t *src, int64_t n) {
        for (int64_t i = 0; i < n; i++) dst[i] = src[i] == MRP_NEG_I32 ? -__builtin_inf() : (double) src[i];
    };
    for (size_t i = 0; i < b->outs.size(); i++) {
        const JobOut &o = b->outs[i];
        if (!o.cell_f) continue; /* device-only job */
        if (o.int_path) {
            widen(o.cell_f, f32.data() + o.cell0, o.n_cells);
            widen(o.cell_b, b32.data() + o.cell0, o.n_cells);
            if (o.n_merge > 0) {
                widen(o.merge_f, mf32.data() + o.mcell0, o.n_merge);
                widen(o.merge_b, mb32.data() + o.mcell0, o.n_merge);
            }
        } else {
            memcpy(o.cell_f, f.data() + o.cell0, sizeof(double) * (size_t) o.n_cells);
            memcpy(o.cell_b, bb.data() + o.cell0, sizeof(double) * (size_t) o.n_cells);
            if (o.n_merge > 0) {
                memcpy(o.merge_f, mf.data() + o.mcell0, sizeof(double) * (size_t) o.n_merge);
                memcpy(o.merge_b, mb.data() + o.mcell0, sizeof(double) * (size_t) o.n_merge);
            }
        }
        memcpy(o.col_total, tot.data() + o.col0, sizeof(double) * (size_t) o.n_cols);
        *o.hmm_f = fb[2 * i];
        *o.hmm_b = fb[2 * i + 1];
    }
    return MRP_OK;
}

int mrp_fb_run(mrp_context *ctx, int64_t n_jobs, const mrp_hmm_job *jobs) {
    if (!ctx || n_jobs < 0 || (n_jobs > 0 && !jobs)) return fail(MRP_ERR_ARG, "mrp_fb_run: bad arguments");
    if (n_jobs == 0) return MRP_OK;
    mrp_batch *b = nullptr;
    int rc = mrp_batch_create(ctx, &b);
    for (int64_t i = 0; rc == MRP_OK && i < n_jobs; i++) rc = mrp_batch_add(b, &jobs[i]);
    if (rc == MRP_OK) rc = mrp_batch_upload(b);
    if (rc == MRP_OK) rc = mrp_batch_launch(b);
    if (rc == MRP_OK) rc = mrp_batch_download(b);
    mrp_batch_destroy(b);
    return rc;
}

/* ---- emission-only seam -------------------------------------------------------------------- */
static int one_column(mrp_context *ctx, const mrp_chunk *chunk, int32_t first_site, int32_t n_sites, int32_t depth,
                      const int64_t *read_byte_off, DevCol *col) {
    if (!ctx || !chunk || chunk->ctx != ctx) return fail(MRP_ERR_ARG, "bad context/chunk");
    if (chunk->host_wait() != hipSuccess) return fail(MRP_ERR_HIP, "chunk upload failed");
    if (depth < 0 || depth > MRP_MAX_READ_PARTITIONING_DEPTH || n_sites < 0 || first_site < 0 ||
        (int64_t) first_site + n_sites > chunk->n_sites || (depth > 0 && !read_byte_off))
        return fail(MRP_ERR_ARG, "bad column description");
    memset(col, 0, sizeof(*col));
    col->n_cells = 0;
    col->site_start = first_site;
    col->n_sites = n_sites;
    col->depth = depth;
    col->n_slots = (int32_t) (chunk->allele_offset[first_site + n_sites] - chunk->allele_offset[first_site]);
    for (int i = 0; i < depth; i++)
        if (read_byte_off[i] < 0 || read_byte_off[i] + col->n_slots > chunk->pool_bytes)
            return fail(MRP_ERR_ARG, "read %d: profile bytes outside the pool", i);
    return MRP_OK;
}

static int run_planes(mrp_context *ctx, const mrp_chunk *chunk, const DevCol &col, const int64_t *read_byte_off,
                      DevBuf<DevCol> &d_col, DevBuf<DevChunk> &d_chunk, DevBuf<int64_t> &d_off,
                      DevBuf<uint64_t> &d_planes, DevBuf<uint32_t> &d_tot) {
    hipStream_t s = ctx->stream;
    DevBuf<uint32_t> d_bytes;
    DevBuf<PlaneCol> d_pcol;
    PlaneCol pc{};
    pc.pool = chunk->dev.pool;
    pc.read_off = 0;
    pc.slot_off = 0;
    pc.depth = col.depth;
    pc.n_slots = col.n_slots;
    pc.need_planes = 1;
    /* host staging of the queued uploads */
    std::vector<DevCol> hc(1, col);
    std::vector<DevChunk> hch(1, chunk->dev);
    std::vector<int64_t> ho(read_byte_off, read_byte_off + col.depth);
    std::vector<PlaneCol> hpc(1, pc);
    /* declared last, so it runs first: whatever way this function is left, the stream is drained before the staging vectors
     * above and the scratch buffers (written by the kernel) go */
    struct Drain { hipStream_t s; ~Drain() { (void) hipStreamSynchronize(s); } } drain{s};
    HIP_TRY(d_col.upload(hc, s));
    HIP_TRY(d_chunk.upload(hch, s));
    HIP_TRY(d_off.upload(ho, s));
    HIP_TRY(d_planes.alloc((size_t) col.n_slots * 8));
    HIP_TRY(d_tot.alloc((size_t) col.n_slots));
    HIP_TRY(d_bytes.alloc((size_t) col.n_slots * 16));
    HIP_TRY(d_pcol.upload(hpc, s));
    MrpBatchDev d{};
    d.pcols = d_pcol.p;
    d.cols = d_col.p;
    d.chunks = d_chunk.p;
    d.read_byte_off = d_off.p;
    d.planes = d_planes.p;
    d.slot_total = d_tot.p;
    d.slot_bytes = d_bytes.p;
    d.n_cols = 1;
    HIP_TRY(mrp_launch_planes(d, s));
    return MRP_OK;
}

int mrp_count_bit_vectors(mrp_context *ctx, const mrp_chunk *chunk, int32_t first_site, int32_t n_sites, int32_t depth,
                          const int64_t *read_byte_off, uint64_t *planes_out) {
    DevCol col;
    int rc = one_column(ctx, chunk, first_site, n_sites, depth, read_byte_off, &col);
    if (rc != MRP_OK) return rc;
    if (col.n_slots == 0) return MRP_OK;
    if (!planes_out) return fail(MRP_ERR_ARG, "planes_out is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    DevBuf<DevCol> d_col; DevBuf<DevChunk> d_chunk; DevBuf<int64_t> d_off; DevBuf<uint64_t> d_planes; DevBuf<uint32_t> d_tot;
    rc = run_planes(ctx, chunk, col, read_byte_off, d_col, d_chunk, d_off, d_planes, d_tot);
    if (rc != MRP_OK) return rc;
    HIP_TRY(hipMemcpyAsync(planes_out, d_planes.p, sizeof(uint64_t) * 8 * (size_t) col.n_slots, hipMemcpyDeviceToHost,
                           ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return MRP_OK;
}

int mrp_emissions(mrp_context *ctx, const mrp_chunk *chunk, int32_t first_site, int32_t n_sites, int32_t depth,
                  const int64_t *read_byte_off, uint32_t flags, int64_t n_cells, const uint64_t *partitions,
                  double *out) {
    DevCol col;
    int rc = one_column(ctx, chunk, first_site, n_sites, depth, read_byte_off, &col);
    if (rc != MRP_OK) return rc;
    if (n_cells < 0 || (n_cells > 0 && (!partitions || !out))) return fail(MRP_ERR_ARG, "bad cell arrays");
    if (n_cells == 0) return MRP_OK;
    if (flags & MRP_FLAG_INCLUDE_ANCESTOR_SUB_PROB)
        for (int s = 0; s < n_sites; s++)
            if (chunk->allele_number[first_site + s] > MRP_MAX_ALLELES)
                return fail(MRP_ERR_UNSUPPORTED, "site %d has more than %d alleles (ancestor mode)", first_site + s,
                            MRP_MAX_ALLELES);
    HIP_TRY(hipSetDevice(ctx->device));
    DevBuf<DevCol> d_col; DevBuf<DevChunk> d_chunk; DevBuf<int64_t> d_off; DevBuf<uint64_t> d_planes; DevBuf<uint32_t> d_tot;
    rc = run_planes(ctx, chunk, col, read_byte_off, d_col, d_chunk, d_off, d_planes, d_tot);
    if (rc != MRP_OK) return rc;
    DevBuf<uint64_t> d_part; DevBuf<double> d_out;
    std::vector<uint64_t> hp(partitions, partitions + n_cells);
    HIP_TRY(d_part.upload(hp, ctx->stream));
    HIP_TRY(d_out.alloc((size_t) n_cells));
    HIP_TRY(mrp_launch_emissions(d_col.p, d_chunk.p, d_planes.p, d_tot.p, flags, n_cells, d_part.p, d_out.p, ctx->stream));
    HIP_TRY(hipMemcpyAsync(out, d_out.p, sizeof(double) * (size_t) n_cells, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return MRP_OK;
}

}  /* extern "C" */
