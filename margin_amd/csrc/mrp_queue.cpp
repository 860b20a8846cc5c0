/*
 * mrp_queue.cpp -- the host-side work queue that spreads genome chunks over the GPUs of one node.
 *
 * Reference: the chunk loop of phase.c.  The chunks are ordered by estimated depth, largest first (phase.c:257-263,
 * SCM_SIZE_DESC), and handed to the threads one at a time as they become free (phase.c:276-279,
 * "#pragma omp parallel for schedule(dynamic,1)"); chunks are independent until stitching, nothing is exchanged.
 * Here a "thread" is a device: one host thread and one context per device pull the next BATCH of chunks (a batch is
 * what one mrp_phase_reads_many call phases; its chunks share kernel launches), upload them, phase them and write the
 * results at the chunks' positions of the caller's array.  No collective, no device-to-device traffic.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "mrp_internal.h"

namespace {

struct QueuePlan {
    std::vector<int64_t> order;     /* chunk indices, largest estimated cost first (ties: input order) */
    std::vector<int64_t> batch_off; /* batch b = order[batch_off[b] .. batch_off[b + 1]) */
};

/* phase.c:257-263: sort by estimated size, largest first; then cut into batches of consecutive chunks */
QueuePlan plan_queue(int64_t n, const int64_t *cost, int64_t chunks_per_batch) {
    QueuePlan p;
    p.order.resize((size_t) n);
    std::iota(p.order.begin(), p.order.end(), (int64_t) 0);
    std::stable_sort(p.order.begin(), p.order.end(), [&](int64_t a, int64_t b) { return cost[a] > cost[b]; });
    if (chunks_per_batch < 1) chunks_per_batch = 1;
    for (int64_t o = 0; o < n; o += chunks_per_batch) p.batch_off.push_back(o);
    p.batch_off.push_back(n);
    return p;
}

/* One worker per device: pulls batch indices from the shared counter until the queue is empty (schedule(dynamic,1)).
 * fn(worker, first position in plan.order, count) returns a status; the first failure stops the queue. */
template <class Fn>
int run_queue(int n_workers, const QueuePlan &plan, Fn fn, std::vector<int32_t> *worker_of_batch) {
    const int64_t n_batches = (int64_t) plan.batch_off.size() - 1;
    std::atomic<int64_t> next{0};
    std::atomic<int> status{MRP_OK};
    if (worker_of_batch) worker_of_batch->assign((size_t) n_batches, -1);
    auto work = [&](int w) {
        for (;;) {
            if (status.load() != MRP_OK) return;
            const int64_t b = next.fetch_add(1);
            if (b >= n_batches) return;
            if (worker_of_batch) (*worker_of_batch)[(size_t) b] = w;
            const int rc = fn(w, b, plan.batch_off[(size_t) b], plan.batch_off[(size_t) b + 1] - plan.batch_off[(size_t) b]);
            if (rc != MRP_OK) { int expect = MRP_OK; status.compare_exchange_strong(expect, rc); return; }
        }
    };
    std::vector<std::thread> th;
    for (int w = 1; w < n_workers; w++) th.emplace_back(work, w);
    work(0);
    for (auto &t : th) t.join();
    return status.load();
}

}  // namespace

extern "C" {

int mrp_queue_plan(int64_t n_chunks, const int64_t *cost, int64_t chunks_per_batch, int64_t *order_out, int64_t *batch_of_chunk_out) {
    if (n_chunks < 0 || (n_chunks > 0 && (!cost || !order_out))) return mrp_set_error(MRP_ERR_ARG, "mrp_queue_plan: bad arguments");
    const QueuePlan p = plan_queue(n_chunks, cost, chunks_per_batch);
    for (int64_t i = 0; i < n_chunks; i++) order_out[i] = p.order[(size_t) i];
    if (batch_of_chunk_out)
        for (size_t b = 0; b + 1 < p.batch_off.size(); b++)
            for (int64_t i = p.batch_off[b]; i < p.batch_off[b + 1]; i++) batch_of_chunk_out[p.order[(size_t) i]] = (int64_t) b;
    return MRP_OK;
}

int mrp_queue_dry_run(int32_t n_workers, int64_t n_chunks, const int64_t *cost, int64_t chunks_per_batch, double usec_per_cost,
                      int32_t *worker_of_chunk_out, int64_t *sequence_out) {
    if (n_workers < 1 || n_workers > MRP_MAX_QUEUE_DEVICES || n_chunks < 0 || (n_chunks > 0 && (!cost || !worker_of_chunk_out)))
        return mrp_set_error(MRP_ERR_ARG, "mrp_queue_dry_run: bad arguments");
    const QueuePlan p = plan_queue(n_chunks, cost, chunks_per_batch);
    std::atomic<int64_t> seq{0};
    std::vector<int32_t> wob;
    const int rc = run_queue(n_workers, p, [&](int w, int64_t, int64_t first, int64_t count) {
        int64_t c = 0;
        for (int64_t i = first; i < first + count; i++) {
            const int64_t chunk = p.order[(size_t) i];
            worker_of_chunk_out[chunk] = w;
            if (sequence_out) sequence_out[chunk] = seq.fetch_add(1);
            c += cost[chunk];
        }
        if (usec_per_cost > 0) std::this_thread::sleep_for(std::chrono::microseconds((int64_t) (usec_per_cost * (double) c)));
        return (int) MRP_OK;
    }, &wob);
    return rc;
}

struct mrp_queue {
    std::vector<int32_t> devices;
    std::vector<mrp_context *> ctx; /* one per worker, created by the worker on its first batch, kept between calls */
};

int mrp_queue_create(const int32_t *devices, int32_t n_devices, mrp_queue **out) {
    if (!devices || !out || n_devices < 1 || n_devices > MRP_MAX_QUEUE_DEVICES) return mrp_set_error(MRP_ERR_ARG, "mrp_queue_create: bad arguments");
    *out = nullptr;
    const int visible = mrp_device_count();
    if (visible <= 0) return mrp_set_error(MRP_ERR_NO_DEVICE, "no HIP device visible: the stRPHmm sweep has no CPU fallback");
    for (int d = 0; d < n_devices; d++)
        if (devices[d] < 0 || devices[d] >= visible) return mrp_set_error(MRP_ERR_ARG, "device %d out of range (0..%d)", devices[d], visible - 1);
    mrp_queue *q = new (std::nothrow) mrp_queue();
    if (!q) return mrp_set_error(MRP_ERR_NOMEM, "out of host memory");
    q->devices.assign(devices, devices + n_devices);
    q->ctx.assign((size_t) n_devices, nullptr);
    *out = q;
    return MRP_OK;
}

void mrp_queue_destroy(mrp_queue *q) {
    if (!q) return;
    for (auto *c : q->ctx) mrp_context_destroy(c);
    delete q;
}

int mrp_queue_phase_chunks(mrp_queue *q, int64_t n_chunks, const mrp_chunk_desc *chunks, const mrp_params *params, int64_t chunks_per_batch,
                           mrp_phase_result **out, mrp_queue_stats *stats) {
    if (!q || n_chunks < 0 || !params || (n_chunks > 0 && (!chunks || !out))) return mrp_set_error(MRP_ERR_ARG, "mrp_queue_phase_chunks: bad arguments");
    const int n_devices = (int) q->devices.size();
    const int32_t *devices = q->devices.data();
    std::vector<mrp_context *> &ctx = q->ctx;
    if (stats) { memset(stats, 0, sizeof(*stats)); stats->n_devices = n_devices; }
    for (int64_t i = 0; i < n_chunks; i++) out[i] = nullptr;
    if (n_chunks == 0) return MRP_OK;
    /* estimated cost of a chunk: its het-sites x reads (what the depth estimate of phase.c:259 stands for) */
    std::vector<int64_t> cost((size_t) n_chunks, 0);
    for (int64_t i = 0; i < n_chunks; i++) {
        const mrp_chunk_desc &c = chunks[i];
        if (c.n_sites < 0 || c.n_reads < 0 || (c.n_reads > 0 && !c.reads) || (c.n_sites > 0 && !c.allele_number))
            return mrp_set_error(MRP_ERR_ARG, "chunk %lld: bad description", (long long) i);
        for (int64_t r = 0; r < c.n_reads; r++) cost[(size_t) i] += c.reads[r].length;
    }
    if (chunks_per_batch < 1) chunks_per_batch = 48;
    const QueuePlan plan = plan_queue(n_chunks, cost.data(), chunks_per_batch);
    if (stats) stats->batches = (int64_t) plan.batch_off.size() - 1;

    std::vector<std::string> errs((size_t) n_devices);
    struct PerDev { int64_t chunks = 0, units = 0, fallback = 0; double busy_ms = 0; };
    std::vector<PerDev> per((size_t) n_devices);
    const int rc = run_queue(n_devices, plan, [&](int w, int64_t, int64_t first, int64_t count) {
        auto t0 = std::chrono::steady_clock::now();
        int r = MRP_OK;
        if (!ctx[(size_t) w]) r = mrp_context_create(devices[w], &ctx[(size_t) w]);
        std::vector<mrp_chunk *> dch((size_t) count, nullptr);
        std::vector<const mrp_chunk *> cch((size_t) count);
        std::vector<const mrp_read *> rd((size_t) count);
        std::vector<int64_t> nr((size_t) count);
        std::vector<mrp_phase_result *> res((size_t) count, nullptr);
        for (int64_t i = 0; i < count && r == MRP_OK; i++) { /* upload: site tables + profile bytes of the batch's chunks */
            const mrp_chunk_desc &c = chunks[plan.order[(size_t) (first + i)]];
            r = mrp_chunk_create(ctx[(size_t) w], c.n_sites, c.allele_number, c.substitution_log_probs, c.allele_prior_log_probs, c.profile_pool,
                                 c.pool_bytes, &dch[(size_t) i]);
            cch[(size_t) i] = dch[(size_t) i]; rd[(size_t) i] = c.reads; nr[(size_t) i] = c.n_reads;
        }
        mrp_phase_many_stats ps{};
        if (r == MRP_OK) r = mrp_phase_reads_many(ctx[(size_t) w], count, cch.data(), rd.data(), nr.data(), params, res.data(), &ps);
        for (int64_t i = 0; i < count; i++) {
            const int64_t chunk = plan.order[(size_t) (first + i)];
            if (r == MRP_OK) { out[chunk] = res[(size_t) i]; per[(size_t) w].units += cost[(size_t) chunk]; }
            if (dch[(size_t) i]) mrp_chunk_destroy(dch[(size_t) i]);
        }
        if (r == MRP_OK) { per[(size_t) w].chunks += count; per[(size_t) w].fallback += ps.resident ? ps.fallback_chunks : count; }
        else errs[(size_t) w] = mrp_last_error();
        per[(size_t) w].busy_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return r;
    }, nullptr);
    if (stats)
        for (int d = 0; d < n_devices; d++) {
            stats->chunks_per_device[d] = per[(size_t) d].chunks;
            stats->units_per_device[d] = per[(size_t) d].units;
            stats->busy_ms_per_device[d] = per[(size_t) d].busy_ms;
            stats->fallback_chunks += per[(size_t) d].fallback;
        }
    if (rc != MRP_OK) {
        for (int64_t i = 0; i < n_chunks; i++) { mrp_phase_result_destroy(out[i]); out[i] = nullptr; }
        for (auto &e : errs)
            if (!e.empty()) return mrp_set_error(rc, "%s", e.c_str());
        return mrp_set_error(rc, "work queue stopped");
    }
    return MRP_OK;
}

int mrp_phase_chunks_on_devices(const int32_t *devices, int32_t n_devices, int64_t n_chunks, const mrp_chunk_desc *chunks,
                                const mrp_params *params, int64_t chunks_per_batch, mrp_phase_result **out, mrp_queue_stats *stats) {
    mrp_queue *q = nullptr;
    int rc = mrp_queue_create(devices, n_devices, &q);
    if (rc == MRP_OK) rc = mrp_queue_phase_chunks(q, n_chunks, chunks, params, chunks_per_batch, out, stats);
    mrp_queue_destroy(q);
    return rc;
}

}  /* extern "C" */
