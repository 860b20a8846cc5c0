/*
 * mrp_queue.cpp -- the host-side work queue that spreads genome chunks over the GPUs of one node.
 *
 * Reference: the chunk loop of phase.c.  The chunks are ordered by estimated depth, largest first (phase.c:257-263,
 * SCM_SIZE_DESC), and handed to the threads one at a time as they become free (phase.c:276-279,
 * "#pragma omp parallel for schedule(dynamic,1)"); chunks are independent until stitching, nothing is exchanged.
 * Here a "thread" is a device: one host thread and one context per device pull the next BATCH of chunks (a batch is
 * what one mrp_phase_reads_many call phases; its chunks share kernel launches), upload them, phase them and write the
 * results at the chunks' positions of the caller's array.  No collective, no device-to-device traffic.  The uploads of a
 * worker's next batch run on a second stream while the current batch is phased; every worker has its own host thread pool.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <functional>
#include <system_error>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include <pthread.h>
#include <sched.h>
#include <cctype>

#include "mrp_internal.h"

int mrp_queue_threads_per_device(int n_devices);

namespace {

struct QueuePlan {
    std::vector<int64_t> order;     /* chunk indices, largest estimated cost first (ties: input order) */
    std::vector<int64_t> batch_off; /* batch b = order[batch_off[b] .. batch_off[b + 1]) */
};

/* phase.c:257-263: sort by estimated size, largest first; then cut into batches of consecutive chunks.
 * chunks_per_batch >= 1: batches of that many chunks.  0: the library's choice for n_workers pulling threads on n_devices
 * devices.  A batch is one mrp_phase_reads_many call; a call begins and ends with host work and walks its merge levels one
 * after the other, the top ones bound by per-column latency whatever the number of chunks.  A short queue (up to 1 280 chunks
 * of the 1 Mb kind per device -- 640 through round 3, when a device held twice as much per chunk; sizes are counted in units, below) is ONE batch per device: the call runs it as eight concurrent batches of its own (576 chunks: 170-182 ms).  A
 * longer one is handed out in batches of MRP_QUEUE_DEFAULT_BATCH chunks to the lanes of the devices (four per device, each
 * call two concurrent batches: the same eight in flight, but at different levels -- while one lane is in its host-bound
 * head or its latency-bound top levels the others stream; measured on 2 304 chunks: 153-160 ms per 576 against 174-177 for
 * one lane of 576-chunk calls), shrinking towards the end (several devices: "guided" schedule, remaining / workers, at least 96) so that the
 * devices finish within two percent of each other (tests/test_work_queue.py: 8 devices, 31 000 chunks) without the tail of
 * the queue dissolving into small, latency-bound calls. */
QueuePlan plan_queue(int64_t n, const int64_t *cost, int64_t chunks_per_batch, int n_workers = 1, int n_devices = 1) {
    QueuePlan p;
    p.order.resize((size_t) n);
    std::iota(p.order.begin(), p.order.end(), (int64_t) 0);
    std::stable_sort(p.order.begin(), p.order.end(), [&](int64_t a, int64_t b) { return cost[a] > cost[b]; });
    /* Sizes are counted in UNITS (a chunk's cost: its reads x the het sites they span), with the 1 Mb, 30x chunk of
     * BASELINE.json configs[1] (60 000 units) as the yardstick the policy was measured on: what a call costs the device and
     * what it keeps there grow with its units, not with its chunks -- 192 chunks of 130 sites would be a call of 9 ms. */
    const int64_t unit_chunk = MRP_QUEUE_UNITS_PER_CHUNK;
    int64_t short_chunks = MRP_QUEUE_SHORT_CHUNKS;
    if (const char *ev = getenv("MRP_QUEUE_SHORT_CHUNKS")) { const long v = atol(ev); if (v > 0) short_chunks = v; } /* (development) */
    const int64_t big = MRP_QUEUE_DEFAULT_BATCH * unit_chunk, short_queue = short_chunks * unit_chunk;
    auto units = [&](int64_t pos) { return std::max<int64_t>(1, cost[p.order[(size_t) pos]]); };
    int64_t total = 0;
    for (int64_t i = 0; i < n; i++) total += units(i);
    if (chunks_per_batch >= 1) {
        for (int64_t o = 0; o < n; o += chunks_per_batch) p.batch_off.push_back(o);
    } else if (total <= (int64_t) n_devices * short_queue) {
        /* up to MRP_QUEUE_SHORT_CHUNKS yardstick chunks per device ONE call per device is the fastest */
        const int64_t parts = std::min<int64_t>(n, n_devices);
        /* these batches all start at once: deal the chunks out in stripes (batch b takes the b-th, (b + parts)-th, ... of the
         * cost order), so that every batch gets its share of the expensive ones -- consecutive runs of a largest-first order
         * would make the first batch the slowest by a third */
        std::vector<int64_t> striped;
        striped.reserve((size_t) n);
        for (int64_t b = 0; b < parts; b++) {
            p.batch_off.push_back((int64_t) striped.size());
            for (int64_t i = b; i < n; i += parts) striped.push_back(p.order[(size_t) i]);
        }
        p.order.swap(striped);
    } else {
        int64_t left = total;
        /* one device: its lanes share it, so there is nothing to even out between them towards the end -- but they should all be
         * busy to the end: as many equal batches as fill whole rounds of the lanes (1 152 chunks on four lanes: 8 batches of 144
         * rather than 6 of 192, of which the last two would run on half the lanes) */
        const int64_t lanes_of_device = std::max(1, n_workers / std::max(1, n_devices));
        const int64_t rounds = (total + lanes_of_device * big - 1) / (lanes_of_device * big);
        const int64_t even = (total + rounds * lanes_of_device - 1) / (rounds * lanes_of_device);
        for (int64_t o = 0; o < n;) { /* several devices: full batches while every lane can still get two, then shrinking, at least a quarter batch */
            p.batch_off.push_back(o);
            /* (a lane holds two batches, the one it phases and the one it took ahead: what is left is shared out as if there were
             * twice the lanes; the floor of a quarter batch keeps the tail from dissolving into latency-bound calls) */
            const int64_t target = n_devices > 1 ? std::max<int64_t>(big / 4, std::min<int64_t>(big, left / (2 * (int64_t) std::max(1, n_workers)))) : even;
            int64_t got = 0;
            do { got += units(o); o++; } while (o < n && got + units(o) / 2 < target);
            if (left - got < target / 2) /* (what would be left is no batch of its own) */
                while (o < n) { got += units(o); o++; }
            left -= got;
        }
    }
    p.batch_off.push_back(n);
    return p;
}

/* How many calls of a device run at a time: a device holds what its calls in flight need (some 170 MB per 1 Mb chunk = 60 000 units,
 * DevPool): four lanes for batches of up to 288 such chunks, two up to 576, one beyond; a queue of no more batches than
 * devices needs one lane each. */
int active_lanes_of(const QueuePlan &plan, const int64_t *cost, int n_devices, int lanes) {
    const int64_t n_batches = (int64_t) plan.batch_off.size() - 1;
    int64_t max_batch = 0;
    for (int64_t b = 0; b < n_batches; b++) {
        int64_t u = 0;
        for (int64_t i = plan.batch_off[(size_t) b]; i < plan.batch_off[(size_t) b + 1]; i++) u += cost[(size_t) plan.order[(size_t) i]];
        max_batch = std::max(max_batch, u);
    }
    const int64_t yard = MRP_QUEUE_UNITS_PER_CHUNK;
    return n_batches <= (int64_t) n_devices ? 1 : (max_batch <= 288 * yard ? lanes : (max_batch <= 576 * yard ? std::min(lanes, 2) : 1));
}

/* The hand-out of the queue, shared by the real workers and the dry run: lane w = device w / lanes; a lane that is active takes
 * its first batch (fixed, below), then -- schedule(dynamic,1), one batch ahead -- takes its NEXT batch when it starts on the current one
 * (`prefetch(w, next batch)` runs beside `phase(w, current batch)`: the real worker uploads there) and goes on until the shared
 * counter runs out.  A lane without a first batch does nothing (no contexts, no streams).  begin(w) / end(w) bracket a lane that
 * got work; any non-zero status stops the queue.  Every lane is a host thread; lane 0 runs on the caller's thread unless
 * `own_thread_for_lane0`. */
struct LaneHooks {
    std::function<int(int)> begin;                     /* lane */
    std::function<int(int, int64_t)> prefetch;         /* lane, batch: may run on another thread than phase */
    std::function<int(int, int64_t)> phase;            /* lane, batch */
    std::function<void(int, int64_t)> retire;          /* lane, batch: after phase AND after the prefetch that ran beside it has ended */
    std::function<void(int, int64_t)> discard;         /* lane, batch: a prefetched batch that will not be phased */
    std::function<void(int)> end;
};
int run_lanes(int n_devices, int lanes, int active_lanes, int64_t n_batches, const LaneHooks &hk, bool own_thread_for_lane0,
              const std::function<void(int)> &on_thread_start) {
    const int n_workers = n_devices * lanes;
    /* The FIRST batch of a lane is fixed: batch (lane of device) * n_devices + device, so the largest batches start on different
     * devices and no lane that starts early can take (and, one ahead, reserve) the work of a device whose thread is not up yet --
     * with as many batches as devices every device gets exactly one.  From there on the hand-out is dynamic. */
    std::atomic<int64_t> next{std::min<int64_t>(n_batches, (int64_t) n_devices * active_lanes)};
    std::atomic<int> status{MRP_OK};
    auto fail = [&](int rc) { int expect = MRP_OK; status.compare_exchange_strong(expect, rc); };
    auto work = [&](int w) {
        if (on_thread_start) on_thread_start(w);
        if (w % lanes >= active_lanes) return; /* (large caller-chosen batches: fewer calls of a device in flight) */
        const int64_t first = (int64_t) (w % lanes) * n_devices + w / lanes;
        if (first >= n_batches) return;
        int rc = hk.begin ? hk.begin(w) : MRP_OK;
        if (rc != MRP_OK) { fail(rc); return; }
        rc = hk.prefetch ? hk.prefetch(w, first) : MRP_OK;
        int64_t cur = first;
        while (cur >= 0) {
            if (rc != MRP_OK) { fail(rc); if (hk.discard) hk.discard(w, cur); break; }
            if (status.load() != MRP_OK) { if (hk.discard) hk.discard(w, cur); break; }
            const int64_t nb = next.fetch_add(1);
            int prc = MRP_OK;
            std::thread stager;
            bool inline_stage = false;
            if (nb < n_batches && hk.prefetch) {
                try { stager = std::thread([&, nb] { prc = hk.prefetch(w, nb); }); }
                catch (const std::system_error &) { inline_stage = true; } /* no thread to be had: stage after the call instead */
            }
            rc = hk.phase(w, cur);
            if (stager.joinable()) stager.join();
            if (hk.retire) hk.retire(w, cur); /* (what the batch held goes back with no other thread of the lane at work) */
            if (inline_stage) prc = hk.prefetch(w, nb);
            if (rc != MRP_OK) { fail(rc); if (nb < n_batches && hk.discard) hk.discard(w, nb); break; }
            cur = nb < n_batches ? nb : -1;
            rc = prc;
        }
        if (hk.end) hk.end(w);
    };
    std::vector<std::thread> th;
    std::vector<int> inline_lanes; /* lanes whose thread could not be created run on the caller's thread, after lane 0 */
    for (int w = 1; w < n_workers; w++) {
        try { th.emplace_back(work, w); }
        catch (const std::system_error &) { inline_lanes.push_back(w); }
    }
    if (own_thread_for_lane0) {
        bool started = false;
        try { std::thread t0(work, 0); started = true; t0.join(); }
        catch (const std::system_error &) { if (!started) work(0); }
    } else work(0);
    for (int w : inline_lanes) work(w);
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

int mrp_queue_dry_run(int32_t n_devices, int32_t lanes, int64_t n_chunks, const int64_t *cost, int64_t chunks_per_batch, double usec_per_cost,
                      int32_t *worker_of_chunk_out, int64_t *sequence_out) {
    if (n_devices < 1 || n_devices > MRP_MAX_QUEUE_DEVICES || lanes < 1 || lanes > 4 || n_chunks < 0 || (n_chunks > 0 && (!cost || !worker_of_chunk_out)))
        return mrp_set_error(MRP_ERR_ARG, "mrp_queue_dry_run: bad arguments");
    /* the plan and the hand-out of mrp_queue_phase_chunks for the same devices and lanes (worker = device * lanes + lane); the
     * stand-in for a call sleeps in proportion to the batch's cost, the stand-in for the upload of the next batch does nothing */
    const QueuePlan p = plan_queue(n_chunks, cost, chunks_per_batch, n_devices * lanes, n_devices);
    const int64_t n_batches = (int64_t) p.batch_off.size() - 1;
    std::atomic<int64_t> seq{0};
    LaneHooks hk;
    hk.phase = [&](int w, int64_t b) {
        int64_t c = 0;
        for (int64_t i = p.batch_off[(size_t) b]; i < p.batch_off[(size_t) b + 1]; i++) {
            const int64_t chunk = p.order[(size_t) i];
            worker_of_chunk_out[chunk] = w;
            if (sequence_out) sequence_out[chunk] = seq.fetch_add(1);
            c += cost[chunk];
        }
        if (usec_per_cost > 0) std::this_thread::sleep_for(std::chrono::microseconds((int64_t) (usec_per_cost * (double) c)));
        return (int) MRP_OK;
    };
    return run_lanes(n_devices, lanes, active_lanes_of(p, cost, n_devices, lanes), n_batches, hk, false, nullptr);
}

struct mrp_queue {
    std::vector<int32_t> devices;
    /* per lane (up to four per device): the context its batches are phased on, a second context on the same device whose
     * stream carries the uploads of the NEXT batch while the current one is phased, and its own host worker pool -- all
     * created by the worker on its first batch and kept between calls */
    std::vector<mrp_context *> ctx, stage_ctx; /* [device * lanes + lane] */
    std::vector<mrp_host_pool *> pools;        /* [device] */
    std::vector<mrp_chunk_block *> blocks;     /* [(device * lanes + lane) * 2 + parity] storage of a lane's current / next batch */
    int threads_per_device = 0;
    /* Calls of a device in flight at a time ("lanes", each a host thread with its own contexts); see plan_queue.  A short
     * queue uses one of them.  MRP_QUEUE_LANES=1..4 overrides. */
    int lanes = 4;
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
    if (const char *le = getenv("MRP_QUEUE_LANES")) { const int v = atoi(le); if (v >= 1 && v <= 4) q->lanes = v; }
    q->ctx.assign((size_t) n_devices * (size_t) q->lanes, nullptr);
    q->stage_ctx.assign((size_t) n_devices * (size_t) q->lanes, nullptr);
    q->pools.assign((size_t) n_devices, nullptr);
    q->blocks.assign((size_t) n_devices * (size_t) q->lanes * 2, nullptr);
    q->threads_per_device = mrp_queue_threads_per_device(n_devices);
    *out = q;
    return MRP_OK;
}

void mrp_queue_destroy(mrp_queue *q) {
    if (!q) return;
    for (size_t i = 0; i < q->blocks.size(); i++)
        if (q->blocks[i]) { (void) hipSetDevice(q->devices[i / (2 * (size_t) q->lanes)]); delete q->blocks[i]; }
    for (auto *c : q->stage_ctx) mrp_context_destroy(c);
    for (auto *c : q->ctx) mrp_context_destroy(c);
    for (auto *p : q->pools) mrp_host_pool_destroy(p);
    delete q;
}

}  /* extern "C" */

/* Host threads a worker gets: every device its own pool (phase.c:276-279 uses every core of the machine; one pool of sixteen
 * threads shared by eight devices would starve them).  mrp_set_host_threads() is the number PER DEVICE; by default the
 * machine's hardware threads are split evenly, at most 32 (16 through round 4) and at least 2 each. */
int mrp_queue_threads_per_device(int n_devices) {
    const int hw = (int) std::max(1u, std::thread::hardware_concurrency());
    const int asked = mrp_host_threads_setting();
    if (asked > 0) return asked;
    return std::max(2, std::min(32, hw / std::max(1, n_devices)));
}

/* The CPUs next to a device (Linux: /sys/bus/pci/devices/<bus id>/local_cpulist), for the worker of that device and every
 * thread it starts.  Only when a queue drives more than one device, and unless MRP_QUEUE_AFFINITY=0: a worker that reaches
 * across sockets for every staging copy and descriptor loses a third of its host bandwidth. */
static bool device_cpuset(int device, cpu_set_t *set) {
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int) sizeof(bus) - 1, device) != hipSuccess) return false;
    for (char *c = bus; *c; c++) *c = (char) tolower((unsigned char) *c);
    const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/local_cpulist";
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return false;
    char line[4096] = {0};
    const bool got = fgets(line, sizeof(line), f) != nullptr;
    fclose(f);
    if (!got) return false;
    CPU_ZERO(set);
    int n = 0;
    for (char *tok = strtok(line, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
        int lo = 0, hi = 0;
        if (sscanf(tok, "%d-%d", &lo, &hi) == 2) { for (int c = lo; c <= hi && c < CPU_SETSIZE; c++) { CPU_SET(c, set); n++; } }
        else if (sscanf(tok, "%d", &lo) == 1 && lo < CPU_SETSIZE) { CPU_SET(lo, set); n++; }
    }
    return n > 0;
}

extern "C" {

int mrp_queue_phase_chunks(mrp_queue *q, int64_t n_chunks, const mrp_chunk_desc *chunks, const mrp_params *params, int64_t chunks_per_batch,
                           mrp_phase_result **out, mrp_queue_stats *stats) {
    if (!q || n_chunks < 0 || !params || (n_chunks > 0 && (!chunks || !out))) return mrp_set_error(MRP_ERR_ARG, "mrp_queue_phase_chunks: bad arguments");
    const int n_devices = (int) q->devices.size();
    const int32_t *devices = q->devices.data();
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
    const int lanes = q->lanes, n_workers = n_devices * lanes;
    const QueuePlan plan = plan_queue(n_chunks, cost.data(), chunks_per_batch, n_workers, n_devices);
    const int64_t n_batches = (int64_t) plan.batch_off.size() - 1;
    if (stats) stats->batches = n_batches;

    /* one error text per lane and ROLE: the call (phase) and the upload of the next batch (prefetch) run on two threads at a time */
    std::vector<std::string> errs((size_t) n_workers), stage_errs((size_t) n_workers);
    struct PerDev { int64_t chunks = 0, units = 0, fallback = 0; double busy_ms = 0, stage_wait_ms = 0; };
    std::vector<PerDev> per((size_t) n_workers);
    std::mutex pool_mu;
    const int active_lanes = active_lanes_of(plan, cost.data(), n_devices, lanes);
    const char *aff_env = getenv("MRP_QUEUE_AFFINITY");
    const bool bind = n_devices > 1 && !(aff_env && aff_env[0] == '0');

    /* the chunks of one batch on the device (uploads queued on the staging context's stream; the chunks carry the event that
     * ends the upload, the first device work that reads one waits for it) */
    struct Staged {
        std::atomic<int64_t> batch{-1}; /* (the call's thread looks its batch up while the stager fills the lane's OTHER slot) */
        int64_t first = 0, count = 0;
        std::vector<mrp_chunk *> dch;
    };
    struct Lane { Staged st[2]; int n_staged = 0; void *caller_pool = nullptr; };
    std::vector<Lane> lane_state((size_t) n_workers);
    const bool timing = getenv("MRP_TIMING") != nullptr;
    const auto t_call = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count(); };
    auto drop = [&](Staged *st) { /* (a thousand chunks of a one-call queue: on the lane's pool, not one after the other) */
        mrp_parallel_for((int64_t) st->dch.size(), 8, [&](int64_t i) { if (st->dch[(size_t) i]) mrp_chunk_destroy(st->dch[(size_t) i]); });
        st->dch.clear();
        st->batch = -1;
    };
    auto staged_of = [&](int w, int64_t b) -> Staged * {
        Lane &L = lane_state[(size_t) w];
        return L.st[0].batch.load() == b ? &L.st[0] : (L.st[1].batch.load() == b ? &L.st[1] : nullptr);
    };
    LaneHooks hk;
    hk.begin = [&](int w) {
        const int d = w / lanes; /* the device this lane works for */
        int r = MRP_OK;
        if (!q->ctx[(size_t) w]) r = mrp_context_create(devices[d], &q->ctx[(size_t) w]);
        if (r == MRP_OK && !q->stage_ctx[(size_t) w]) r = mrp_context_create(devices[d], &q->stage_ctx[(size_t) w]);
        if (r == MRP_OK) {
            std::lock_guard<std::mutex> lk(pool_mu);
            if (!q->pools[(size_t) d]) {
                q->pools[(size_t) d] = mrp_host_pool_create(q->threads_per_device);
                if (!q->pools[(size_t) d]) r = mrp_set_error(MRP_ERR_NOMEM, "out of host memory");
            }
        }
        if (r != MRP_OK) { errs[(size_t) w] = mrp_last_error(); return r; }
        if (lanes > 1) (void) mrp_context_set_grouped(q->ctx[(size_t) w], 1); /* the lanes of a device are concurrent batches of it */
        q->ctx[(size_t) w]->calls_sharing_device = n_batches > (int64_t) n_devices ? active_lanes : 1;
        lane_state[(size_t) w].caller_pool = mrp_pool_current(); /* (lane 0 may run on the caller's thread: its pool comes back at the end) */
        mrp_pool_adopt(q->pools[(size_t) d]);
        /* the lanes of a device share it: a call of a long queue runs 8 / lanes concurrent batches, the one call of a short
         * queue as many as its size asks for */
        (void) mrp_context_set_phase_groups(q->ctx[(size_t) w], n_batches > (int64_t) n_devices ? std::max(1, 8 / active_lanes) : 0);
        return (int) MRP_OK;
    };
    hk.prefetch = [&](int w, int64_t b) { /* (the first batch on the lane's thread, the later ones on a thread beside the call) */
        const auto ts0 = std::chrono::steady_clock::now();
        const int d = w / lanes;
        mrp_pool_adopt(q->pools[(size_t) d]);
        Lane &L = lane_state[(size_t) w];
        const int parity = L.n_staged++ & 1;
        Staged *st = &L.st[parity];
        st->first = plan.batch_off[(size_t) b]; st->count = plan.batch_off[(size_t) b + 1] - st->first;
        st->dch.assign((size_t) st->count, nullptr);
        st->batch.store(b);
        mrp_context *sc = q->stage_ctx[(size_t) w];
        sc->pool.reclaim(); /* the block of the batch before last was emptied after its call returned */
        mrp_chunk_block *&blk = q->blocks[(size_t) w * 2 + (size_t) parity];
        if (!blk) blk = new (std::nothrow) mrp_chunk_block();
        std::vector<const mrp_chunk_desc *> dl((size_t) st->count);
        for (int64_t i = 0; i < st->count; i++) dl[(size_t) i] = &chunks[plan.order[(size_t) (st->first + i)]];
        /* (uploaded in the groups the call will deal the chunks to: its first batch starts on the device when an eighth of the bytes is there) */
        const int rc = blk ? mrp_chunk_block_create(sc, st->count, dl.data(), st->dch.data(), blk, mrp_phase_groups_for(q->ctx[(size_t) w], st->count))
                           : mrp_set_error(MRP_ERR_NOMEM, "out of host memory");
        if (rc != MRP_OK) stage_errs[(size_t) w] = mrp_last_error();
        if (timing) fprintf(stderr, "  [%7.1f] queue lane %d: batch %lld (%lld chunks) staged in %.1f ms\n", since(), w, (long long) b, (long long) st->count,
                            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ts0).count());
        return rc;
    };
    hk.discard = [&](int w, int64_t b) { if (Staged *st = staged_of(w, b)) drop(st); };
    hk.phase = [&](int w, int64_t b) {
        Staged *cur = staged_of(w, b);
        if (!cur) { errs[(size_t) w] = "work queue: batch was not staged"; return (int) MRP_ERR_ARG; }
        auto t0 = std::chrono::steady_clock::now();
        const int64_t count = cur->count;
        std::vector<const mrp_chunk *> cch((size_t) count);
        std::vector<const mrp_read *> rd((size_t) count);
        std::vector<int64_t> nr((size_t) count);
        std::vector<mrp_phase_result *> res((size_t) count, nullptr);
        for (int64_t i = 0; i < count; i++) {
            const mrp_chunk_desc &c = chunks[plan.order[(size_t) (cur->first + i)]];
            cch[(size_t) i] = cur->dch[(size_t) i]; rd[(size_t) i] = c.reads; nr[(size_t) i] = c.n_reads;
        }
        mrp_phase_many_stats ps{};
        const int r = mrp_phase_reads_many(q->ctx[(size_t) w], count, cch.data(), rd.data(), nr.data(), params, res.data(), &ps);
        if (r != MRP_OK) errs[(size_t) w] = mrp_last_error();
        for (int64_t i = 0; i < count; i++) {
            const int64_t chunk = plan.order[(size_t) (cur->first + i)];
            if (r == MRP_OK) { out[chunk] = res[(size_t) i]; per[(size_t) w].units += cost[(size_t) chunk]; }
        }
        if (r == MRP_OK) { per[(size_t) w].chunks += count; per[(size_t) w].fallback += ps.resident ? ps.fallback_chunks : count; }
        if (timing) fprintf(stderr, "  [%7.1f] queue lane %d: batch %lld phased in %.1f ms\n", since(), w, (long long) b,
                            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        per[(size_t) w].busy_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return r;
    };
    /* the batch's chunks live on the lane's STAGING context, on which the stager thread makes the next batch's block while the call
     * runs: they are destroyed after that thread has ended (one host thread per context, include/margin_rphmm.h) */
    hk.retire = [&](int w, int64_t b) {
        const auto t0 = std::chrono::steady_clock::now();
        if (Staged *st = staged_of(w, b)) drop(st);
        per[(size_t) w].busy_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    hk.end = [&](int w) { mrp_pool_adopt(lane_state[(size_t) w].caller_pool); };
    /* before the first thread of a worker is created (they inherit it): the CPUs next to its device */
    std::function<void(int)> on_start;
    if (bind) on_start = [&](int w) {
        cpu_set_t set;
        if (device_cpuset(devices[w / lanes], &set)) (void) pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
    };
    int rc = run_lanes(n_devices, lanes, active_lanes, n_batches, hk, /* the caller's own affinity is left alone */ bind, on_start);
    if (timing) fprintf(stderr, "  [%7.1f] queue: workers joined\n", since());
    if (stats)
        for (int w = 0; w < n_workers; w++) {
            const int d = w / lanes;
            stats->chunks_per_device[d] += per[(size_t) w].chunks;
            stats->units_per_device[d] += per[(size_t) w].units;
            stats->busy_ms_per_device[d] = std::max(stats->busy_ms_per_device[d], per[(size_t) w].busy_ms);
            stats->fallback_chunks += per[(size_t) w].fallback;
        }
    if (rc != MRP_OK) {
        for (int64_t i = 0; i < n_chunks; i++) { mrp_phase_result_destroy(out[i]); out[i] = nullptr; }
        for (auto &e : errs)
            if (!e.empty()) return mrp_set_error(rc, "%s", e.c_str());
        for (auto &e : stage_errs)
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
