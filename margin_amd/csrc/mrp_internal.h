/*
 * mrp_internal.h -- host-side objects behind the opaque handles of include/margin_rphmm.h, shared by
 * mrp_api.cpp (forward/backward seam) and mrp_engine.cpp (device-resident merge levels).
 */
#ifndef MRP_INTERNAL_H_
#define MRP_INTERNAL_H_

#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdlib>

#include <time.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <functional>
#include <iterator>
#include <map>
#include <memory>
#include <new>
#include <type_traits>
#include <mutex>
#include <utility>
#include <vector>

#include "../../include/margin_rphmm.h"
#include "mrp_device.h"
#include "mrp_kernels.h"
#include "rphmm_host.h"

/* A kernel attribute (the opt-in to more than 64 KB of dynamic LDS) belongs to the function ON A DEVICE: set once per device,
 * by whichever context of that device launches first (the C-ABI lets one process hold contexts on several devices). */
struct PerDeviceOnce {
    std::mutex mu;
    bool done[64] = {false};
    template <class F> hipError_t run(F f) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev < 0 || dev >= 64) return f();
        std::lock_guard<std::mutex> lock(mu);
        if (done[dev]) return hipSuccess;
        e = f();
        if (e == hipSuccess) done[dev] = true;
        return e;
    }
};

/* std::vector whose resize() leaves trivially constructible elements uninitialised: the descriptor arrays of a level
 * are tens of megabytes that the filling threads overwrite entirely; zero-filling them first is a serial pass. */
template <class T>
struct DefaultInitAllocator : std::allocator<T> {
    template <class U> struct rebind { using other = DefaultInitAllocator<U>; };
    using std::allocator<T>::allocator;
    template <class U> void construct(U *p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void *>(p)) U; }
    template <class U, class... Args> void construct(U *p, Args &&...args) { ::new (static_cast<void *>(p)) U(std::forward<Args>(args)...); }
};
template <class T> using HostVec = std::vector<T, DefaultInitAllocator<T>>;

/* Caching device allocator of a context.  hipMalloc / hipFree of multi-GB arrays cost hundreds of
 * milliseconds and hipFree synchronizes the device, so blocks are kept and handed out again by size
 * class (1/8-octave rounding).  A released block becomes reusable only at the next reclaim(), which the
 * owners call right after they synchronized the context's stream: nothing in flight can still touch it. */
struct DevPool;
/* Every pool of the process, by device: what the pools of a device hold together (live and cached) is kept below the
 * device's memory less a head room (the runtime allocates scratch and queue memory of its own and ABORTS when it cannot),
 * and a pool the driver refuses a block gives back what idles in all of them, whoever owns them (the halves of one call,
 * a work queue's lanes, a caller's other contexts). */
struct DevPoolRegistry {
    static constexpr int MAX_DEVICES = 64;
    std::mutex mu;
    std::vector<DevPool *> pools;
    std::atomic<size_t> held[MAX_DEVICES];   /* bytes the pools of a device got from hipMalloc and have not given back */
    std::atomic<size_t> budget[MAX_DEVICES]; /* 0: not asked yet */
    DevPoolRegistry() { for (int d = 0; d < MAX_DEVICES; d++) { held[d].store(0); budget[d].store(0); oom_events[d].store(0); inject_oom[d].store(0); } }
    static DevPoolRegistry &get() { static DevPoolRegistry r; return r; }
    std::atomic<int> inject_oom[MAX_DEVICES];      /* test hook (mrp_context_set_test_hooks bit 2): refuse the next large allocation */
    std::atomic<uint64_t> oom_events[MAX_DEVICES]; /* allocations the driver refused for good (after every cache was given back) */
    /* What the pools of a device may hold together: the memory that is FREE when the first pool of the process asks (another
     * process, or allocations of the caller's own, are not ours to count on) less a head room of a sixteenth of the device,
     * and never more than the device less an eighth.  MRP_POOL_BUDGET_MB overrides. */
    size_t budget_of(int device) {
        size_t b = budget[device].load(std::memory_order_relaxed);
        if (b == 0) {
            size_t total = 0, free_now = 0, tot2 = 0; /* (by ordinal: the calling thread's current device may be another one) */
            b = hipDeviceTotalMem(&total, device) == hipSuccess && total > 0 ? total - total / 8 : ~(size_t) 0;
            int cur = -1;
            if (total > 0 && hipGetDevice(&cur) == hipSuccess && hipSetDevice(device) == hipSuccess) {
                if (hipMemGetInfo(&free_now, &tot2) == hipSuccess && free_now > 0) {
                    const size_t ours = held[device].load(std::memory_order_relaxed), room = total / 16;
                    const size_t avail = free_now + ours > room ? free_now + ours - room : free_now + ours;
                    if (avail < b) b = avail;
                }
                (void) hipSetDevice(cur);
            }
            (void) hipGetLastError();
            if (const char *e = getenv("MRP_POOL_BUDGET_MB")) { const long long v = atoll(e); if (v > 0) b = (size_t) v << 20; }
            budget[device].store(b, std::memory_order_relaxed);
        }
        return b;
    }
    /* the budget is asked for again at its next use (mrp_context_trim: the caches have just been given back, so what is free now
     * is what another tenant's transient pressure -- the reason a budget was shrunk -- has left or given back) */
    void forget_budget(int device) { if (device >= 0 && device < MAX_DEVICES) budget[device].store(0, std::memory_order_relaxed); }
    /* after the driver refused an allocation although nothing idles in any pool: what we hold now is what there is */
    void shrink_budget(int device) {
        const size_t ours = held[device].load(std::memory_order_relaxed);
        size_t b = budget[device].load(std::memory_order_relaxed);
        if (ours > ((size_t) 1 << 30) && (b == 0 || ours < b)) budget[device].store(ours, std::memory_order_relaxed);
    }
    inline void trim_device(int device, DevPool *but);
};

struct DevPool {
    std::mutex mu;
    std::multimap<size_t, void *> free_blocks;
    std::vector<std::pair<size_t, void *>> pending;
    size_t cached_bytes = 0;                 /* bytes in free_blocks */
    size_t cache_limit = (size_t) 64 << 30;  /* beyond this the largest cached blocks go back to the driver; what really bounds the
                                              * caches is the device's budget (registry): a batch of a call keeps some 170 MB
                                              * per 1 Mb chunk of its widest level for reuse (576 chunks in eight batches: 107 GB
                                              * held, 96 of them idle between calls); a limit of 24 GB per pool made every call
                                              * with more than ~100 chunks per batch give back and re-allocate (4x slower) */
    int device = -1;                         /* attach(): the device whose registry entry this pool is */
    std::atomic<uint64_t> oom_local{0};      /* allocations the driver refused THIS pool for good: a call looks at the pools of its own contexts,
                                              * not at the device-wide count (another lane's refusal is not this call's) */
    void attach(int dev) {
        device = dev >= 0 && dev < DevPoolRegistry::MAX_DEVICES ? dev : -1;
        if (device < 0) return;
        DevPoolRegistry &r = DevPoolRegistry::get();
        std::lock_guard<std::mutex> lock(r.mu);
        r.pools.push_back(this);
    }
    void detach() {
        if (device < 0) return;
        DevPoolRegistry &r = DevPoolRegistry::get();
        std::lock_guard<std::mutex> lock(r.mu);
        for (size_t i = 0; i < r.pools.size(); i++)
            if (r.pools[i] == this) { r.pools.erase(r.pools.begin() + (long) i); break; }
        device = -1;
    }
    void held_add(size_t b) { if (device >= 0) DevPoolRegistry::get().held[device].fetch_add(b, std::memory_order_relaxed); }
    void held_sub(size_t b) { if (device >= 0) DevPoolRegistry::get().held[device].fetch_sub(b, std::memory_order_relaxed); }
    bool over_budget() const {
        if (device < 0) return false;
        DevPoolRegistry &r = DevPoolRegistry::get();
        return r.held[device].load(std::memory_order_relaxed) > r.budget_of(device);
    }
    static size_t size_class(size_t bytes) {
        if (bytes < 256) bytes = 256;
        const int lg = 63 - __builtin_clzll((unsigned long long) bytes);
        const size_t step = (size_t) 1 << (lg > 3 ? lg - 3 : 0);
        return (bytes + step - 1) & ~(step - 1);
    }
    hipError_t alloc(void **p, size_t bytes, size_t *cls_out) {
        const size_t cls = size_class(bytes);
        *cls_out = cls;
        {
            std::lock_guard<std::mutex> lock(mu);
            /* best fit: the smallest idle block of this class or up to half as large again (the batches of a work queue differ
             * by some per cent from call to call: exact classes alone would miss, and every miss is a hipMalloc of a gigabyte) */
            auto it = free_blocks.lower_bound(cls);
            if (it != free_blocks.end() && it->first <= cls + cls / 2) {
                *p = it->second;
                *cls_out = it->first;
                cached_bytes -= it->first;
                free_blocks.erase(it);
                return hipSuccess;
            }
        }
        if (device >= 0 && DevPoolRegistry::get().held[device].load(std::memory_order_relaxed) + cls > DevPoolRegistry::get().budget_of(device)) {
            trim(); /* the device's budget: first what idles here, then what idles in the other pools */
            if (DevPoolRegistry::get().held[device].load(std::memory_order_relaxed) + cls > DevPoolRegistry::get().budget_of(device))
                DevPoolRegistry::get().trim_device(device, this);
        }
        if (device >= 0 && cls >= ((size_t) 1 << 20) && DevPoolRegistry::get().inject_oom[device].load(std::memory_order_relaxed) > 0 &&
            DevPoolRegistry::get().inject_oom[device].exchange(0) > 0) { /* fault injection of the test suite */
            DevPoolRegistry::get().oom_events[device].fetch_add(1, std::memory_order_relaxed);
            oom_local.fetch_add(1, std::memory_order_relaxed);
            return hipErrorOutOfMemory;
        }
        hipError_t e = hipMalloc(p, cls);
        if (e != hipSuccess) { /* give the cache back and try once more */
            (void) hipGetLastError();
            trim();
            e = hipMalloc(p, cls);
        }
        if (e != hipSuccess && device >= 0) { /* ... and what idles in the other contexts of this device */
            (void) hipGetLastError();
            DevPoolRegistry::get().trim_device(device, this);
            e = hipMalloc(p, cls);
        }
        if (e == hipSuccess) held_add(cls);
        else if (e == hipErrorOutOfMemory && device >= 0) { /* callers that can re-slice their work look at this count */
            DevPoolRegistry::get().oom_events[device].fetch_add(1, std::memory_order_relaxed);
            oom_local.fetch_add(1, std::memory_order_relaxed);
            DevPoolRegistry::get().shrink_budget(device);
        }
        return e;
    }
    void release(void *p, size_t cls) {
        std::lock_guard<std::mutex> lock(mu);
        pending.emplace_back(cls, p);
    }
    void reclaim() { /* only after the context's streams were synchronized */
        std::lock_guard<std::mutex> lock(mu);
        for (auto &b : pending) {
            free_blocks.emplace(b.first, b.second);
            cached_bytes += b.first;
        }
        pending.clear();
        while ((cached_bytes > cache_limit || over_budget()) && !free_blocks.empty()) {
            auto it = std::prev(free_blocks.end());
            (void) hipFree(it->second);
            held_sub(it->first);
            cached_bytes -= it->first;
            free_blocks.erase(it);
        }
    }
    void trim() { /* frees the reusable blocks */
        std::lock_guard<std::mutex> lock(mu);
        for (auto &b : free_blocks) { (void) hipFree(b.second); held_sub(b.first); }
        free_blocks.clear();
        cached_bytes = 0;
    }
    void destroy() { /* context teardown: everything, after a device synchronize */
        reclaim();
        trim();
        detach();
    }
};

inline void DevPoolRegistry::trim_device(int device, DevPool *but) {
    std::lock_guard<std::mutex> lock(mu);
    for (DevPool *q : pools)
        if (q != but && q->device == device) q->trim();
}

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevPool *pool = nullptr; /* NULL: plain hipMalloc / hipFree */
    size_t cls = 0;
    ~DevBuf() { release(); }
    void release() {
        if (p) {
            if (pool) pool->release(p, cls);
            else (void) hipFree(p);
        }
        p = nullptr;
        n = 0;
    }
    hipError_t alloc(size_t count) {
        release();
        n = count;
        const size_t bytes = std::max<size_t>(count, 1) * sizeof(T) + 64;
        hipError_t e = pool ? pool->alloc((void **) &p, bytes, &cls) : hipMalloc((void **) &p, bytes);
#ifdef MRP_POISON_ALLOC /* debugging aid: every buffer starts as garbage, so a read of a never-written element shows */
        if (e == hipSuccess) {
            (void) hipDeviceSynchronize();
            e = hipMemset(p, 0xA5, bytes);
            (void) hipDeviceSynchronize();
        }
#endif
        return e;
    }
    template <class A>
    hipError_t upload(const std::vector<T, A> &h, hipStream_t s) {
        hipError_t e = alloc(h.size());
        if (e != hipSuccess || h.empty()) return e;
        return hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s);
    }
};

struct mrp_context {
    int device = 0;
    hipStream_t stream = nullptr;
    /* Streams beyond the main one are made when first asked for: the device's hardware queues (GPU_MAX_HW_QUEUES, 16 after
     * mrp_runtime_init) are dealt to streams in creation order, and every stream more than that shares a queue with another --
     * whose kernels it then waits for.  A context that is one of several concurrent batches (`grouped`: the siblings of
     * mrp_phase_reads_many, a work queue's contexts) runs its size classes on its main stream (2 streams a batch, 16 for the
     * eight batches of a call); a context on its own runs them side by side on two more. */
    hipStream_t aux[2] = {nullptr, nullptr};
    bool grouped = false;
    int concurrent_batches = 1; /* how many batches of a mrp_phase_reads_many call run side by side on the device (this context's is one of them) */
    int calls_sharing_device = 1; /* a work queue's lanes: this many calls run on the device at a time (set on the lane's context) */
    hipError_t side_streams(hipStream_t *a0, hipStream_t *a1) {
        if (grouped) { *a0 = *a1 = stream; return hipSuccess; }
        for (int i = 0; i < 2; i++)
            if (!aux[i]) { const hipError_t e = hipStreamCreateWithFlags(&aux[i], hipStreamNonBlocking); if (e != hipSuccess) return e; }
        *a0 = aux[0]; *a1 = aux[1];
        return hipSuccess;
    }
    hipError_t copy_stream(hipStream_t *cs) { /* uploads of a staged level of the resident engine, beside the kernels of the level before */
        if (!pre) { const hipError_t e = hipStreamCreateWithFlags(&pre, hipStreamNonBlocking); if (e != hipSuccess) return e; }
        *cs = pre;
        return hipSuccess;
    }
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t pre = nullptr; /* copy_stream() */
    hipEvent_t last_emission = nullptr; /* end of the emission kernel of the most recent launch on this context (owned by its batch) */
    hipEvent_t fork = nullptr, join[2] = {nullptr, nullptr};
    /* Waiting for a stream without burning a core: hipStreamSynchronize polls (a batch thread of mrp_phase_reads_many spent
     * 85 % of a call spinning in it); an event created with hipEventBlockingSync makes the waiter sleep until the
     * interrupt.  One event per context: a context is driven by one host thread at a time. */
    hipEvent_t block_ev = nullptr;
    double wait_cpu_ms = 0, wait_wall_ms = 0; /* MRP_TIMING: thread CPU / wall time spent inside wait_stream */
    hipError_t wait_stream(hipStream_t s) {
        timespec c0, c1, w0, w1;
        clock_gettime(CLOCK_THREAD_CPUTIME_ID, &c0); clock_gettime(CLOCK_MONOTONIC, &w0);
        const hipError_t e = wait_stream_impl(s);
        clock_gettime(CLOCK_THREAD_CPUTIME_ID, &c1); clock_gettime(CLOCK_MONOTONIC, &w1);
        wait_cpu_ms += 1e3 * (c1.tv_sec - c0.tv_sec) + 1e-6 * (c1.tv_nsec - c0.tv_nsec);
        wait_wall_ms += 1e3 * (w1.tv_sec - w0.tv_sec) + 1e-6 * (w1.tv_nsec - w0.tv_nsec);
        return e;
    }
    hipError_t wait_stream_impl(hipStream_t s) {
        if (!block_ev) {
            hipError_t e = hipEventCreateWithFlags(&block_ev, hipEventDisableTiming);
            if (e != hipSuccess) { block_ev = nullptr; return hipStreamSynchronize(s); }
        }
        hipError_t e = hipEventRecord(block_ev, s);
        if (e != hipSuccess) return e;
        /* hipEventSynchronize spins on this runtime even for an event created with hipEventBlockingSync (measured: 68 ms of
         * wall = 68 ms of thread CPU per batch thread and call): query and sleep instead -- a level waits milliseconds, 30 us
         * of extra latency per wait is nothing */
        for (;;) {
            e = hipEventQuery(block_ev);
            if (e != hipErrorNotReady) return e;
            timespec ts{0, 30000};
            nanosleep(&ts, nullptr);
        }
    }
    DevPool pool;
    int test_hooks = 0;   /* mrp_context_set_test_hooks (test suite only) */
    int phase_groups = 0; /* concurrent batches of mrp_phase_reads_many (mrp_context_set_phase_groups); 0: chosen by batch size */
    std::vector<mrp_context *> siblings; /* further contexts on the same device (concurrent batches of mrp_phase_reads_many) */
    std::mutex sibling_mu;
    /* the emptied batch object of the last resident engine on this context: its host arrays keep their capacity (and their
     * mapped pages) from one mrp_phase_reads_many call to the next */
    struct mrp_batch *spare_batch = nullptr;
    /* and its level objects (page-locked staging blocks, events) with a second emptied batch: kept from call to call, allocating
     * page-locked memory takes milliseconds and synchronizes the device (owned here, managed by mrp_engine.cpp) */
    std::vector<struct mrp_engine_level_state *> spare_levels;
    std::vector<struct mrp_batch *> spare_batches;
    /* page-locked host staging for the small per-level results of the resident engine (grow-only) */
    void *pinned = nullptr;
    size_t pinned_bytes = 0;
    hipError_t pinned_reserve(size_t bytes) {
        if (bytes <= pinned_bytes) return hipSuccess;
        if (pinned) (void) hipHostFree(pinned);
        pinned = nullptr;
        pinned_bytes = 0;
        const size_t want = std::max<size_t>(bytes + bytes / 2, (size_t) 1 << 20);
        hipError_t e = hipHostMalloc(&pinned, want, hipHostMallocDefault);
        if (e == hipSuccess) pinned_bytes = want;
        return e;
    }
};

struct mrp_chunk {
    mrp_context *ctx = nullptr;
    int64_t n_sites = 0, pool_bytes = 0;
    std::vector<uint32_t> allele_number, allele_offset, sub_offset;
    std::vector<int32_t> same_until; /* [n_sites] first site after i whose allele count differs from site i's (n_sites if none) */
    std::vector<uint16_t> sub, prior; /* host copies for the structural code (rphmm_host.c) */
    std::vector<uint8_t> pool;
    const uint8_t *pool_host = nullptr; /* the profile bytes on the host: pool.data(), or the copy in the page-locked block of a work queue's batch */
    uint32_t max_sub = 0, max_prior = 0, max_alleles = 1;
    DevBuf<uint32_t> d_allele_number, d_allele_offset, d_sub_offset;
    DevBuf<int32_t> d_same_until;
    DevBuf<uint16_t> d_sub, d_prior;
    DevBuf<uint8_t> d_pool;
    DevChunk dev{};
    /* a chunk whose uploads were queued but not waited for (a work queue uploads its next batch beside the current one): the
     * event ends them.  Device work that reads the chunk waits for it on its stream (mrp_engine.cpp) or on the host
     * (host_wait: the paths that upload on the default path). */
    mutable hipEvent_t ready = nullptr;
    bool owns_ready = true; /* false: the event belongs to the block the chunk was uploaded with (mrp_chunk_block) */
    mutable std::atomic<bool> ready_pending{false};
    hipError_t host_wait() const {
        if (!ready_pending.load()) return hipSuccess;
        const hipError_t e = hipEventSynchronize(ready);
        if (e == hipSuccess) ready_pending.store(false);
        return e;
    }
    ~mrp_chunk() {
        if (ready && owns_ready) { (void) hipEventSynchronize(ready); (void) hipEventDestroy(ready); }
    }
};

/* page-locked, grow-only host buffer: source of asynchronous uploads */
struct PinnedBuf {
    void *p = nullptr;
    size_t bytes = 0;
    ~PinnedBuf() { if (p) (void) hipHostFree(p); }
    hipError_t reserve(size_t want) {
        if (want <= bytes) return hipSuccess;
        if (p) (void) hipHostFree(p);
        p = nullptr; bytes = 0;
        const size_t sz = std::max<size_t>(want + want / 4, (size_t) 1 << 20);
        hipError_t e = hipHostMalloc(&p, sz, hipHostMallocDefault);
        if (e == hipSuccess) bytes = sz;
        return e;
    }
};

/* device + staging storage of the chunks of one batch of a work queue (mrp_chunk_block_create); outlives its chunks and
 * is reused for the batch after next */
struct mrp_chunk_block {
    PinnedBuf host;
    DevBuf<uint8_t> dev;
    hipEvent_t ready = nullptr;              /* end of the whole upload */
    std::vector<hipEvent_t> group_ready;     /* end of the upload of every group of chunks (the chunks of a group are contiguous in the block) */
    ~mrp_chunk_block() {
        if (ready) { (void) hipEventSynchronize(ready); (void) hipEventDestroy(ready); }
        for (hipEvent_t e : group_ready) if (e) (void) hipEventDestroy(e);
    }
};

struct JobOut {
    double *cell_f, *cell_b, *merge_f, *merge_b, *col_total, *hmm_f, *hmm_b;
    int64_t cell0, n_cells, mcell0, n_merge, col0, n_cols;
    bool int_path = false; /* swept by the max-plus int32 kernel (decided in mrp_batch_upload) */
};

struct mrp_batch {
    mrp_context *ctx = nullptr;
    std::mutex mu; /* mrp_batch_add may be called from several host threads (recording) */
    std::vector<const mrp_chunk *> chunks;
    HostVec<DevHmm> hmms;
    HostVec<DevCol> cols;
    HostVec<int64_t> read_byte_off;
    HostVec<uint64_t> partition;
    HostVec<SweepCol> scols;
    HostVec<PlaneCol> pcols;
    HostVec<uint32_t> cell_next, cell_prev, cell_np;
    HostVec<EmitTile> tiles;
    int64_t n_fast_tiles = 0;
    HostVec<TileCol> tilecols; /* resident batches: the tiles are written on the device from these */
    int64_t n_tiles_dev = 0;   /* their number (tiles stays empty) */
    /* resident merge levels: launched between the byte packing and the recursion kernels, in place of the emission kernel
     * (cross product + emission in one pass, mrp_launch_cross_emit) */
    std::function<hipError_t(hipStream_t)> pre_sweep;
    DevBuf<TileCol> d_tilecols;
    bool need_wide = false;
    std::vector<JobOut> outs;
    int64_t n_merge = 0, n_slots = 0;
    int64_t n_cells_total = 0; /* cells incl. alignment padding */
    bool resident = false;     /* cell arrays are produced on the device */
    mrp_launch_stats stats{};
    /* launch plan */
    std::vector<int32_t> order_wide, order_mid, order_narrow, order_f64, order_lse, order_lse_big, order_gen; /* order_f64 = order_lse + order_lse_big + order_gen */
    int max_merge_wide = 1, max_merge_mid = 1, max_merge_narrow = 1, max_merge_lse = 1, max_merge_lse_big = 1;
    /* device */
    bool uploaded = false, launched = false;
    /* events of the most recent launches: [0] planes start, [1] planes end, [2] sweeps end, [3] emission end, [4] emission start */
    static constexpr int EV_RING = 32;
    std::vector<std::array<hipEvent_t, 5>> ev_ring;
    int64_t n_launches = 0, stats_mark = 0; /* stats_mark: launches already reported by an earlier mrp_batch_stats */
    DevBuf<DevHmm> d_hmms;
    DevBuf<DevCol> d_cols;
    DevBuf<DevChunk> d_chunks;
    DevBuf<int64_t> d_read_byte_off;
    DevBuf<uint64_t> d_partition, d_planes;
    DevBuf<SweepCol> d_scols;
    DevBuf<PlaneCol> d_pcols;
    DevBuf<uint32_t> d_next, d_prev, d_np, d_slot_total, d_slot_bytes, d_cost;
    DevBuf<double> d_f, d_b, d_mf, d_mb, d_total, d_hmm_fb;
    DevBuf<int32_t> d_f32, d_b32, d_mf32, d_mb32;
    DevBuf<int32_t> d_order_wide, d_order_mid, d_order_narrow, d_order_f64, d_order_lse, d_order_lse_big, d_pack_list, d_plane_list;
    DevBuf<EmitTile> d_tiles;
    MrpBatchDev dev{};
    /* back to the empty state, keeping the capacity of the host arrays (the resident engine reuses one batch object for
     * all levels of a run: their descriptor arrays are tens of megabytes).  Only after the stream was synchronized. */
    void recycle() {
        d_hmms.release(); d_cols.release(); d_chunks.release(); d_read_byte_off.release(); d_partition.release(); d_planes.release();
        d_scols.release(); d_pcols.release(); d_next.release(); d_prev.release(); d_np.release(); d_slot_total.release();
        d_slot_bytes.release(); d_cost.release(); d_f.release(); d_b.release(); d_mf.release(); d_mb.release(); d_total.release();
        d_hmm_fb.release(); d_f32.release(); d_b32.release(); d_mf32.release(); d_mb32.release(); d_order_wide.release();
        d_order_mid.release(); d_order_narrow.release(); d_order_f64.release(); d_order_lse.release(); d_order_lse_big.release(); d_tiles.release(); d_pack_list.release(); d_plane_list.release(); d_tilecols.release();
        chunks.clear(); hmms.clear(); cols.clear(); read_byte_off.clear(); partition.clear(); scols.clear(); pcols.clear();
        cell_next.clear(); cell_prev.clear(); cell_np.clear(); tiles.clear(); tilecols.clear(); outs.clear();
        order_wide.clear(); order_mid.clear(); order_narrow.clear(); order_f64.clear(); order_lse.clear(); order_lse_big.clear(); order_gen.clear();
        n_fast_tiles = 0; n_tiles_dev = 0; need_wide = false; n_merge = 0; n_slots = 0; n_cells_total = 0; resident = false;
        stats = mrp_launch_stats{};
        max_merge_wide = max_merge_mid = max_merge_narrow = max_merge_lse = max_merge_lse_big = 1;
        uploaded = launched = false;
        n_launches = 0; stats_mark = 0; /* (the events stay: every launch of the emptied batch has been waited for) */
        dev = MrpBatchDev{};
    }
    void bind_pool(DevPool *pl) {
        d_hmms.pool = pl; d_cols.pool = pl; d_chunks.pool = pl; d_read_byte_off.pool = pl; d_partition.pool = pl; d_planes.pool = pl;
        d_scols.pool = pl; d_pcols.pool = pl; d_next.pool = pl; d_prev.pool = pl; d_np.pool = pl; d_slot_total.pool = pl;
        d_slot_bytes.pool = pl; d_cost.pool = pl; d_f.pool = pl; d_b.pool = pl; d_mf.pool = pl; d_mb.pool = pl; d_total.pool = pl;
        d_hmm_fb.pool = pl; d_f32.pool = pl; d_b32.pool = pl; d_mf32.pool = pl; d_mb32.pool = pl; d_order_wide.pool = pl;
        d_order_mid.pool = pl; d_order_narrow.pool = pl; d_order_f64.pool = pl; d_order_lse.pool = pl; d_order_lse_big.pool = pl; d_tiles.pool = pl; d_pack_list.pool = pl; d_plane_list.pool = pl; d_tilecols.pool = pl;
    }
};


/* appends one hmm to a batch.  resident = the cell arrays (partition, transitions) are produced on
 * the device (mrp_engine.cpp): only the column structure is taken from the job. */
int mrp_batch_add_impl(mrp_batch *b, const mrp_hmm_job *job, bool resident, int64_t *cell0_out, int64_t *mcell0_out,
                       int64_t *col0_out);


#include <atomic>
#include <thread>
extern "C" void mrp_pool_run(int64_t n, int64_t grain, void (*fn)(int64_t, void *), void *arg);
/* releases what mrp_engine.cpp parked in the context (mrp_context_destroy) */
void mrp_engine_release_context_cache(mrp_context *ctx);
/* groups > 1: chunk i belongs to group i % groups (the concurrent batches mrp_phase_reads_many will deal the chunks to); the block is
 * laid out and uploaded group by group, and a chunk is ready when its group's copy has ended -- the first batch's kernels need not wait
 * for the last batch's bytes */
int mrp_chunk_block_create(mrp_context *ctx, int64_t n, const mrp_chunk_desc *const *descs, mrp_chunk **out, mrp_chunk_block *blk, int groups = 1);
int mrp_host_threads_setting(void); /* what mrp_set_host_threads() was given, 0 if it was never called */
/* host worker pools (mrp_api.cpp) */
extern "C" void mrp_batch_last_launch_ms(struct mrp_batch *b, float *pack, float *emission, float *recursion);
struct mrp_host_pool;
mrp_host_pool *mrp_host_pool_create(int threads);
void mrp_host_pool_destroy(mrp_host_pool *p);
template <class F>
static inline void mrp_parallel_for(int64_t n, int64_t grain, F f) {
    mrp_pool_run(n, grain, [](int64_t i, void *a) { (*static_cast<F *>(a))(i); }, &f);
}


/* development: MRP_DUP=<letters> launches the named kernel families of a resident level TWICE (they are idempotent) -- the slow-down of a
 * step is that family's marginal cost in situ: p packing, x cross product + emission, s recursion, r prune, c compaction, l layout,
 * m the one-wave kernel of the small hmms, t the structure kernel, b the trace back */
static inline bool mrp_dup(char c) { const char *e = getenv("MRP_DUP"); return e && strchr(e, c) != nullptr; }
#endif
