/*
 * mrp_internal.h -- host-side objects behind the opaque handles of include/margin_rphmm.h, shared by
 * mrp_api.cpp (forward/backward seam) and mrp_engine.cpp (device-resident merge levels).
 */
#ifndef MRP_INTERNAL_H_
#define MRP_INTERNAL_H_

#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "../../include/margin_rphmm.h"
#include "mrp_device.h"
#include "mrp_kernels.h"
#include "rphmm_host.h"

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void) hipFree(p);
        p = nullptr;
        n = 0;
    }
    hipError_t alloc(size_t count) {
        release();
        n = count;
        return hipMalloc((void **) &p, std::max<size_t>(count, 1) * sizeof(T) + 64);
    }
    hipError_t upload(const std::vector<T> &h, hipStream_t s) {
        hipError_t e = alloc(h.size());
        if (e != hipSuccess || h.empty()) return e;
        return hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s);
    }
};


struct mrp_context {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t aux[2] = {nullptr, nullptr}; /* size classes of the recursion kernel run side by side */
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t fork = nullptr, join[2] = {nullptr, nullptr};
};

struct mrp_chunk {
    mrp_context *ctx = nullptr;
    int64_t n_sites = 0, pool_bytes = 0;
    std::vector<uint32_t> allele_number, allele_offset, sub_offset;
    std::vector<uint16_t> sub, prior; /* host copies for the structural code (rphmm_host.c) */
    std::vector<uint8_t> pool;
    uint32_t max_sub = 0, max_prior = 0;
    DevBuf<uint32_t> d_allele_number, d_allele_offset, d_sub_offset;
    DevBuf<uint16_t> d_sub, d_prior;
    DevBuf<uint8_t> d_pool;
    DevChunk dev{};
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
    std::vector<DevHmm> hmms;
    std::vector<DevCol> cols;
    std::vector<int64_t> read_byte_off;
    std::vector<uint64_t> partition;
    std::vector<SweepCol> scols;
    std::vector<PlaneCol> pcols;
    std::vector<uint32_t> cell_next, cell_prev, cell_np;
    std::vector<EmitTile> tiles;
    int64_t n_fast_tiles = 0;
    bool need_wide = false;
    std::vector<JobOut> outs;
    int64_t n_merge = 0, n_slots = 0;
    int64_t n_cells_total = 0; /* cells incl. alignment padding */
    bool resident = false;     /* cell arrays are produced on the device */
    mrp_launch_stats stats{};
    /* launch plan */
    std::vector<int32_t> order_wide, order_mid, order_narrow, order_f64;
    int max_merge_wide = 1, max_merge_mid = 1, max_merge_narrow = 1;
    /* device */
    bool uploaded = false, launched = false;
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
    DevBuf<int32_t> d_order_wide, d_order_mid, d_order_narrow, d_order_f64;
    DevBuf<EmitTile> d_tiles;
    MrpBatchDev dev{};
};


/* appends one hmm to a batch.  resident = the cell arrays (partition, transitions) are produced on
 * the device (mrp_engine.cpp): only the column structure is taken from the job. */
int mrp_batch_add_impl(mrp_batch *b, const mrp_hmm_job *job, bool resident, int64_t *cell0_out, int64_t *mcell0_out,
                       int64_t *col0_out);

#endif
