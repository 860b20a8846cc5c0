/* mrp_kernels.h -- launch wrappers of the gfx950 kernels (definitions in mrp_kernels.hip). */
#ifndef MRP_KERNELS_H_
#define MRP_KERNELS_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mrp_device.h"

struct MrpBatchDev {
    /* inputs */
    const DevHmm *hmms;
    const DevCol *cols;
    const SweepCol *scols;
    const PlaneCol *pcols;
    const DevChunk *chunks;
    const int64_t *read_byte_off;
    const int32_t *pack_list;  /* columns of the uniform-allele fast path: packed bytes only (mrp_pack_kernel) */
    const int32_t *plane_list; /* columns that also need the bit planes (general emission path) */
    int64_t n_pack_list, n_plane_list;
    int32_t list_filter;       /* 1 (levels described on the device, no lists): both kernels scan every column, the pack kernel
                                * takes those without need_planes, the bit-plane kernel those with it; n_pack_list / n_plane_list
                                * are then 0 ("no such column") or the number of columns */
    int32_t pad_filter;
    const uint64_t *partition;
    const uint32_t *cell_np;   /* next | prev << 16 */
    const uint32_t *cell_next; /* only when some hmm has > 65535 merge cells in a column */
    const uint32_t *cell_prev;
    /* scratch */
    uint64_t *planes;
    uint32_t *slot_total;
    uint32_t *slot_bytes; /* [n_slots * 16] read-major packed profile bytes (4 reads per word) */
    uint32_t *cell_cost;
    /* outputs: int32 for hmms swept by the max-plus kernel, fp64 for the log-sum-exp kernel */
    int32_t *cell_f32;
    int32_t *cell_b32;
    int32_t *merge_f32;
    int32_t *merge_b32;
    double *cell_f;
    double *cell_b;
    double *merge_f;
    double *merge_b;
    double *col_total;
    double *hmm_fb; /* [2 * n_hmms] forward, backward */
    int64_t n_hmms, n_cols, n_cells, n_merge, n_slots;
};

/* usable dynamic LDS per workgroup for the sweep kernels, bytes */
#define MRP_LDS_BUDGET (160 * 1024 - 1024)

hipError_t mrp_launch_planes(const MrpBatchDev &d, hipStream_t stream);
/* MRP_EMIT_TILE cells per tile */
#define MRP_EMIT_TILE 512
/* workgroups of the grid-striding kernels: 256 CUs x 8 workgroups of 256 threads */
#define MRP_PERSISTENT_GRID 2048
/* tiles_dev[0..n_fast) take the uniform-allele fast path, the next n_general the general path */
/* the EmitTile array of a resident batch from its column descriptors (one thread per column) */
hipError_t mrp_launch_tiles(const DevCol *cols_dev, const TileCol *tilecols_dev, int64_t n_cols, EmitTile *tiles_dev, hipStream_t stream);
hipError_t mrp_launch_emission(const MrpBatchDev &d, const EmitTile *tiles_dev, int64_t n_fast, int64_t n_general,
                               hipStream_t stream);
/* order[0..n) = indices into d.hmms handled by this launch, one workgroup each */
hipError_t mrp_launch_sweep_i32(const MrpBatchDev &d, const int32_t *order_dev, int64_t n, int block_threads,
                                int max_merge, hipStream_t stream);
hipError_t mrp_launch_sweep_f64(const MrpBatchDev &d, const int32_t *order_dev, int64_t n, int block_threads,
                                hipStream_t stream);
/* sum mode with the merge column in LDS: reproducible log-sum-exp (integer atomics), at most MRP_LSE_MAX_MERGE merge cells
 * per merge column, 16-bit transitions */
#define MRP_LSE_CUR_LDS_MAX_MERGE 8000 /* 20 B of LDS per merge cell up to here, 12 B above (finished values read back from HBM) */
#define MRP_LSE_MAX_MERGE 13600
#define MRP_LSE_MAX_TERMS 16384          /* cells per column: 2^14 terms of at most 2^50 units each fit the 64-bit accumulator */
#define MRP_LSE_MAX_COST (1ll << 27)     /* bound of |log p|: the float reference points of the kernel resolve 8 up to here */
hipError_t mrp_launch_sweep_lse(const MrpBatchDev &d, const int32_t *order_dev, int64_t n, int max_merge, hipStream_t stream);
hipError_t mrp_launch_fill_f64(double *p, int64_t n, double v, hipStream_t stream);
hipError_t mrp_launch_emissions(const DevCol *col_dev, const DevChunk *chunks, const uint64_t *planes,
                                const uint32_t *slot_total, uint32_t flags, int64_t n_cells,
                                const uint64_t *partitions, double *out, hipStream_t stream);

#endif
