/*
 * mrp_device.h -- structures shared by the host side of libmargin_rphmm.so and its gfx950 kernels.
 *
 * HBM layout of a batch (DESIGN.md "Data layout in HBM"): every array is the concatenation over
 * all HMMs of the batch, so a launch touches a handful of large, contiguous allocations:
 *
 *   per cell   (sum C) : partition u64 | next u32 | prev u32 | f f64 | b f64 | e u32 (scratch)
 *   per merge  (sum M) : mf f64 | mb f64
 *   per column (sum K) : DevCol (64 B, read through the scalar cache) | total f64
 *   per allele slot    : 8 bit planes u64 (emissions.c:91-123) | column-wide byte sum u32
 *   per hmm            : DevHmm | forward f64, backward f64
 */
#ifndef MRP_DEVICE_H_
#define MRP_DEVICE_H_

#include <stdint.h>

struct DevChunk {
    const uint32_t *allele_number; /* [n_sites] */
    const uint32_t *allele_offset; /* [n_sites+1] */
    const uint32_t *sub_offset;    /* [n_sites+1] prefix sum of A^2 */
    const uint16_t *sub;           /* substitutionLogProbs */
    const uint16_t *prior;         /* allelePriorLogProbs, indexed by allele_offset */
    const uint8_t *pool;           /* profile bytes */
    const int32_t *same_until;     /* [n_sites] first site after i whose allele count differs from site i's */
};

struct DevCol {
    int64_t cell_off;   /* first cell of the column in the batch cell arrays */
    int64_t mcell_off;  /* first merge cell of the merge column that FOLLOWS this column */
    int64_t slot_off;   /* first allele slot of the column in the plane arrays */
    int64_t read_off;   /* first entry of the column in read_byte_off */
    int32_t n_cells;
    int32_t n_merge;    /* merge cells of the following merge column; 0 for the last column */
    int32_t site_start;
    int32_t n_sites;
    int32_t depth;
    int32_t n_slots;    /* alleles summed over the column's sites */
    int32_t chunk;
    uint32_t flags;     /* MRP_FLAG_* of the owning hmm */
};

/* resident batches: where a column's emission tiles go in the tile array (written on the device, mrp_tiles_kernel) */
struct TileCol {
    int64_t first;           /* index of the column's first tile */
    int32_t uniform_alleles; /* EmitTile.uniform_alleles of the column */
    int32_t pad;
};

/* what the bit-plane kernel needs of a column (read through the scalar cache) */
struct PlaneCol {
    const uint8_t *pool; /* profile pool of the column's chunk */
    int64_t read_off;    /* first entry of the column in read_byte_off */
    int64_t slot_off;    /* first allele slot of the column */
    int32_t depth;
    int32_t n_slots;
    int32_t need_planes; /* the column is handled by the general emission path (mixed allele counts / ancestor model) */
    int32_t pad;
};

/* what the recursion kernels need of a column (read through the scalar cache) */
struct SweepCol {
    int64_t cell_off;
    int64_t mcell_off;
    int32_t n_cells;
    int32_t n_merge;
    int32_t pad[2];
};

/* one unit of work of the emission kernel: up to MRP_EMIT_TILE consecutive cells of one column */
struct EmitTile {
    int64_t cell_off;        /* first cell of the tile in the batch cell arrays */
    int64_t slot_off;        /* first allele slot of the column */
    int32_t n;               /* cells in the tile */
    int32_t col;             /* column index in the batch (general path) */
    int32_t n_sites;
    int32_t uniform_alleles; /* allele count shared by every site of the column, 0 if they differ */
    int32_t depth;
    uint32_t flags;
    int32_t pad[2];
};

struct DevHmm {
    int64_t col0;       /* first column in the batch column arrays */
    int32_t n_cols;
    uint32_t flags;
    int32_t max_merge;  /* largest merge column of this hmm */
    int32_t max_cells;  /* largest column */
    int32_t wide_idx;   /* transitions do not fit 16 bits: kernels read cell_next/cell_prev */
    int32_t pad;
    int64_t n_cells;    /* cells / merge cells of this hmm (contiguous in the batch arrays) */
    int64_t n_merge;
    int64_t cost_bound; /* upper bound of |forward| over the whole hmm (selects the int32 path) */
};

/* sentinel for log(0) in the integer (max-plus) kernels */
#define MRP_NEG_I32 ((int32_t) 0x80000000)

#endif
