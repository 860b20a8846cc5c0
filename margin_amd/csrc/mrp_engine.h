/*
 * mrp_engine.h -- device structures and launch wrappers of the device-resident merge level
 * (cross product -> forward/backward -> prune), SURVEY.md 8(f-1).
 *
 * A pruned hmm lives in HBM with a fixed stride of S cells / S merge cells per column
 * (S = max(minPartitionsInAColumn, maxPartitionsInAColumn) rounded up to 4):
 *     part[k*S + i]  u64   partition of cell i of column k
 *     np[k*S + i]    u32   next | prev << 16 (merge cell indices)
 *     n_cells[k], n_merge[k]
 * (a merge cell's keys are part & mask of any cell that feeds it / is fed by it, so they are not stored)
 * The unpruned cross product of a level lives in that level's batch arrays (mrp_kernels.h).
 */
#ifndef MRP_ENGINE_H_
#define MRP_ENGINE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mrp_kernels.h"

#include "rphmm_host.h" /* MRP_CONN_*: connector kinds between consecutive aligned pieces */

/* CrossCol.flags */
#define MRP_XF_INVERTED 1u      /* includeInvertedPartitions: partition/complement pair order (hmm.c:627-655) */
#define MRP_XF_OUT_A_PAIRED 2u  /* the merge cells of side A's out connector come in complement pairs */
#define MRP_XF_OUT_B_PAIRED 4u
#define MRP_XF_IN_A_PAIRED 8u
#define MRP_XF_IN_B_PAIRED 16u
/* The level's cell arrays hold UNITS: with inverted partitions every cell of a cross product column sits next to its complement
 * (cells 2u, 2u + 1), with the same emission cost, f and b, and the twin's merge cells are the twins of its merge cells -- so the
 * recursion over one member of each pair IS the recursion (max over the cells that feed a merge cell = max over the units that feed
 * its unit).  cross product + emission writes cost and transitions once per unit (transition = merge UNIT index), the recursion
 * kernel runs unchanged on arrays of half the size, the prune reads f and b per unit.  A column / merge column of one
 * self-complementary cell is one unit.  Set for the fused merge levels when the prune runs on pairs (PruneParams.pairs == 2). */
#define MRP_XF_UNITS 32u

/* one column of stRPHmm_createCrossProductOfTwoAlignedHmm (hmm.c:534-750) */
struct CrossCol {
    const uint64_t *a_part; /* cells of side A's column; NULL = gap column (one cell, partition 0, depth 0) */
    const uint64_t *b_part;
    const uint32_t *a_np;
    const uint32_t *b_np;
    int64_t x_cell_off;     /* first cell of the column in the level's batch arrays */
    uint16_t C1, C2;        /* cells per side */
    uint16_t Ma, Mb;        /* merge cells per side of the connector that leaves the column */
    uint16_t Pa, Pb;        /* merge cells per side of the connector that enters the column */
    uint8_t d1, d2;         /* depth per side */
    uint8_t out_a, out_b, in_a, in_b; /* MRP_CONN_* */
    uint8_t flags;
    uint8_t pad;
};

/* ---- layout of a level on the device ------------------------------------------------------------------------
 * The host describes a level by what does not depend on any forward/backward result: the column structure of the cross
 * products (sites, reads, connector kinds, masks) and WHERE the parents' per-column cell / merge cell counts will be
 * found in HBM once the level that produces them has run.  The sizes of the cross product columns (C1 x C2 cells,
 * Ma x Mb merge cells), every offset derived from them and all kernel descriptors are computed here, on the device;
 * the host reads back four totals per level (what its allocator needs). */
struct PlanCol {
    const uint64_t *a_part, *b_part;    /* parents' cells (NULL = gap column: one cell, partition 0) */
    const uint32_t *a_np, *b_np;
    const int32_t *a_ncells, *b_ncells; /* device: cells of the parent column (NULL = 1) */
    const int32_t *a_nmerge, *b_nmerge; /* device: merge cells of the parent's merge column after it (REAL connectors) */
    int64_t read_off;                   /* first entry of the column in the batch's read_byte_off */
    int64_t slot_off;                   /* first allele slot of the column in the batch */
    int32_t site_start, n_sites, depth, n_slots;
    int32_t chunk;                      /* index into the batch's DevChunk table */
    int32_t uniform_alleles;            /* allele count shared by the column's sites, 0 if they differ */
    uint8_t d1, d2, out_a, out_b;       /* depth per side, connector kinds out of the column (MRP_CONN_*) */
    uint8_t out_a_paired, out_b_paired, need_planes, last; /* last: last column of its hmm */
    uint32_t pad;
};
struct PlanHmm {
    int64_t col0;        /* first column of the hmm in the level */
    int32_t n_cols;
    uint32_t flags;      /* MRP_FLAG_* of the sweep */
    int64_t cost_bound;
};
struct LayoutTot {       /* per hmm, after the counting pass */
    int64_t cells, merge;    /* entries of the level's cell / merge cell arrays (units when MRP_XF_UNITS) */
    int32_t tiles_fast, tiles_gen, max_cells, max_merge;
    int64_t acells, amerge;  /* cells / merge cells of the cross product itself (SURVEY.md 8d counts these) */
};
struct LayoutBase {      /* per hmm, after the scan over the hmms */
    int64_t cell0, mcell0, tile_fast0, tile_gen0;
};
struct LayoutOut {       /* everything the layout kernels write */
    uint16_t *dims;      /* [n_cols][4] C1, C2, Ma, Mb (scratch between the passes) */
    LayoutTot *tot;      /* [n_hmms] */
    LayoutBase *base;    /* [n_hmms] */
    int64_t *tile_sums;  /* [ceil(n_hmms / 256)][6] scratch of the scan over the hmms */
    int64_t *totals;     /* [6] array entries for cells (padded to a multiple of 4 per hmm) and merge cells, fast tiles, general tiles; [4], [5]:
                          * cells and merge cells of the cross products themselves (equal to [0], [1] up to the padding unless MRP_XF_UNITS) */
    DevHmm *hmms;
    DevCol *cols;
    SweepCol *scols;
    PlaneCol *pcols;
    TileCol *tilecols;
    CrossCol *ccols;
};
/* ---- column structure of a level, on the device -----------------------------------------------------------------
 * The columns of a cross product (stRPHmm_fuse + stRPHmm_alignColumns + stRPHmm_createCrossProductOfTwoAlignedHmm,
 * hmm.c:283-750) depend on read intervals only.  The host decides WHICH hmms are merged (tiling paths, overlap
 * components: a few hundred intervals per chunk) and merges their column boundaries (4 bytes per column); everything
 * else a column needs -- which parent column each side is cut from, connector kinds, the column's reads and where their
 * profile bytes start, allele slots, the PlanCol the layout kernels read -- is derived here, one thread per column,
 * from the parents' own column tables, which stay in HBM from the level that built them (ResCol). */
struct ResCol {          /* one column of an hmm that lives in a segment (persistent: its children look it up) */
    int64_t rbo_off;     /* first entry of the column in the segment's read_byte_off array */
    int32_t start;       /* first site */
    uint8_t depth;
    uint8_t cont;        /* some read of the column goes on into the next column (maskFrom != 0, mergeColumn.c) */
    uint16_t pad;
};
struct SegDev {          /* the arrays of one level's segment */
    const uint64_t *part;
    const uint32_t *np;
    const int32_t *n_cells, *n_merge;
    const ResCol *cols;
    const int64_t *rbo;
};
struct XDesc {           /* one cross product hmm of the level being described */
    int64_t col0;        /* first column in the level */
    int64_t read0;       /* first entry in the level's read_byte_off */
    int64_t slot0;       /* first allele slot */
    int64_t par0;        /* first mrp_xpar of tiling path A; path B follows */
    int32_t ref_start, ref_end;
    int32_t n_cols, n_a, n_b;
    int32_t chunk;       /* index into the level's DevChunk table */
    uint32_t flags;      /* MRP_FLAG_* of the sweep */
    int32_t prune_pos;   /* position of the hmm in the level's PruneHmm array (-> col_hmm) */
};
struct StructureIn {
    const XDesc *xd; int64_t n_hmms, n_cols;
    const mrp_xpar *par;
    const int32_t *col_start;  /* [n_cols] */
    const int32_t *col_roff;   /* [n_cols] offset of the column's reads within its hmm */
    const SegDev *segs;
    const DevChunk *chunks;
    const uint64_t *leaf_part; const uint32_t *leaf_np; const int32_t *leaf_count; /* the column of a stRPHmm_construct hmm */
    int32_t stride;            /* cells per column of the resident layout */
    int32_t fused;             /* cross product and emission in one pass: no column needs bit planes */
    /* outputs */
    PlanCol *plan; ResCol *cols; int64_t *rbo; int32_t *col_hmm;
    int32_t *err, *err_hmm;    /* MRP_ENGINE_ERR_RANGE: a column that no parent column covers consistently */
};
hipError_t mrp_launch_structure(const StructureIn &in, hipStream_t stream);

/* plan_dev / hmms_dev: PlanCol [n_cols], PlanHmm [n_hmms]; chunks_dev: the batch's DevChunk table; S: cells a pruned column
 * can have at most (parents' counts are clamped to it, so that a discarded parent cannot blow the level up) */
hipError_t mrp_launch_layout(const PlanCol *plan_dev, const PlanHmm *hmms_dev, int64_t n_hmms, int64_t n_cols, const DevChunk *chunks_dev,
                             int32_t S, uint32_t xflags, LayoutOut out, hipStream_t stream); /* xflags: MRP_XF_INVERTED, MRP_XF_UNITS */

/* one cross product hmm of a level, as the prune kernels see it */
struct PruneHmm {
    int64_t col0;           /* first column in the level's batch column arrays / in the scratch lists */
    int32_t n_cols;
    int32_t hmm_index;      /* index into MrpBatchDev.hmm_fb */
    uint64_t *out_part;     /* pruned hmm, column k at + k * S; final level: partition of the traced-back cell per column */
    uint32_t *out_np;
    int32_t *out_n_cells;   /* [n_cols]; final level: index of the traced-back cell per column */
    int32_t *out_n_merge;   /* [n_cols] (last entry 0) */
};

struct PruneParams {
    int32_t S;              /* stride of the pruned layout and capacity of the kept lists */
    int32_t min_p, max_p;   /* min/maxPartitionsInAColumn */
    int32_t n_bins;         /* posterior keys: bin = min(total - f - b, n_bins - 1); exp(-(n_bins - 1)) == 0 */
    int32_t thr_bin;        /* bins <= thr_bin have posterior >= minPosteriorProbabilityForPartition */
    int32_t max_cells, max_merge; /* largest column / merge column of the level (LDS sizing) */
    int32_t pad;            /* fault injection of the tests (mrp_context_set_test_hooks bit 0) */
    int32_t pairs;          /* includeInvertedPartitions with even min_p / max_p: the prune chain runs on complement pairs; 2: and the
                             * level's f / b / merge arrays hold units (MRP_XF_UNITS) */
    int32_t pad2;
};

struct PruneScratch {
    uint16_t *kept;         /* [n_cols * S] kept cells of each column (index within the column), in kept order */
    uint32_t *kept_np;      /* [n_cols * S] their next | prev << 16 */
    uint16_t *keptm;        /* [n_cols * S] kept merge cells of the merge column after each column */
    int32_t *n_kept;        /* [n_cols] */
    int32_t *n_keptm;       /* [n_cols] */
    int32_t *err;           /* [4] bit flags: MRP_ENGINE_ERR_* (all hmms of the level) */
    int32_t *err_hmm;       /* [n_hmms] the same per hmm, indexed like the PruneHmm array */
};

/* ---- genome fragments of the final hmms on the device (emissions.c:246-343, genomeFragment.c:40-232, bubbleGraph.c:2761-2779) ----
 * One record per final hmm (= per chunk); the arrays it points into are the level's own. */
struct FragRead { int32_t ref_start, length; int64_t pool_offset; };
struct FragSite { uint8_t ancestor, hap1, hap2, support1, support2, pad[3]; float genotype_prob, hap_prob1, hap_prob2; }; /* 20 B */
struct FragHmm {
    int64_t reads0;        /* first FragRead of the chunk's reads in the level's table; the same offset into by_pool */
    int64_t disc0;         /* first entry of its discarded reads (coverage filter, coordination.c:443-488) */
    int64_t site0;         /* first FragSite of its result */
    int64_t list0;         /* its reads1 / reads2 results start at 2 x list0 in lists (capacity 2 n_reads + 2 each), its work lists in work */
    int64_t slot0;         /* first entry of its per-(column, read) scratch (= its first entry in read_byte_off) */
    int32_t n_reads, n_discarded;
    int32_t ref_start, length;  /* sites of the fragment */
    int32_t max_iterations;     /* roundsOfIterativeRefinement */
    int32_t pad;
};
struct FragArrays {
    const FragHmm *hmms;
    const FragRead *reads;      /* per chunk, in the caller's read order */
    const int32_t *by_pool;     /* per chunk: its read indices sorted by pool_offset (a column names a read by its profile bytes) */
    const int32_t *discarded;
    FragSite *sites;            /* out */
    int32_t *lists;             /* out: per chunk 2 x (2 n_reads + 2): reads1, reads2 */
    int32_t *work;              /* scratch: the same size, the lists of the round in progress */
    int32_t *counts;            /* out: [n_hmms][2] entries of reads1 / reads2 */
    int32_t *col_read;          /* scratch: read index per (column, slot), indexed like read_byte_off */
    uint64_t *col_part;         /* scratch: current partition per column (the trace back's, then refined) */
    uint32_t *read_key;         /* scratch: per read 2 x first sighting key, then the move flags */
    int32_t *col_cnt;           /* scratch: per column 2 x first-sighting counts */
};
hipError_t mrp_launch_fragments(const MrpBatchDev &d, const PruneHmm *hmms_dev, int64_t n_hmms, FragArrays fa, int32_t *err, int32_t *err_hmm,
                                hipStream_t stream);

#define MRP_ENGINE_ERR_STRUCTURE 1 /* a parent is not in complement-pair order: closed-form cross product not valid */
#define MRP_ENGINE_ERR_POSTERIOR 2 /* f + b > total (column.c:183 "invalid prob") */
#define MRP_ENGINE_ERR_RANGE 4     /* index out of range */
#define MRP_ENGINE_ERR_MERGE 8     /* a next merge cell of the kept cells would itself be pruned (hmm.c:1090-1100): not handled on the device */

/* largest column the prune kernel handles (cell and merge cell indices travel in 14 bits) */
#define MRP_PRUNE_MAX_CELLS 13824 /* 6 waves x 64 lanes x 36 cells: what one bin-streaming group of the prune kernel holds in registers */
#define MRP_PRUNE_MAX_S 116 /* 116 * 116 cells per cross product column <= MRP_PRUNE_MAX_CELLS */

hipError_t mrp_launch_cross(const CrossCol *cols_dev, int64_t n_cols, uint64_t *partition, uint32_t *cell_np, int32_t *err,
                            const int32_t *col_hmm_dev, int32_t *err_hmm, hipStream_t stream);
/* Cross product and emission in one pass (merge levels without the ancestor substitution model): writes d.cell_np and
 * d.cell_cost from the parents' cells and the packed profile bytes (d.slot_bytes, d.slot_total); no partition array. */
hipError_t mrp_launch_cross_emit(const CrossCol *cols_dev, const MrpBatchDev &d, int32_t *err, const int32_t *col_hmm_dev, int32_t *err_hmm,
                                 int32_t level_max_cells /* largest column of the level by the static bounds: sizes the kernel's tables */,
                                 bool mostly_narrow /* most hmms of the level hold at most 64 array entries per column */, hipStream_t stream);
/* ccols_dev: the level's cross product descriptors, indexed like the batch's columns (the prune kernel enumerates the
 * cells linked to a kept merge cell from the parents' transition arrays) */
hipError_t mrp_launch_prune(const MrpBatchDev &d, const CrossCol *ccols_dev, const PruneHmm *hmms_dev, int64_t n_hmms, PruneParams p,
                            PruneScratch s, hipStream_t stream);
/* stRPHmm_forwardTraceBack (hmm.c:165-219) for every hmm of the level: out_n_cells[k] = cell index,
 * out_part[k] = its partition */
hipError_t mrp_launch_traceback(const MrpBatchDev &d, const PruneHmm *hmms_dev, int64_t n_hmms, int32_t *err, int32_t *err_hmm,
                                hipStream_t stream);
/* d.partition == NULL: the level was produced by mrp_launch_cross_emit, partitions of the kept cells come from the parents */
hipError_t mrp_launch_compact(const MrpBatchDev &d, const CrossCol *ccols_dev, const PruneHmm *hmms_dev, const int32_t *col_hmm_dev, int64_t n_cols,
                              int64_t n_hmms_here, PruneParams p, PruneScratch s, hipStream_t stream); /* columns of hmms at positions >= n_hmms_here are skipped */
/* Recursion, prune and compaction of hmms whose columns hold at most 64 units and 64 merge units (MRP_XF_UNITS levels) by one wave
 * each: hmms_dev[0 .. n_hmms) are the level's PruneHmm records from position hmm0 on (the error flags are indexed by position). */
#define MRP_MINI_MAX_UNITS 64
hipError_t mrp_launch_mini(const MrpBatchDev &d, const CrossCol *ccols_dev, const PruneHmm *hmms_dev, int64_t n_hmms, int64_t hmm0, PruneParams p,
                           PruneScratch s, hipStream_t stream);

#endif
