/*
 * rphmm_host.h -- internal interface between the C host pipeline (rphmm_host.c) and the HIP side
 * of libmargin_rphmm.so (mrp_api.cpp).  Public declarations live in include/margin_rphmm.h.
 */
#ifndef RPHMM_HOST_H_
#define RPHMM_HOST_H_

#include <stdint.h>

#include "../../include/margin_rphmm.h"

#ifdef __cplusplus
extern "C" {
#endif

/* host-side copies of a chunk's tables (kept by mrp_chunk for the structural code) */
typedef struct mrp_chunk_host {
    int64_t n_sites;
    const uint32_t *allele_number;
    const uint32_t *allele_offset; /* [n_sites+1] */
    const uint32_t *sub_offset;    /* [n_sites+1] */
    const uint16_t *sub;
    const uint16_t *prior;
    const uint8_t *pool;
    int64_t pool_bytes;
} mrp_chunk_host;

void mrp_chunk_host_view(const mrp_chunk *chunk, mrp_chunk_host *out);
mrp_context *mrp_chunk_context(const mrp_chunk *chunk);
/* the concurrent batch (0 .. G - 1) of every chunk of an mrp_phase_reads_many call (rphmm_host.c) */
void mrp_phase_group_assign(int64_t n_chunks, int G, int64_t total_sites, uint8_t *group_of);
int mrp_context_device(const mrp_context *ctx);
/* the context is one of several concurrent batches of its device: no side streams (mrp_internal.h) */
int mrp_context_set_grouped(mrp_context *ctx, int grouped); /* returns the previous setting */
void mrp_context_set_concurrent_batches(mrp_context *ctx, int n); /* how many batches of the call share the device (launch shapes of the chain kernels) */
int mrp_context_calls_sharing_device(const mrp_context *ctx); /* calls a work queue runs on the context's device at a time (1 outside a queue) */
/* device memory of the context's pool: bytes cached for reuse, and what all pools of its device hold together (live + cached) */
void mrp_context_pool_bytes(mrp_context *ctx, int64_t *cached, int64_t *device_held);
void mrp_warn_hw_queues_once(int concurrent_batches); /* one stderr line per process when concurrent batches outnumber the hardware queues */
uint64_t mrp_context_oom_events(mrp_context *ctx);   /* device allocations the driver refused for good on the context's device */
int64_t mrp_context_device_budget(mrp_context *ctx); /* bytes all pools of the context's device may hold together; 0 = unknown */
/* a further context (stream, allocator cache) on the same device, owned by ctx and destroyed with it; i = 0, 1, ... */
mrp_context *mrp_context_sibling(mrp_context *ctx, int i);
int mrp_set_error(int code, const char *fmt, ...);
/* host worker threads for structural code and descriptor building (mrp_set_host_threads, default min(16, cores)) */
int mrp_host_threads(void);
int mrp_context_phase_groups(const mrp_context *ctx);
int mrp_phase_groups_for(const mrp_context *ctx, int64_t n_chunks); /* concurrent batches of a mrp_phase_reads_many call over n_chunks (chunk i: batch i % G) */
int mrp_context_test_hooks(const mrp_context *ctx);
/* urgency of the parallel loops the calling thread posts to the host worker pool from now on (smaller = served first) */
void mrp_pool_set_priority(int p);
void mrp_pool_set_tag(int t);
/* the pool the calling thread posts its loops to (NULL: the process-wide one); threads started on behalf of a caller adopt it */
void *mrp_pool_current(void);
void mrp_pool_adopt(void *pool);
long long mrp_pool_tag_cpu_ns(int tag);
/* fn(i, arg) for every i in [0, n), grain indices at a time, on the caller and the persistent worker pool (mrp_api.cpp) */
void mrp_pool_run(int64_t n, int64_t grain, void (*fn)(int64_t, void *), void *arg);
/* rough cost of one index of the calling thread's next loops, nanoseconds (0: unknown): short loops run on the caller, longer ones wake only
 * as many workers as they can keep busy */
void mrp_pool_set_weight(int ns_per_index);

/* ---- device-resident merge levels (mrp_engine.cpp), driven by the structural code of rphmm_host.c ---- */
typedef struct mrp_engine mrp_engine;

/* One hmm of a tiling path, as a parent of a cross product.  Nothing in it depends on a forward/backward result: the hmm
 * is named by WHERE its pruned form will be (segment = the level that produces it, first column), which is known as soon as
 * that level has been staged -- the level itself may still be running. */
typedef struct mrp_xpar {
    int32_t start, end;   /* site interval [refStart, refStart + refLength) */
    int32_t n_cols;
    int32_t seg;          /* segment of the engine that holds the hmm; -1: a stRPHmm_construct hmm (one column {1, 0}) */
    int64_t col0;         /* its first column in that segment; seg < 0: offset of the read's profile bytes in the chunk's pool */
} mrp_xpar;

/* One cross product hmm of a level: stRPHmm_fuse of two tiling paths, stRPHmm_alignColumns and
 * stRPHmm_createCrossProductOfTwoAlignedHmm (hmm.c:283-750) described by the two paths and the merged column boundaries.
 * Per column the host supplies 8 bytes; the columns' parents, connectors, reads and allele slots are derived on the device
 * (mrp_structure_kernel). */
typedef struct mrp_xhmm {
    const mrp_chunk *chunk;
    uint32_t flags;
    int32_t ref_start, ref_end;
    int32_t n_cols, n_a, n_b;     /* columns; hmms of tiling path A and B (n_b = 0: stRPHmm_fuse of path A alone) */
    const mrp_xpar *par;          /* [n_a + n_b] path A then path B, each in reference order */
    const int32_t *col_start;     /* [n_cols] first site of every column (the last one ends at ref_end) */
    const int32_t *col_read_off;  /* [n_cols + 1] prefix sums of the column depths */
    const int *discarded;         /* optional: non-zero once the hmm's chunk has been given up (a parent was discarded at an earlier level) */
    /* static bounds (mrp_side_bound per side and column): launch classes and range checks; the exact sizes are computed on
     * the device from the parents' counts */
    int64_t bound_cells, bound_merge;
    int64_t depth_sites;          /* sum over the columns of depth x sites */
    int64_t n_col_reads, n_slots; /* col_read_off[n_cols]; allele slots of [ref_start, ref_end) (so that staging a level reads the descriptions only) */
    int32_t bound_max_cells, bound_max_merge;
    /* results, known as soon as the level is staged: where the pruned hmm will be */
    int32_t seg;
    int64_t col0;
    /* mrp_engine_final instead: n_cells[k] = index of the traced-back cell of column k, path_part[k] its
     * partition (host, caller-allocated), and the totals of the final sweep */
    int32_t *n_cells;
    uint64_t *path_part;
    double hmm_forward, hmm_backward;
    /* MRP_ENGINE_ERR_* bits the kernels raised for this hmm; non-zero (structure / merge): its results are not valid and
     * the chunk it belongs to has to be redone on the hashing path */
    int32_t err;
    /* mrp_engine_final_stage, optional (frag_reads != NULL for EVERY hmm of the stage): the genome fragment of the traced-back
     * path on the device -- stGenomeFragment_construct, the refinement rounds and the re-adding of the coverage-filtered reads
     * (genomeFragment.c:40-232, bubbleGraph.c:2761-2779).  In: the chunk's reads, their indices sorted by pool_offset, the
     * filtered-out reads in the order bubbleGraph.c:2772 walks them.  Out (caller-allocated): 20 bytes per site of
     * [ref_start, ref_end) (ancestor, hap1, hap2, reads on either side as bytes; three floats), reads1 / reads2 (capacity
     * 2 n_reads + 2 each) and their lengths; frag_done = 1 when they were filled. */
    const struct mrp_read *frag_reads;
    const int32_t *frag_by_pool, *frag_discarded;
    int32_t frag_n_reads, frag_n_discarded, frag_iterations;
    void *frag_sites;
    int32_t *frag_reads1, *frag_reads2;
    int32_t frag_n1, frag_n2, frag_done;
} mrp_xhmm;

/* static upper bound of the cells one side contributes to a cross product column: a pruned column has at most S cells,
 * and never more than the bipartitions of its reads */
#ifdef __HIPCC__
__host__ __device__
#endif
static inline int64_t mrp_side_bound(int depth, int S) { return depth >= 7 ? S : (((int64_t) 1 << depth) < S ? ((int64_t) 1 << depth) : S); }

typedef struct mrp_engine_stats {
    int64_t levels, hmms, columns, cells, merge_cells;
    double device_ms; /* summed over levels: cross + planes + emission + recursion + prune + compaction */
    double cross_ms, sweep_ms, prune_ms;
    double pack_ms, cross_emit_ms, recursion_ms, prune_kernel_ms, compact_ms; /* the same time by kernel family */
} mrp_engine_stats;

/* MRP_ERR_UNSUPPORTED when the parameters are outside what the resident path handles (log-sum-exp mode,
 * more than MRP_PRUNE_MAX_S partitions per column): the caller then uses the hashing path. */
int mrp_engine_create(mrp_context *ctx, const mrp_params *params, mrp_engine **out);
void mrp_engine_destroy(mrp_engine *e);
int32_t mrp_engine_stride(const mrp_engine *e);
/* device arrays of the hmm that starts at column col0 of segment seg (fixed-stride resident layout, mrp_engine.h) */
int mrp_engine_locate(const mrp_engine *e, int32_t seg, int64_t col0, const uint64_t **part, const uint32_t **np,
                      const int32_t **n_cells, const int32_t **n_merge);
/* A level = cross product -> forward/backward -> prune for n independent hmms, in three steps:
 *   stage   host only (description built and uploaded, column structure derived on the copy stream; may run while the level
 *           before is still on the device); fills seg / col0 of every x[i];
 *   launch  lays the level out on the device and queues its kernels; a large level first waits for the levels before it (their
 *           x[i].err are set then: mrp_engine_levels_ended), a small one does not;
 *   end     waits for the level and sets its x[i].err.
 * x must stay valid until the level has ended. */
int mrp_engine_level_stage(mrp_engine *e, int64_t n, mrp_xhmm *x);
int mrp_engine_level_launch(mrp_engine *e);
int mrp_engine_level_end(mrp_engine *e);
/* Round 5: a launch need not end the level before (small levels are launched without the wait and several are in flight at a time;
 * they end in order, by a later launch or by mrp_engine_level_end, which ends them all).  The number of levels ended so far: the k-th
 * level launched has ended -- its x[i].err are set, x may go -- once this is at least k. */
int64_t mrp_engine_levels_ended(const mrp_engine *e);
/* stage + launch */
int mrp_engine_level_begin(mrp_engine *e, int64_t n, mrp_xhmm *x);
/* stage + launch + end */
int mrp_engine_level(mrp_engine *e, int64_t n, mrp_xhmm *x);
/* the last step of bubbleGraph_phaseBubbleGraph (bubbleGraph.c:2745-2755) for n fused hmms: cross product with
 * nothing (= stRPHmm_fuse with its gap columns), forward/backward with the flags given, stRPHmm_forwardTraceBack */
int mrp_engine_final_stage(mrp_engine *e, int64_t n, mrp_xhmm *x);
int mrp_engine_final(mrp_engine *e, int64_t n, mrp_xhmm *x);
/* device -> host copy of resident arrays (queued), and the wait for all queued copies */
int mrp_engine_fetch(mrp_engine *e, void *dst, const void *src_dev, int64_t bytes);
int mrp_engine_sync(mrp_engine *e);
void mrp_engine_get_stats(const mrp_engine *e, mrp_engine_stats *out);

#define MRP_CONN_NONE 0
#define MRP_CONN_REAL 1
#define MRP_CONN_ZERO 2
#define MRP_CONN_IDENT 3

#ifdef __cplusplus
}
#endif
#endif
