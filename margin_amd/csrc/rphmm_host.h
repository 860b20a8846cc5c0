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
int mrp_context_device(const mrp_context *ctx);
/* a further context (stream, allocator cache) on the same device, owned by ctx and destroyed with it; i = 0, 1, ... */
mrp_context *mrp_context_sibling(mrp_context *ctx, int i);
int mrp_set_error(int code, const char *fmt, ...);
/* host worker threads for structural code and descriptor building (mrp_set_host_threads, default min(16, cores)) */
int mrp_host_threads(void);
int mrp_context_phase_groups(const mrp_context *ctx);
/* urgency of the parallel loops the calling thread posts to the host worker pool from now on (smaller = served first) */
void mrp_pool_set_priority(int p);
void mrp_pool_set_tag(int t);
long long mrp_pool_tag_cpu_ns(int tag);
/* fn(i, arg) for every i in [0, n), grain indices at a time, on the caller and the persistent worker pool (mrp_api.cpp) */
void mrp_pool_run(int64_t n, int64_t grain, void (*fn)(int64_t, void *), void *arg);

/* ---- device-resident merge levels (mrp_engine.cpp), driven by the structural code of rphmm_host.c ---- */
typedef struct mrp_engine mrp_engine;

/* one column of a cross product to build on the device (one step of the aligned piece lists).  Nothing in it depends on a
 * forward/backward result: the number of cells of a parent column is not known when the level is described (the level that
 * prunes the parent may still be running), only WHERE it will be found in HBM. */
typedef struct mrp_xcol {
    const uint64_t *a_part, *b_part; /* device: cells of each side's column; NULL = gap column */
    const uint32_t *a_np, *b_np;     /* device: next | prev << 16 of those cells */
    const int32_t *a_ncells, *b_ncells; /* device: number of cells of each side's column; NULL = 1 (gap column) */
    const int32_t *a_nmerge, *b_nmerge; /* device: merge cells of the parent's merge column after it (MRP_CONN_REAL connectors only) */
    uint8_t d1, d2;                  /* depth per side */
    uint8_t out_a, out_b;            /* connector kinds (MRP_CONN_*) */
    uint8_t out_a_paired, out_b_paired; /* connector mask != 0: its merge cells come in complement pairs */
    uint8_t pad[2];
    uint64_t mask_from, mask_to;     /* masks of the cross product's merge column after this column */
} mrp_xcol;

/* one cross product hmm of a level */
typedef struct mrp_xhmm {
    const mrp_chunk *chunk;
    int32_t n_cols;
    uint32_t flags;
    const mrp_xcol *cols;
    const int32_t *col_ref_start, *col_length, *col_depth;
    const int64_t *col_read_off;  /* [n_cols + 1] */
    const int64_t *read_byte_off;
    /* results, known as soon as the level is staged: where the pruned hmm will be (resident layout, device) */
    uint64_t *d_part;
    uint32_t *d_np;
    int32_t *d_ncells, *d_nmerge; /* [n_cols] cells per column / merge cells of the merge column after it */
    /* mrp_engine_final instead: n_cells[k] = index of the traced-back cell of column k, path_part[k] its
     * partition (host, caller-allocated), and the totals of the final sweep */
    int32_t *n_cells;
    uint64_t *path_part;
    double hmm_forward, hmm_backward;
    /* MRP_ENGINE_ERR_* bits the kernels raised for this hmm; non-zero (structure / merge): its results are not valid and
     * the chunk it belongs to has to be redone on the hashing path */
    int32_t err;
} mrp_xhmm;

typedef struct mrp_engine_stats {
    int64_t levels, hmms, columns, cells, merge_cells;
    double device_ms; /* summed over levels: cross + planes + emission + recursion + prune + compaction */
    double cross_ms, sweep_ms, prune_ms;
} mrp_engine_stats;

/* MRP_ERR_UNSUPPORTED when the parameters are outside what the resident path handles (log-sum-exp mode,
 * more than MRP_PRUNE_MAX_S partitions per column): the caller then uses the hashing path. */
int mrp_engine_create(mrp_context *ctx, const mrp_params *params, mrp_engine **out);
void mrp_engine_destroy(mrp_engine *e);
int32_t mrp_engine_stride(const mrp_engine *e);
/* the column every stRPHmm_construct hmm consists of (hmm.c:97-133): cells {1, 0}, and its cell count (2) */
void mrp_engine_leaf(const mrp_engine *e, const uint64_t **part, const uint32_t **np, const int32_t **n_cells);
/* A level = cross product -> forward/backward -> prune for n independent hmms, in three steps:
 *   stage   host only (description built and uploaded; may run while the level before is still on the device); fills
 *           d_part / d_np / d_ncells / d_nmerge of every x[i];
 *   launch  waits for the level before (its x[i].err are set then), lays the level out on the device and queues its kernels;
 *   end     waits for the level and sets its x[i].err.
 * x must stay valid until the level has ended. */
int mrp_engine_level_stage(mrp_engine *e, int64_t n, mrp_xhmm *x);
int mrp_engine_level_launch(mrp_engine *e);
int mrp_engine_level_end(mrp_engine *e);
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
