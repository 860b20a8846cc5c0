/*
 * margin_rphmm.h -- C-ABI of libmargin_rphmm.so: the MI355X (gfx950) engine for margin's
 * read-partitioning HMM (stRPHmm) forward/backward hot path.
 *
 * This is the drop-in boundary.  Every entry point is extern "C", takes
 * plain pointers and sizes, returns an int status (MRP_OK == 0) and never aborts the process.
 * The reference interface each entry point stands in for is cited as <file>:<line> relative to
 * the upstream tree (UCSC-nanopore-cgl/margin @ v1).  INTEGRATION.md shows the adaptor a margin
 * maintainer adds to impl/hmm.c to route stRPHmm_forwardBackward through mrp_fb_run().
 *
 * Vocabulary follows the reference: a *chunk* owns a reference (sites x alleles) and the uint8
 * profile bytes of its reads; an *hmm job* is one stRPHmm flattened to arrays: columns, cells
 * (64-bit read bipartitions), merge columns (maskFrom/maskTo) and merge cells.
 *
 * Threading: a context is bound to one device and one HIP stream and is NOT shared between
 * threads; create one context per host thread (reference: phase.c:276 OpenMP chunk loop).
 */
#ifndef MARGIN_RPHMM_H_
#define MARGIN_RPHMM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes -------------------------------------------------------------------------- */
#define MRP_OK 0
#define MRP_ERR_ARG 1        /* malformed job (reference: st_errAbort paths, hmm.c:293-319,542-555) */
#define MRP_ERR_NO_DEVICE 2  /* no usable gfx950 device: the product path fails loudly, no CPU fallback */
#define MRP_ERR_HIP 3        /* a HIP runtime call failed; see mrp_last_error() */
#define MRP_ERR_NOMEM 4
#define MRP_ERR_UNSUPPORTED 5 /* e.g. more than MRP_MAX_ALLELES alleles at a site in ancestor mode */
#define MRP_ERR_LOOKUP 6     /* a cell's masked partition has no merge cell (mergeColumn.c:63-79 returned NULL) */

/* ---- limits (inc/margin.h:135,142) --------------------------------------------------------- */
#define MRP_ALLELE_LOG_PROB_BITS 8
#define MRP_MAX_READ_PARTITIONING_DEPTH 64
#define MRP_MAX_ALLELES 16

/* ---- flags: the two stRPHmmParameters fields read by the sweep (hmm.c:819, emissions.c:236) - */
#define MRP_FLAG_MAX_NOT_SUM 1u             /* maxNotSumTransitions: logAddP = max (hmm.c:15-20) */
#define MRP_FLAG_INCLUDE_ANCESTOR_SUB_PROB 2u /* includeAncestorSubProb (emissions.c:205-218) */

typedef struct mrp_context mrp_context;
typedef struct mrp_chunk mrp_chunk;
typedef struct mrp_batch mrp_batch;

/* Human-readable text for the last failure on the calling thread. */
const char *mrp_last_error(void);
/* "margin_rphmm <version> gfx950" */
const char *mrp_version(void);
/* The binary interface of this header: bumped whenever a function's parameters or a structure's layout change (round 4 changed
 * mrp_queue_dry_run and mrp_phase_many_stats without one).  A caller built against another header finds out by comparing
 * mrp_abi_version() with the MRP_ABI_VERSION it was compiled with, before it passes a structure in
 * (integration/stRPHmm_forwardBackward_adaptor.c does so when it makes its first context). */
#define MRP_ABI_VERSION 5
int mrp_abi_version(void);
/* Number of visible HIP devices (0 if none / runtime unavailable). */
int mrp_device_count(void);

/* Context: device + stream + reusable device workspace.  (No reference counterpart: the CPU
 * path allocates scratch per column, hmm.c:836,872.)  ONE HOST THREAD PER CONTEXT at a time: every entry point that takes a
 * context (mrp_chunk_create, mrp_batch_*, mrp_fb_run, mrp_phase_reads*, mrp_get_rp_hmms*, ...) uses its streams, its event and
 * its allocator cache without locking -- concurrent callers (phase.c:276 runs its chunk loop under OpenMP) take a context
 * each, as the adaptor in integration/ does per thread; contexts are cheap and share the device's memory budget. */
int mrp_context_create(int device, mrp_context **out);
void mrp_context_destroy(mrp_context *ctx);
int mrp_context_synchronize(mrp_context *ctx);
/* A context keeps the device memory of its last calls cached (allocating and freeing gigabytes per call costs more than the
 * call); this gives the cache back to the driver, e.g. before another context of the same device starts a large job. */
int mrp_context_trim(mrp_context *ctx);
/* Optional, once, BEFORE the process's first HIP call (its own or this library's): asks the ROCm runtime for 16 hardware
 * queues (GPU_MAX_HW_QUEUES, unless the environment already sets it).  The runtime multiplexes HIP streams onto 4 hardware
 * queues by default and kernels of streams that share a queue serialize; the eight concurrent batches of a
 * mrp_phase_reads_many call launch on two streams each, 16 in all (576 chunks: 189 ms with 16 queues, 225 with 8).  Without the
 * call everything works, slower. */
int mrp_runtime_init(void);
/* mrp_phase_reads_many splits its chunks into this many interleaved batches that run concurrently on the context and its
 * sibling contexts (one batch's host work beside the others' kernels); 1..16, or 0 (default): one batch per 12 chunks up to
 * 4, eight from 192 chunks on.  (A call whose units exceed what the device's memory budget holds runs as consecutive slices.) */
int mrp_context_set_phase_groups(mrp_context *ctx, int groups);
/* Test suite only (a private switch, not a parameter: a caller's uninitialised struct field cannot turn it on).  Bit 0, fault
 * injection: the resident path reports MRP_ENGINE_ERR_MERGE for one hmm of its second level, which must send exactly that
 * chunk to the hashing path.  Bit 1: the resident merge levels run the separate cross product and emission kernels -- the
 * path the final level and the ancestor model take -- instead of the one-pass kernel, for A/B parity and timing.  Bit 2 (one shot:
 * cleared when it fires): the next device allocation of at least 1 MB on the context's device is refused as out of memory, which
 * a resident call must survive by redoing itself in two halves.  Bit 3: the prune kernel's general chain (one entry per cell) instead
 * of the chain on complement pairs that includeInvertedPartitions with even column limits selects, for A/B parity. */
int mrp_context_set_test_hooks(mrp_context *ctx, int hooks);
/* size of the host worker pool (structure of the merge levels, descriptors, classification of alignment pairs): the
 * process-wide pool of contexts used directly, and EACH worker's own pool of a work queue (mrp_queue_*: one pool per device).
 * Default: min(16, hardware threads) for the process-wide pool, hardware threads / devices (2..16) per queue worker.
 * Takes effect for pools not yet started. */
int mrp_set_host_threads(int n);

/*
 * Chunk = stReference + all stProfileSeq bytes of one genome chunk, uploaded once.
 *   n_sites, allele_number[n_sites]                 stSite.alleleNumber   (inc/margin.h:164-171)
 *   substitution_log_probs: concatenated A_s*A_s uint16 tables, site-major ([from*A+to],
 *       emissions.c:13-19); NULL = all zero
 *   allele_prior_log_probs: concatenated A_s uint16; NULL = all zero
 *   profile_pool[pool_bytes]: the profileProbs arrays of every read, back to back
 *       (profileSeq.c:13-29; bubbleGraph.c:2423-2435 for the byte encoding)
 * stSite.alleleOffset is the prefix sum of allele_number and is derived internally.
 */
int mrp_chunk_create(mrp_context *ctx, int64_t n_sites, const uint32_t *allele_number,
                     const uint16_t *substitution_log_probs, const uint16_t *allele_prior_log_probs,
                     const uint8_t *profile_pool, int64_t pool_bytes, mrp_chunk **out);
void mrp_chunk_destroy(mrp_chunk *chunk);

/*
 * One flattened stRPHmm (inc/margin.h:340-353,393-402,421-425,439-445,463-467).
 * All arrays are host memory owned by the caller; cells are in LIST ORDER (column->head ...
 * ->nCell) and the outputs are written back in the same order.  A job whose seven output
 * pointers are all NULL is device-only: it is swept, its results stay in HBM.
 */
typedef struct mrp_hmm_job {
    const mrp_chunk *chunk;
    int32_t n_columns;            /* stRPHmm.columnNumber (>= 1) */
    uint32_t flags;               /* MRP_FLAG_* */
    const int32_t *col_ref_start; /* [K] stRPColumn.refStart (site index) */
    const int32_t *col_length;    /* [K] stRPColumn.length   (sites, > 0) */
    const int32_t *col_depth;     /* [K] stRPColumn.depth    (0..64; 0 = gap column, hmm.c:337-345) */
    const int64_t *col_cell_off;  /* [K+1] prefix sum of cells per column */
    const int64_t *col_read_off;  /* [K+1] prefix sum of depth */
    const int64_t *read_byte_off; /* [sum depth] offset into profile_pool of column->seqs[i] (hmm.c:121-122) */
    const uint64_t *partition;    /* [sum C] stRPCell.partition */
    const uint64_t *mask_from;    /* [K-1] stRPMergeColumn.maskFrom */
    const uint64_t *mask_to;      /* [K-1] stRPMergeColumn.maskTo */
    const int64_t *mcol_cell_off; /* [K]   prefix sum of merge cells per merge column (K-1 entries + 1) */
    const uint64_t *merge_from;   /* [sum M] stRPMergeCell.fromPartition */
    const uint64_t *merge_to;     /* [sum M] stRPMergeCell.toPartition */
    /* Optional pre-resolved transitions: index (within the adjacent merge column) of the merge
     * cell each cell feeds (mergeColumn.c:63) / is fed by (mergeColumn.c:72).  NULL = resolved by
     * the library from partition & mask against merge_from / merge_to.  First column's cell_prev
     * and last column's cell_next entries are ignored. */
    const uint32_t *cell_next;    /* [sum C] or NULL */
    const uint32_t *cell_prev;    /* [sum C] or NULL */
    /* Outputs = post-conditions of stRPHmm_forwardBackward (hmm.c:931-942) */
    double *cell_forward;         /* [sum C] stRPCell.forwardLogProb */
    double *cell_backward;        /* [sum C] stRPCell.backwardLogProb (excludes own emission, hmm.c:881-892) */
    double *merge_forward;        /* [sum M] stRPMergeCell.forwardLogProb */
    double *merge_backward;       /* [sum M] stRPMergeCell.backwardLogProb */
    double *col_total;            /* [K] stRPColumn.totalLogProb */
    double *hmm_forward;          /* [1] stRPHmm.forwardLogProb */
    double *hmm_backward;         /* [1] stRPHmm.backwardLogProb */
} mrp_hmm_job;

/*
 * stRPHmm_forwardBackward for n_jobs independent HMMs in one device batch ("run_many").
 * Replaces: impl/hmm.c:931 (callers coordination.c:312, bubbleGraph.c:2749, hmm.c:1332).
 * Upload -> bit-plane kernel -> emission kernel -> recursion kernel -> download, synchronous on return.
 */
int mrp_fb_run(mrp_context *ctx, int64_t n_jobs, const mrp_hmm_job *jobs);

/*
 * Device-resident batches: the same sweep split into its stages so that many HMMs (all
 * components of a merge level, both strands, many chunks) stay in HBM and launches can be
 * timed on their own.  mrp_fb_run == create + add* + upload + launch + download + destroy.
 */
int mrp_batch_create(mrp_context *ctx, mrp_batch **out);
int mrp_batch_add(mrp_batch *batch, const mrp_hmm_job *job); /* copies the job's inputs */
int mrp_batch_upload(mrp_batch *batch);                      /* H2D + transition resolve */
int mrp_batch_launch(mrp_batch *batch);                      /* async on the context stream */
int mrp_batch_download(mrp_batch *batch);                    /* D2H + scatter into job outputs; syncs */
void mrp_batch_destroy(mrp_batch *batch);

/* Launch statistics of the most recent mrp_batch_launch on this batch. */
typedef struct mrp_launch_stats {
    double planes_ms;   /* bit-plane kernel, HIP-event time on the context stream */
    double emission_ms; /* emission kernel (all cells of the batch), HIP-event time */
    double sweep_ms;    /* forward/backward recursion kernel(s), HIP-event time */
    int64_t n_hmms, n_columns, n_cells, n_merge_cells;
    int64_t profile_bytes;      /* sum_k depth_k * alleles_k */
    int64_t algorithmic_bytes;  /* sum_k 24*C_k + 32*M_k + depth_k*alleles_k + 8 */
    int64_t popcount_ops;       /* sum_k C_k * L_k * 2 * A * 8 popcount64 of the CPU formulation */
    int64_t units;              /* not known to the library; 0 */
    /* the same three durations averaged over the batch's launches since the previous mrp_batch_stats call (at most the 32
     * most recent): launches may be queued back to back without waiting for each other, every one keeps its own events */
    double avg_planes_ms, avg_emission_ms, avg_sweep_ms;
    int64_t launches_averaged;
    /* which recursion kernel the hmms of the batch took: max-plus int32 (merge column in LDS), log-sum-exp with the merge
     * column in LDS (sum mode, reproducible), generic fp64 (oversize merge columns / 32-bit transitions) */
    int64_t n_hmms_int32, n_hmms_lse, n_hmms_generic;
} mrp_launch_stats;
/* Waits for the launch to finish, then fills stats. */
int mrp_batch_stats(mrp_batch *batch, mrp_launch_stats *out);

/*
 * Emission-only entry points, the unit-tested secondary seam (inc/margin.h:217-233).
 * mrp_count_bit_vectors replaces calculateCountBitVectors (emissions.c:91-123): for one column
 * (depth reads, n_alleles consecutive allele slots, seqs given as offsets into the chunk's
 * profile pool) returns planes[n_alleles*8] computed on the device.
 * mrp_emissions replaces emissionLogProbability (emissions.c:221-240) for n_cells partitions of
 * one column; out[i] = -(double)cost, exactly as the reference returns it.
 */
int mrp_count_bit_vectors(mrp_context *ctx, const mrp_chunk *chunk, int32_t first_site,
                          int32_t n_sites, int32_t depth, const int64_t *read_byte_off,
                          uint64_t *planes_out);
int mrp_emissions(mrp_context *ctx, const mrp_chunk *chunk, int32_t first_site, int32_t n_sites,
                  int32_t depth, const int64_t *read_byte_off, uint32_t flags, int64_t n_cells,
                  const uint64_t *partitions, double *out);

/* ============================================================================================
 * Host pipeline around the sweep (SURVEY.md section 8a last row / 8f): the structural stRPHmm
 * operations that DEFINE the inputs of every sweep, written against flat arrays instead of
 * linked lists + hash tables so that an hmm is, byte for byte, a device-ready mrp_hmm_job.
 * Same names, argument meaning and error behaviour as the reference functions they mirror;
 * every stRPHmm_forwardBackward inside them runs on the device through mrp_fb_run.
 * ============================================================================================ */

/* The stRPHmmParameters fields read by impl/hmm.c, coordination.c, genomeFragment.c
 * (inc/margin.h:239-322; shipped values: params/base_params.json "phase"). */
typedef struct mrp_params {
    int32_t max_not_sum_transitions;
    int32_t include_inverted_partitions;
    int32_t include_ancestor_sub_prob;
    int32_t reserved; /* must be 0 (the resident path rejects anything else: MRP_ERR_ARG) */
    int64_t min_partitions_in_a_column;
    int64_t max_partitions_in_a_column;
    double min_posterior_probability_for_partition;
    int64_t max_coverage_depth;
    int64_t min_read_coverage_to_support_phasing_between_heterozygous_sites;
    int64_t rounds_of_iterative_refinement;
} mrp_params;

/* One stProfileSeq (inc/margin.h:191-203): its bytes live in the chunk's profile pool. */
typedef struct mrp_read {
    const char *name;      /* readId; orders hmms that share start and length (hmm.c:82-87) */
    int32_t ref_start;     /* first site */
    int32_t length;        /* number of sites (> 0) */
    int32_t forward_strand; /* BamChunkRead.forwardStrand (bubbleGraph.c:2709) */
    int32_t reserved;
    int64_t pool_offset;   /* offset of profileProbs in the chunk's pool */
} mrp_read;

typedef struct mrp_hmm mrp_hmm; /* a flat stRPHmm */

/* getRPHmms (coordination.c:490-516) over reads[read_index[0..n)): tiling paths, hierarchical
 * merge (fuse -> align -> cross product -> forward/backward on the device -> prune).  Returns a
 * malloc'd array of hmms ordered and non-overlapping in reference coordinates.  If record is not
 * NULL every sweep issued is also appended to that batch (mrp_batch_add) for later replay. */
int mrp_get_rp_hmms(mrp_context *ctx, const mrp_chunk *chunk, const mrp_read *reads, const int32_t *read_index,
                    int64_t n, const mrp_params *params, mrp_batch *record, mrp_hmm ***hmms_out, int64_t *n_out);
void mrp_hmm_destroy(mrp_hmm *hmm);
void mrp_free(void *p);
/* Expose an hmm's arrays as an mrp_hmm_job (inputs and the outputs of its last sweep; pointers
 * stay owned by the hmm).  col_reads_out (optional) receives the per-column read ids. */
int mrp_hmm_view(const mrp_hmm *hmm, mrp_hmm_job *view, const int32_t **col_reads_out, int32_t *ref_start,
                 int32_t *ref_length);
/* stRPHmm_forwardBackward (hmm.c:931) on a flat hmm; flags from params. */
int mrp_hmm_forward_backward(mrp_context *ctx, const mrp_chunk *chunk, mrp_hmm *hmm, const mrp_params *params,
                             mrp_batch *record);
/* stRPHmm_prune (hmm.c:1160) */
int mrp_hmm_prune(mrp_hmm *hmm, const mrp_params *params);
/* stRPHmm_forwardTraceBack (hmm.c:165-219): cell index (within its column) per column. */
int mrp_hmm_forward_trace_back(const mrp_hmm *hmm, int32_t *cell_index_per_column);
/* stRPHmm_split (hmm.c:1231-1300): hmm keeps [refStart, split_point), *suffix_out receives the rest (same reads array as
 * the one the hmm was built from).  The column holding the split point is cut in two; results of a sweep are dropped.
 * MRP_ERR_ARG where the reference aborts (split point outside (refStart, refStart + refLength)). */
int mrp_hmm_split(const mrp_chunk *chunk, const mrp_read *reads, int64_t n_reads, mrp_hmm *hmm, int32_t split_point,
                  mrp_hmm **suffix_out);
/* stRPHMM_splitWherePhasingIsUncertain (hmm.c:1322-1383): forward/backward on the device, trace back, predicted
 * haplotypes; between consecutive heterozygous sites spanned by fewer than
 * min_read_coverage_to_support_phasing_between_heterozygous_sites reads the hmm is cut half way.  Returns a malloc'd array
 * whose first entry is the (shortened) input hmm; the caller owns all of them. */
int mrp_hmm_split_where_phasing_is_uncertain(mrp_context *ctx, const mrp_chunk *chunk, const mrp_read *reads, int64_t n_reads,
                                             mrp_hmm *hmm, const mrp_params *params, mrp_hmm ***hmms_out, int64_t *n_out);

/* stGenomeFragment (inc/margin.h:482-520) + read partition, as plain arrays. */
typedef struct mrp_phase_result {
    int32_t ref_start, length;
    uint64_t *genotype_string, *haplotype_string1, *haplotype_string2, *ancestor_string;
    uint64_t *reads_supporting_haplotype1, *reads_supporting_haplotype2;
    float *genotype_probs, *haplotype_probs1, *haplotype_probs2;
    int32_t *reads1, *reads2; /* indices into the reads array */
    int64_t n_reads1, n_reads2;
    double hmm_forward, hmm_backward; /* of the final sweep */
    int64_t n_sweeps;                 /* forward/backward sweeps issued */
} mrp_phase_result;

/* bubbleGraph_phaseBubbleGraph (bubbleGraph.c:2673-2801) from profile sequences: coverage
 * filter, strand split, getRPHmms per strand (ancestor model off), join, final sweep (ancestor
 * model on), trace back, genome fragment, iterative refinement, re-adding filtered reads. */
int mrp_phase_reads(mrp_context *ctx, const mrp_chunk *chunk, const mrp_read *reads, int64_t n_reads,
                    const mrp_params *params, mrp_batch *record, mrp_phase_result **out);
void mrp_phase_result_destroy(mrp_phase_result *r);

/* ---- device-resident merge --------------------------------------------------
 * The same results as mrp_get_rp_hmms / mrp_phase_reads, but the hmms stay in HBM across
 *   stRPHmm_createCrossProductOfTwoAlignedHmm (hmm.c:534) -> stRPHmm_forwardBackward (hmm.c:931)
 *   -> stRPHmm_prune (hmm.c:1160)
 * of every merge level (coordination.c:263-409); only per-column cell counts return to the host between
 * levels and only the final, pruned hmms are copied back.  Max-plus mode (maxNotSumTransitions, every
 * shipped parameter file) with at most 116 partitions per column; otherwise MRP_ERR_UNSUPPORTED
 * (mrp_get_rp_hmms_resident) or the per-chunk path is taken (mrp_phase_reads_many, stats->resident = 0). */
int mrp_get_rp_hmms_resident(mrp_context *ctx, const mrp_chunk *chunk, const mrp_read *reads, const int32_t *read_index,
                             int64_t n, const mrp_params *params, mrp_hmm ***hmms_out, int64_t *n_out);

typedef struct mrp_phase_many_stats {
    int32_t resident;  /* 1: the device-resident merge was used */
    int32_t fallback_chunks; /* chunks of a resident call that one of the kernels' checks sent to the per-chunk hashing path */
    int64_t levels, hmms, columns, cells, merge_cells; /* of the merge levels */
    double device_ms, cross_ms, sweep_ms, prune_ms;    /* summed HIP-event times of the merge levels */
    /* resident == 0: why the whole call left the device-resident path (log-sum-exp mode, more partitions per column than
     * its kernels keep, ...).  The hashing path it takes instead phases one chunk per host thread with a device sweep per merge
     * level and is two orders of magnitude slower; the first such call of a process also says so on stderr (MRP_QUIET=1
     * silences it). */
    char note[160];
    /* device_ms by kernel family (summed HIP-event times; the concurrent batches of a call add up): profile byte packing,
     * cross product + emission, forward/backward recursion, prune, compaction (+ the final level's trace back) */
    double pack_ms, cross_emit_ms, recursion_ms, prune_kernel_ms, compact_ms;
} mrp_phase_many_stats;

/* bubbleGraph_phaseBubbleGraph (bubbleGraph.c:2673-2801) for n_chunks independent chunks in one call: the
 * body of the chunk loop of phase.c:276-473.  Merge levels of all chunks (and both strands) are batched
 * into the same kernel launches.  out[n_chunks] receives one result per chunk; stats may be NULL.
 * Parameters outside the resident kernels' range (maxNotSumTransitions = false: the prune ranks doubles; more than 116
 * partitions per column, the reference's code default is 200, parser.c:22-23) do not fail: every chunk then goes through
 * mrp_phase_reads, one chunk per host thread -- results identical, throughput that of the hashing path; stats->resident = 0
 * and stats->note say so. */
int mrp_phase_reads_many(mrp_context *ctx, int64_t n_chunks, const mrp_chunk *const *chunks, const mrp_read *const *reads,
                         const int64_t *n_reads, const mrp_params *params, mrp_phase_result **out,
                         mrp_phase_many_stats *stats);

/* ---- the chunks of a node's worth of work over its GPUs ---------------------------------------------
 * Replaces the chunk loop of phase.c:276-473 together with its ordering (phase.c:257-263, chunks by estimated depth,
 * largest first) and its schedule ("#pragma omp parallel for schedule(dynamic,1)", :276-279), with devices in the role
 * of the threads: up to four host threads ("lanes", MRP_QUEUE_LANES) per entry of devices[] pull the next batch of chunks_per_batch
 * (0: the library's choice, cut by units -- one batch per device for a short queue, MRP_QUEUE_DEFAULT_BATCH = 192 yardstick chunks
 * otherwise) consecutive chunks of that order, phase it (mrp_phase_reads_many) and stores the results at the chunks' own positions of out[];
 * while a batch is phased the worker's NEXT batch is uploaded on a second stream (site tables + profile bytes, one wait per
 * batch).  Every worker has its own host thread pool (mrp_set_host_threads per device) and, when the queue drives more than
 * one device, runs on the CPUs next to its device (/sys/bus/pci/devices/.../local_cpulist; MRP_QUEUE_AFFINITY=0 turns that
 * off).  Chunks are independent until stitching: no collective, no traffic between devices.  A device may be listed more
 * than once (two workers sharing it). */
#define MRP_MAX_QUEUE_DEVICES 16
#define MRP_QUEUE_DEFAULT_BATCH 192 /* batch size when the caller passes 0 and the queue is longer than MRP_QUEUE_SHORT_CHUNKS chunks per device, both counted in chunks of
                                     * MRP_QUEUE_UNITS_PER_CHUNK units (a shorter queue is one batch per device; smaller batches at the end) */
#define MRP_QUEUE_SHORT_CHUNKS 1280 /* a queue of up to this many yardstick chunks per device is ONE call per device (round 4: 1 152 chunks 200 ms
                                     * as one call, 219 ms as eight batches on four lanes; 2 304 chunks: the same either way) */
#define MRP_QUEUE_UNITS_PER_CHUNK 60000 /* (read, het site) units of the 1 Mb, 30x chunk the batch sizes were measured on: the library's own batches
                                         * are cut by units, so that 3 000 chunks of 130 sites make the call that 192 chunks of 2 000 sites make */
typedef struct mrp_chunk_desc {     /* one genome chunk in host memory: what mrp_chunk_create and mrp_phase_reads take */
    int64_t n_sites;
    const uint32_t *allele_number;
    const uint16_t *substitution_log_probs, *allele_prior_log_probs; /* NULL = all zero */
    const uint8_t *profile_pool;
    int64_t pool_bytes;
    const mrp_read *reads;
    int64_t n_reads;
} mrp_chunk_desc;
typedef struct mrp_queue_stats {
    int32_t n_devices, reserved;
    int64_t batches, fallback_chunks;
    int64_t chunks_per_device[MRP_MAX_QUEUE_DEVICES];
    int64_t units_per_device[MRP_MAX_QUEUE_DEVICES];  /* het-sites x reads phased by each worker */
    double busy_ms_per_device[MRP_MAX_QUEUE_DEVICES]; /* host wall time each worker spent on its batches (upload included) */
} mrp_queue_stats;
typedef struct mrp_queue mrp_queue; /* the workers' contexts (device allocator caches, staging), kept from call to call */
int mrp_queue_create(const int32_t *devices, int32_t n_devices, mrp_queue **out);
void mrp_queue_destroy(mrp_queue *q);
int mrp_queue_phase_chunks(mrp_queue *q, int64_t n_chunks, const mrp_chunk_desc *chunks, const mrp_params *params, int64_t chunks_per_batch,
                           mrp_phase_result **out, mrp_queue_stats *stats);
/* create + phase + destroy */
int mrp_phase_chunks_on_devices(const int32_t *devices, int32_t n_devices, int64_t n_chunks, const mrp_chunk_desc *chunks,
                                const mrp_params *params, int64_t chunks_per_batch, mrp_phase_result **out, mrp_queue_stats *stats);
/* the order of the queue alone (host only): order_out[n_chunks] = chunk indices, largest cost first, ties in input
 * order; batch_of_chunk_out (optional) = the batch each chunk travels in */
int mrp_queue_plan(int64_t n_chunks, const int64_t *cost, int64_t chunks_per_batch, int64_t *order_out, int64_t *batch_of_chunk_out);
/* the queue itself with stand-in workers (host only; what the CPU test-suite drives): the plan and the hand-out
 * mrp_queue_phase_chunks uses for n_devices devices with `lanes` pulling threads each (the real queue: 4, MRP_QUEUE_LANES) --
 * same batches, same lanes in use, a lane takes its next batch when it starts on the current one -- but a call is replaced
 * by a sleep of usec_per_cost microseconds per unit of cost.  worker_of_chunk_out[i] = device * lanes + lane that took chunk
 * i, sequence_out[i] (optional) = the global position at which it was taken */
int mrp_queue_dry_run(int32_t n_devices, int32_t lanes, int64_t n_chunks, const int64_t *cost, int64_t chunks_per_batch, double usec_per_cost,
                      int32_t *worker_of_chunk_out, int64_t *sequence_out);

/* ---- the frame around the path: host code, no device needed ----------------------- */

/* BubbleGraph (inc/margin.h) as flat arrays: what bubbleGraph_getProfileSeqs / bubbleGraph_getReference read */
typedef struct mrp_bubbles {
    int64_t n_bubbles;
    const uint32_t *allele_number;     /* [n_bubbles] Bubble.alleleNo */
    const int64_t *read_off;           /* [n_bubbles + 1] prefix sum of Bubble.readNo */
    const int32_t *reads;              /* read index of Bubble.reads[j]->read */
    const int64_t *support_off;        /* [n_bubbles + 1] prefix sum of alleleNo * readNo */
    const float *allele_read_supports; /* per bubble Bubble.alleleReadSupports[readNo * k + j] */
} mrp_bubbles;

/* bubbleGraph_getReference (bubbleGraph.c:2443-2474): the site tables mrp_chunk_create takes (malloc'd, mrp_free) */
int mrp_reference_from_bubbles(const mrp_bubbles *bg, double het_substitution_probability, uint32_t **allele_number_out,
                               uint16_t **substitution_out, uint16_t **prior_out);
/* bubbleGraph_getProfileSeqs (bubbleGraph.c:2356-2441): one profile sequence per read that is in some bubble, in order
 * of first appearance, its bytes in one pool (mrp_read.pool_offset).  read_of_seq_out[s] = read index of sequence s. */
int mrp_profile_seqs_from_bubbles(const mrp_bubbles *bg, int64_t n_reads, const char *const *read_names,
                                  const int32_t *forward_strand, mrp_read **seqs_out, int32_t **read_of_seq_out,
                                  int64_t *n_seqs_out, uint8_t **pool_out, int64_t *pool_bytes_out);

/* stGenomeFragment_phaseBamChunkReads (genomeFragment.c:234-276): hap_out[r] = 1 / 2, 0 = in the fragment but below
 * minPhredScoreForHaplotypePartition, -1 = not in the fragment; phred_out (optional) the score of :260 */
int mrp_assign_reads_to_haplotypes(int64_t n_sites, const uint32_t *allele_number, const uint8_t *profile_pool,
                                   const mrp_read *reads, int64_t n_reads, const mrp_phase_result *gf, int64_t min_phred,
                                   int8_t *hap_out, double *phred_out);

/* chunkToStitch_phaseAdjacentChunks (stitching.c:345-403): the read sets seen so far per haplotype, and the decision
 * whether the next chunk keeps its relative phasing (cis) or is switched (trans).  counts = cisH1, cisH2, transH1, transH2 */
typedef struct mrp_stitch mrp_stitch;
int mrp_stitch_create(mrp_stitch **out);
void mrp_stitch_destroy(mrp_stitch *s);
int mrp_stitch_chunk(mrp_stitch *s, int64_t n1, const char *const *names1, const double *probs1, int64_t n2,
                     const char *const *names2, const double *probs2, int primary_reads_only, int do_not_switch,
                     int *switched, int64_t counts[4]);
int64_t mrp_stitch_size(const mrp_stitch *s, int hap);
int mrp_stitch_lookup(const mrp_stitch *s, int hap, const char *name, double *prob);

/* writePhasedVcf's phase set rules (vcf.c:869-953) over the variants of one contig that margin updated */
typedef struct mrp_variant {
    int32_t pos;                    /* VcfEntry.refPos */
    int32_t gt1, gt2;               /* called genotype (allele indices) */
    int32_t n_alleles;
    const int64_t *allele_read_off; /* [n_alleles + 1] */
    const int32_t *allele_reads;    /* VcfEntry.alleleIdxToReads, read ids */
} mrp_variant;
#define MRP_PS_SAME 0                 /* stays in the current phase set */
#define MRP_PS_NO_HET 1               /* "NoHet" */
#define MRP_PS_MISSING_CONCORDANCY 2  /* "MissingConcordancy" */
#define MRP_PS_UNLIKELY_CONCORDANCY 3 /* "UnlikelyConcordancy" */
#define MRP_PS_DISCORDANCY 4          /* "Discordancy" */
int mrp_phase_sets(int64_t n_variants, const mrp_variant *v, int64_t min_spanning_reads, double min_binomial_read_split_likelihood,
                   double max_discordant_ratio, int32_t *phase_set_out, int32_t *reason_out);
double mrp_binomial_p_value(int64_t n, int64_t k); /* bubbleGraph.c:2876-2883 */
/* bionomialCoefficient (bubbleGraph.c:2860-2874): the unsigned 128-bit value as hi:lo (either may be NULL) and, returned,
 * converted to double as the reference's test reads it (tests/polisherTest.c:957-963); 0 for k outside [0, n] */
double mrp_binomial_coefficient(int64_t n, int64_t k, uint64_t *hi, uint64_t *lo);

/* ---------------------------------------------------------------------------------------------------------------
 * Read x allele alignment likelihoods: the banded pair-HMM forward probability that fills
 * Bubble.alleleReadSupports (bubbleGraph.c:1421-1464 -> computeForwardProbability, pairwiseAligner.c:849-903).
 * Device work: one kernel with a pair per lane for short x strings, one with a pair per wave for everything else.
 * The arithmetic is the reference's: fp64 + and * only (logAdd is a cubic interpolation, pairwiseAligner.c:279-299),
 * evaluated without fused multiply-add, so results are bit-identical to a strict IEEE build of the reference.
 * Symbols: 0..3 = A C G T, >= 4 = N (convertNucleotideCharToSymbol, stateMachine.c:25-42).  Nucleotide emissions
 * only; the run-length (repeat count) emissions of margin polish (stateMachine.c:722-753) are out of scope.
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct mrp_pair_hmm { /* struct _StateMachine3 (stateMachine.c:507-519) + NucleotideEmissions, all in log space */
    double match_continue, match_from_gap_x, match_from_gap_y, gap_open_x, gap_open_y, gap_extend_x, gap_extend_y,
            gap_switch_to_x, gap_switch_to_y;
    double e_match[16]; /* [x * 4 + y] */
    double e_gap_x[4], e_gap_y[4];
} mrp_pair_hmm;

typedef struct mrp_pairhmm_stats {
    int64_t pairs_lane, pairs_wave; /* pairs handled by the pair-per-lane / pair-per-wave kernel */
    int64_t cells;                  /* dp cells inside the bands, (0,0) included */
    double kernel_ms;               /* HIP events around the launches */
    double total_ms;                /* host wall time of the call */
} mrp_pairhmm_stats;

/* symbol_convertStringToSymbols (stateMachine.c:84-92) with the nucleotide alphabet */
void mrp_symbols_from_chars(const char *s, int64_t n, uint8_t *out);
/* nucleotideEmissions_reverseComplement (stateMachine.c:457-473): the state machine of reverse strand reads (parser.c:356) */
void mrp_pair_hmm_reverse_complement(mrp_pair_hmm *m);
/* band_construct (pairwiseAligner.c:175-226): xmy_l / xmy_r of the lx + ly + 1 diagonals; anchors = n_anchors (x, y)
 * sequence coordinates, strictly increasing in both.  MRP_ERR_ARG where the reference asserts / throws. */
int mrp_band_diagonals(const int64_t *anchors, int64_t n_anchors, int64_t lx, int64_t ly, int64_t expansion, int32_t *xmy_l,
                       int32_t *xmy_r);
/* getKmerAlignmentAnchors (pairwiseAligner.c:1519-1627, KMER_SIZE = 20): the chain of shared 20-mers the reference anchors
 * long (structural variant) alleles with; out receives at most ly - 19 (x, y) pairs, the count is returned (host only) */
int64_t mrp_kmer_alignment_anchors(const uint8_t *x, int64_t lx, const uint8_t *y, int64_t ly, int64_t *out);
/* computeForwardProbability for n_pairs (x, y) string pairs stored in one pool of symbols.  model_index (NULL: all 0)
 * selects models[i] per pair; anchor_off (NULL: no anchors anywhere) holds n_pairs + 1 offsets into anchors (pairs of
 * int64).  A pair without anchors covers its whole matrix, as in the reference.  out[i] = log probability (0.0 for two
 * empty strings, :860-862).  Limits: a diagonal of at most 2 048 cells for pairs that go to the pair-per-wave kernel
 * (x longer than 100 symbols, or anchored); beyond that MRP_ERR_UNSUPPORTED.  stats may be NULL. */
int mrp_forward_probabilities(mrp_context *ctx, const mrp_pair_hmm *models, int32_t n_models, int64_t n_pairs, const uint8_t *pool,
                              int64_t pool_bytes, const int64_t *x_off, const int32_t *x_len, const int64_t *y_off,
                              const int32_t *y_len, const uint8_t *model_index, const int64_t *anchor_off, const int64_t *anchors,
                              int64_t expansion, int ragged_left, int ragged_right, double *out, mrp_pairhmm_stats *stats);
/* The alleleReadSupports loop of bubbleGraph.c:1421-1464 for n_bubbles bubbles.  Bubble b owns alleles
 * [allele_first[b], allele_first[b+1]) and read substrings [read_first[b], read_first[b+1]); x = allele, y = read
 * substring, the read's strand picks the state machine -- except that, as in the reference (cachedScores is keyed by the
 * substring alone), a read whose substring equals that of an earlier read of the bubble copies that read's scores.
 * support is the concatenation over bubbles of float[alleleNo * readNo], entry j * readNo + k.  A pair whose allele or
 * read substring is longer than sv_threshold (referenceExpansionForStructuralVariants, 512 in the shipped parameters) is
 * banded around its k-mer anchors (:1448-1451), every other pair covers its whole matrix. */
int mrp_allele_read_supports(mrp_context *ctx, const mrp_pair_hmm *forward_model, const mrp_pair_hmm *reverse_model, int64_t n_bubbles,
                             const int64_t *allele_first, const int64_t *read_first, const uint8_t *pool, int64_t pool_bytes,
                             const int64_t *allele_off, const int32_t *allele_len, const int64_t *read_off, const int32_t *read_len,
                             const uint8_t *read_forward_strand, int64_t expansion, int64_t sv_threshold, float *support,
                             mrp_pairhmm_stats *stats);

#ifdef __cplusplus
}
#endif
#endif /* MARGIN_RPHMM_H_ */
