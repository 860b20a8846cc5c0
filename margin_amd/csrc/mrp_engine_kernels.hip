/*
 * mrp_engine_kernels.hip -- gfx950 kernels of the device-resident merge level.
 *
 *  mrp_cross_kernel    stRPHmm_createCrossProductOfTwoAlignedHmm (hmm.c:534-750) in closed form.
 *                      The reference builds each column by hashing merged partitions in (c1, c2)
 *                      row-major order and, with includeInvertedPartitions, appending the complement
 *                      right after every new partition (hmm.c:627-655; merge cells :686-740).  When the
 *                      two parents keep their cells / merge cells in adjacent complement pairs (they do
 *                      by construction: hmm.c:97-133 builds {1, 0}, the prune keeps both or neither of a
 *                      pair and its sort is stable) the resulting order is a pure function of the index
 *                      pair, so a cell's partition and the indices of the merge cells it feeds / is fed
 *                      by are computed without any hash table.  The kernel VERIFIES the pair order of
 *                      every parent column it reads and raises MRP_ENGINE_ERR_STRUCTURE otherwise (the
 *                      host then redoes the chunk through the hashing path of rphmm_host.c).
 *  mrp_prune_kernel    stRPHmm_prune (hmm.c:1049-1163): pruneForwards walks the columns keeping the
 *                      best linked cells and merge cells by posterior (stable order on ties),
 *                      pruneBackwards removes what became unreachable.  Max-plus mode only: the
 *                      posterior exp(f + b - total) is monotone in the integer f + b - total, which
 *                      is what is ranked (bins; everything at or below the underflow point of exp
 *                      shares the last bin, exactly as equal doubles tie in the reference).
 *  mrp_compact_kernel  filterMergeCells / relinkCells (hmm.c:964-1019): writes the pruned hmm in the
 *                      fixed-stride resident layout (mrp_engine.h).
 *  mrp_traceback_kernel stRPHmm_forwardTraceBack (hmm.c:165-219) on the final hmm of a chunk: one wave per hmm walks
 *                      the columns from the last to the first and returns one partition per column.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "mrp_engine.h"
#include "../../include/margin_rphmm.h"

#define WAVE 64

#define K_AS(T) const __attribute__((address_space(4))) T *
#define K_PTR(T, p) ((K_AS(T)) (p))
template <typename T>
static __device__ __forceinline__ T k_load(const T *p) {
    static_assert(sizeof(T) % 4 == 0, "dword sized");
    T v;
    K_AS(uint32_t) s = K_PTR(uint32_t, p);
    uint32_t *dst = reinterpret_cast<uint32_t *>(&v);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; i++) dst[i] = s[i];
    return v;
}

/* Workgroup barrier that orders LDS traffic only (__syncthreads() also waits for every global load in
 * flight, which would serialize the prefetch of the next column with the work on the current one). */
static __device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
static __device__ __forceinline__ uint64_t accept_mask(uint32_t depth) { /* partitions.c:13-19 */
    return depth < 64 ? ~(0xFFFFFFFFFFFFFFFFull << depth) : 0xFFFFFFFFFFFFFFFFull;
}
static __device__ __forceinline__ uint32_t lanemask_lt_count(uint64_t m, int lane) {
    return (uint32_t) __popcll(m & ((1ull << lane) - 1ull));
}

/* Position of the pair (i, j) in the list the reference builds by visiting pairs in row-major order
 * and appending the complement pair right after each new one.  i indexes side A (Ma entries),
 * j side B (Mb entries); a side is "paired" when its entries come as (x, complement of x) at (2t, 2t+1),
 * otherwise it has exactly one, self-complementary entry. */
static __device__ __forceinline__ uint32_t pair_index(uint32_t i, uint32_t j, uint32_t Mb, bool inv, bool a_paired,
                                                      bool b_paired) {
    if (!inv) return i * Mb + j;
    if (!a_paired) return j;
    const uint32_t pj = b_paired ? (j ^ 1u) : j;
    return (i & 1u) ? (i - 1u) * Mb + 2u * pj + 1u : i * Mb + 2u * j;
}

/* ------------------------------------------------------------------------------------------ */
/* cross product                                                                               */
/* ------------------------------------------------------------------------------------------ */
static __device__ int verify_side(const uint64_t *part, const uint32_t *np, uint32_t C, uint32_t depth, uint32_t M_out,
                                  uint32_t M_in, uint32_t out_kind, uint32_t in_kind, bool inv, bool out_paired,
                                  bool in_paired) {
    if (!part) return 0;
    int bad = 0;
    const bool cells_paired = inv && depth > 0;
    if (cells_paired && (C & 1u)) return MRP_ENGINE_ERR_STRUCTURE;
    const uint64_t acc = accept_mask(depth);
    for (uint32_t e = threadIdx.x; e < C; e += blockDim.x) {
        const uint32_t v = np[e], nx = v & 0xFFFFu, pv = v >> 16;
        uint32_t vo = v;
        if (cells_paired) {
            if (part[e ^ 1u] != (~part[e] & acc)) bad |= MRP_ENGINE_ERR_STRUCTURE;
            vo = np[e ^ 1u];
        }
        if (out_kind == MRP_CONN_REAL) {
            if (nx >= M_out) bad |= MRP_ENGINE_ERR_RANGE;
            if (inv) {
                if (out_paired) { if ((M_out & 1u) || (vo & 0xFFFFu) != (nx ^ 1u)) bad |= MRP_ENGINE_ERR_STRUCTURE; }
                else if (M_out != 1u) bad |= MRP_ENGINE_ERR_STRUCTURE;
            }
        }
        if (in_kind == MRP_CONN_REAL) {
            if (pv >= M_in) bad |= MRP_ENGINE_ERR_RANGE;
            if (inv) {
                if (in_paired) { if ((M_in & 1u) || (vo >> 16) != (pv ^ 1u)) bad |= MRP_ENGINE_ERR_STRUCTURE; }
                else if (M_in != 1u) bad |= MRP_ENGINE_ERR_STRUCTURE;
            }
        }
    }
    return bad;
}

__global__ void __launch_bounds__(256) mrp_cross_kernel(const CrossCol *__restrict__ cols, int64_t n_cols,
                                                        uint64_t *__restrict__ partition, uint32_t *__restrict__ cell_np,
                                                        int32_t *__restrict__ err) {
    for (int64_t col = blockIdx.x; col < n_cols; col += gridDim.x) {
        const CrossCol c = k_load(cols + col);
        const bool inv = (c.flags & MRP_XF_INVERTED) != 0;
        const uint32_t C1 = c.C1, C2 = c.C2, C = C1 * C2;
        const bool a_cells_paired = inv && c.a_part && c.d1 > 0, b_cells_paired = inv && c.b_part && c.d2 > 0;
        int bad = verify_side(c.a_part, c.a_np, C1, c.d1, c.Ma, c.Pa, c.out_a, c.in_a, inv,
                              (c.flags & MRP_XF_OUT_A_PAIRED) != 0, (c.flags & MRP_XF_IN_A_PAIRED) != 0);
        bad |= verify_side(c.b_part, c.b_np, C2, c.d2, c.Mb, c.Pb, c.out_b, c.in_b, inv,
                           (c.flags & MRP_XF_OUT_B_PAIRED) != 0, (c.flags & MRP_XF_IN_B_PAIRED) != 0);
        if ((!c.a_part && C1 != 1u) || (!c.b_part && C2 != 1u)) bad |= MRP_ENGINE_ERR_RANGE;
        if (bad) atomicOr(err, bad);
        if (__syncthreads_or(bad)) { /* the level is discarded by the host; keep the arrays defined meanwhile */
            for (uint32_t e = threadIdx.x; e < C; e += blockDim.x) {
                partition[c.x_cell_off + e] = 0ull;
                cell_np[c.x_cell_off + e] = 0u;
            }
            continue;
        }
        for (uint32_t e = threadIdx.x; e < C; e += blockDim.x) {
            uint32_t c1, c2;
            if (!inv) { c1 = e / C2; c2 = e - c1 * C2; }
            else if (!a_cells_paired) { c1 = 0; c2 = e; }
            else {
                const uint32_t r = e / (2u * C2), t = e - r * 2u * C2, h = t >> 1;
                if (t & 1u) { c1 = 2u * r + 1u; c2 = b_cells_paired ? (h ^ 1u) : h; }
                else { c1 = 2u * r; c2 = h; }
            }
            const uint64_t p1 = c.a_part ? c.a_part[c1] : 0ull, p2 = c.b_part ? c.b_part[c2] : 0ull;
            const uint32_t n1 = c.a_part ? c.a_np[c1] : 0u, n2 = c.b_part ? c.b_np[c2] : 0u;
            uint32_t nxt = 0, prv = 0;
            if (c.out_a != MRP_CONN_NONE) {
                const uint32_t i = c.out_a == MRP_CONN_REAL ? (n1 & 0xFFFFu) : (c.out_a == MRP_CONN_IDENT ? c1 : 0u);
                const uint32_t j = c.out_b == MRP_CONN_REAL ? (n2 & 0xFFFFu) : (c.out_b == MRP_CONN_IDENT ? c2 : 0u);
                nxt = pair_index(i, j, c.Mb, inv, (c.flags & MRP_XF_OUT_A_PAIRED) != 0, (c.flags & MRP_XF_OUT_B_PAIRED) != 0);
            }
            if (c.in_a != MRP_CONN_NONE) {
                const uint32_t i = c.in_a == MRP_CONN_REAL ? (n1 >> 16) : (c.in_a == MRP_CONN_IDENT ? c1 : 0u);
                const uint32_t j = c.in_b == MRP_CONN_REAL ? (n2 >> 16) : (c.in_b == MRP_CONN_IDENT ? c2 : 0u);
                prv = pair_index(i, j, c.Pb, inv, (c.flags & MRP_XF_IN_A_PAIRED) != 0, (c.flags & MRP_XF_IN_B_PAIRED) != 0);
            }
            partition[c.x_cell_off + e] = c.d1 < 64 ? (p1 | (p2 << c.d1)) : p1; /* mergePartitionsOrMasks partitions.c:21-28 */
            cell_np[c.x_cell_off + e] = nxt | (prv << 16);
        }
    }
}

hipError_t mrp_launch_cross(const CrossCol *cols_dev, int64_t n_cols, uint64_t *partition, uint32_t *cell_np, int32_t *err,
                            hipStream_t stream) {
    if (n_cols <= 0) return hipSuccess;
    const int64_t grid = n_cols < 65536 ? n_cols : 65536;
    hipLaunchKernelGGL(mrp_cross_kernel, dim3((unsigned) grid), dim3(256), 0, stream, cols_dev, n_cols, partition, cell_np, err);
    return hipGetLastError();
}

/* ------------------------------------------------------------------------------------------ */
/* prune                                                                                       */
/* ------------------------------------------------------------------------------------------ */
/* profiling aid: build with -DPRUNE_EXP_CLOCK to sum the shader cycles wave 0 spends in each section of a column
 * (read back and printed by mrp_engine.cpp; profiles/r01/prune_sections_v3.txt) */
#ifdef PRUNE_EXP_CLOCK
#define CLK(slot) do { const uint64_t t_ = __builtin_amdgcn_s_memtime(); if (wave == 0) clk[slot] += t_ - tlast; tlast = t_; } while (0)
#else
#define CLK(slot) do { } while (0)
#endif
#define PRUNE_CPT 16 /* cells per lane held in registers: a column has at most 16 * (threads of the workgroup) cells */

/* n kept of n_link candidates whose first g pass the posterior threshold: the loop of hmm.c:1073-1079 /
 * :1094-1100 ("while n > min && (n > max || last.posterior < threshold) drop last") in closed form */
static __device__ __forceinline__ int kept_count(int n_link, int g, int min_p, int max_p) {
    if (n_link <= min_p) return n_link;
    int n = g < max_p ? g : max_p;
    return n > min_p ? n : min_p;
}

static __device__ __forceinline__ int posterior_bin(int32_t f, int32_t b, int64_t total, int n_bins, int *errbits) {
    if (f == MRP_NEG_I32 || b == MRP_NEG_I32) return n_bins - 1; /* exp(-inf) = 0 */
    const int64_t s = total - (int64_t) f - (int64_t) b;
    if (s < 0) { *errbits |= MRP_ENGINE_ERR_POSTERIOR; return 0; }
    return s < n_bins - 1 ? (int) s : n_bins - 1;
}

/* Cross-lane primitives of the single-wave section, on the DPP path where gfx950 has one (tools/ubench/dpp_prims.hip
 * checks them against the __shfl versions and times them: bitonic128 0.58 us vs 1.10 us, scan 0.07 vs 0.20 us). */
template <int CTRL, int ROWMASK = 0xf>
static __device__ __forceinline__ uint32_t dpp_mov(uint32_t x) {
    return (uint32_t) __builtin_amdgcn_update_dpp(0, (int) x, CTRL, ROWMASK, 0xf, false);
}
static __device__ __forceinline__ int wave_incl_scan(int v, int lane) {
    int t;
    t = (int) dpp_mov<0x111>((uint32_t) v) + v; if ((lane & 15) >= 1) v = t;       /* row_shr:1 */
    t = (int) dpp_mov<0x112>((uint32_t) v) + v; if ((lane & 15) >= 2) v = t;
    t = (int) dpp_mov<0x114>((uint32_t) v) + v; if ((lane & 15) >= 4) v = t;
    t = (int) dpp_mov<0x118>((uint32_t) v) + v; if ((lane & 15) >= 8) v = t;
    t = (int) dpp_mov<0x142, 0xa>((uint32_t) v) + v; if ((lane & 31) >= 16) v = t; /* row_bcast:15 */
    t = (int) dpp_mov<0x143, 0xc>((uint32_t) v) + v; if (lane >= 32) v = t;        /* row_bcast:31 */
    return v;
}
template <int J>
static __device__ __forceinline__ uint32_t lane_xor(uint32_t x, int lane) {
    if (J == 1) return dpp_mov<0xB1>(x); /* quad_perm [1,0,3,2] */
    if (J == 2) return dpp_mov<0x4E>(x); /* quad_perm [2,3,0,1] */
    if (J == 4) { const uint32_t a = dpp_mov<0x104>(x), b = dpp_mov<0x114>(x); return (lane & 4) ? b : a; } /* row_shl:4 / row_shr:4 */
    if (J == 8) { const uint32_t a = dpp_mov<0x108>(x), b = dpp_mov<0x118>(x); return (lane & 8) ? b : a; }
    return (uint32_t) __shfl_xor((int) x, J, WAVE);
}
/* Ascending bitonic sort of 128 distinct keys held two per lane (index lane and lane + 64) by one wave. */
template <int K, int J>
static __device__ __forceinline__ void bitonic_step(uint32_t &k0, uint32_t &k1, int lane) {
    if (J == 64) { /* K == 128: the partner is the lane's other key */
        const uint32_t lo = k0 < k1 ? k0 : k1, hi = k0 < k1 ? k1 : k0;
        k0 = lo; k1 = hi;
    } else {
        const uint32_t p0 = lane_xor<J>(k0, lane), p1 = lane_xor<J>(k1, lane);
        const bool lower = (lane & J) == 0;
        const bool asc0 = (lane & K) == 0, asc1 = ((lane + 64) & K) == 0;
        const uint32_t mn0 = k0 < p0 ? k0 : p0, mx0 = k0 < p0 ? p0 : k0;
        const uint32_t mn1 = k1 < p1 ? k1 : p1, mx1 = k1 < p1 ? p1 : k1;
        k0 = (lower == asc0) ? mn0 : mx0;
        k1 = (lower == asc1) ? mn1 : mx1;
    }
}
template <int K>
static __device__ __forceinline__ void bitonic_merge(uint32_t &k0, uint32_t &k1, int lane) {
    if (K >= 128) bitonic_step<K, 64>(k0, k1, lane);
    if (K >= 64) bitonic_step<K, 32>(k0, k1, lane);
    if (K >= 32) bitonic_step<K, 16>(k0, k1, lane);
    if (K >= 16) bitonic_step<K, 8>(k0, k1, lane);
    if (K >= 8) bitonic_step<K, 4>(k0, k1, lane);
    if (K >= 4) bitonic_step<K, 2>(k0, k1, lane);
    bitonic_step<K, 1>(k0, k1, lane);
}
static __device__ __forceinline__ void wave_bitonic_sort128(uint32_t &k0, uint32_t &k1, int lane) {
    bitonic_merge<2>(k0, k1, lane);
    bitonic_merge<4>(k0, k1, lane);
    bitonic_merge<8>(k0, k1, lane);
    bitonic_merge<16>(k0, k1, lane);
    bitonic_merge<32>(k0, k1, lane);
    bitonic_merge<64>(k0, k1, lane);
    bitonic_merge<128>(k0, k1, lane);
}

/* One workgroup per hmm.  Per column there are two phases separated by a barrier each:
 *   [A] all waves: the column's cells (np, f, b were loaded into registers while the previous column was
 *       processed) are tested against the kept flags of the previous merge column, binned by posterior and
 *       appended, in list order, to the candidate list in LDS;
 *   [B] wave 0 alone, without further barriers: cutoff bin from the histogram, ordered selection of the <= S kept
 *       cells, kept flags of their next merge cells -- all the next column's [A] needs.  Everything else a column
 *       produces (stable sort of the kept cells, distinct next merge cells in order of first use, their posteriors,
 *       stable sort, lists to HBM) is done one and two columns later by waves 1 and 2, beside wave 0's [B] of the
 *       columns that follow, from double-buffered copies of the selection.  The loads of the next column are already in flight. */
/* the few arrays of the level's batch the prune kernel reads (the whole MrpBatchDev by value costs ~60 SGPRs) */
struct PruneIn {
    const SweepCol *scols;
    const uint32_t *cell_np;
    const int32_t *cell_f32, *cell_b32, *merge_f32, *merge_b32;
    const double *hmm_fb;
};

/* buffer descriptor over [p, p + bytes): both made wave-uniform for the compiler (cdna_hip_programming.md T8 / T20) */
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t prune_rsrc(const void *p, int bytes) {
    const uint64_t a = (uint64_t) p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t) a), hi = __builtin_amdgcn_readfirstlane((uint32_t) (a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *) (((uint64_t) hi << 32) | lo), 0, bytes, 0x00020000);
}

template <int T>
__global__ void __launch_bounds__(T) mrp_prune_kernel(PruneIn d, const PruneHmm *__restrict__ hmms, int64_t n_hmms,
                                                      PruneParams p, PruneScratch sc) {
    constexpr int W = T / WAVE;
    extern __shared__ uint32_t lds[];
    const int S = p.S;
    const int nb = p.n_bins;
    const int nb_r = 1024; /* 16 bins per lane of the cutoff search: bin b lives at (b & 15) * 64 + (b >> 4) */
    const int cap_c = ((p.max_cells > p.max_merge ? p.max_cells : p.max_merge) + 3) & ~3;
    /* LDS layout (dwords) */
    /* the selection of a column lives in one of two buffers: wave 0 fills buffer k & 1 while wave 1 turns buffer (k - 1) & 1
     * into the sorted lists of column k - 1 */
    uint32_t *sel = lds;              /* [2][4][S]: gsel (bin << 16 | cell, above the cutoff bin), gnp (next | prev << 16), esel (cell, in the cutoff bin), enp */
    uint32_t *um = sel + 8 * S;       /* [S] posterior bin of the merge cell each selected cell leads to (selection order); wave 1 */
    uint32_t *s1 = um + S;            /* [2][2][S] stage 1 -> stage 2: next | prev and merge posterior bin per sorted kept cell */
    uint32_t *sh = s1 + 4 * S;        /* [64] per-wave counters [0, W); n, nG of the two selection buffers at [32, 36); n of the stage-1 buffers at [36, 38) */
    uint32_t *stg = sh + 64;          /* [W][4][64] staging of linked cells: cell, transitions, f, b */
    uint32_t *hist = stg + W * 4 * WAVE; /* [2][nb_r] */
    uint32_t *cand = hist + 2 * nb_r; /* [cap_c] linked cells of the column, list order per wave segment: bin << 16 | cell */
    uint32_t *cand_np = cand + cap_c; /* [cap_c] */
    uint32_t *htab_key = cand_np + cap_c; /* [256] merge cell -> first kept cell that uses it (open addressing); wave 1 */
    uint32_t *htab_val = htab_key + 256;  /* [256] */
    uint8_t *flags = reinterpret_cast<uint8_t *>(htab_val + 256); /* [max_merge] kept flag per merge cell */

    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
    int errbits = 0;
#ifdef PRUNE_EXP_CLOCK
    uint64_t clk[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t tlast = __builtin_amdgcn_s_memtime();
#endif

    for (int64_t hi_ = blockIdx.x; hi_ < n_hmms; hi_ += gridDim.x) {
        const PruneHmm h = k_load(hmms + hi_);
        const int K = h.n_cols;
        const int64_t total = (int64_t) d.hmm_fb[2 * h.hmm_index]; /* max mode: the same integer for every column */
        for (int i = tid; i < 2 * nb_r; i += T) hist[i] = 0;
        if (tid < 2) sh[40 + tid] = 0u;
        for (int i = tid; i < (p.max_merge + 3) / 4; i += T) reinterpret_cast<uint32_t *>(flags)[i] = 0;
        int n_prev = 0, nG_prev = 0; /* wave 0: the selection whose next merge cells own the kept flags */
        int64_t mcell_prev = 0;       /* first merge cell of the merge column after the previous column */

        uint32_t r_np[PRUNE_CPT];
        int32_t r_f[PRUNE_CPT], r_b[PRUNE_CPT];
        SweepCol col = k_load(d.scols + h.col0);
        SweepCol col_next = K > 1 ? k_load(d.scols + h.col0 + 1) : col;
        /* this wave's share of the column: [lo, hi), 64 cells per step */
#define PRUNE_SHARE(colv, lo_, hi_v, nj_)                                                     \
        const int per_##nj_ = (((colv).n_cells + W - 1) / W + 63) & ~63;                      \
        const int lo_ = wave * per_##nj_ < (colv).n_cells ? wave * per_##nj_ : (colv).n_cells; \
        const int hi_v = lo_ + per_##nj_ < (colv).n_cells ? lo_ + per_##nj_ : (colv).n_cells; \
        const int nj_ = (hi_v - lo_ + 63) >> 6;
#define PRUNE_LOAD(colv, lo_, hi_v, nj_)                                                      \
        {                                                                                     \
            /* buffer loads: one descriptor per array over this wave's share, the hardware's range check instead of an \
             * exec mask per load (out of range reads 0 and is never looked at), immediate offsets */ \
            const int64_t first_ = (colv).cell_off + lo_;                                     \
            const int bytes_ = __builtin_amdgcn_readfirstlane((hi_v - lo_) * 4);              \
            const auto rn_ = prune_rsrc(d.cell_np + first_, bytes_);                          \
            const auto rf_ = prune_rsrc(d.cell_f32 + first_, bytes_);                         \
            const auto rb_ = prune_rsrc(d.cell_b32 + first_, bytes_);                         \
            _Pragma("unroll") for (int j = 0; j < PRUNE_CPT; j++) {                           \
                if (j < nj_) { /* wave-uniform */                                              \
                    r_np[j] = (uint32_t) __builtin_amdgcn_raw_buffer_load_b32(rn_, lane * 4 + j * WAVE * 4, 0, 0); \
                    r_f[j] = (int32_t) __builtin_amdgcn_raw_buffer_load_b32(rf_, lane * 4 + j * WAVE * 4, 0, 0);   \
                    r_b[j] = (int32_t) __builtin_amdgcn_raw_buffer_load_b32(rb_, lane * 4 + j * WAVE * 4, 0, 0);   \
                }                                                                             \
            }                                                                                 \
        }
        {
            PRUNE_SHARE(col, lo0, hi0, nj0)
            (void) nj0;
            PRUNE_LOAD(col, lo0, hi0, nj0)
        }
        __syncthreads();

        /* The sorted lists of a finished selection, in two stages one column apart, so that neither is longer than wave 0's
         * part of a column.  Nothing later in the forward pass reads them: the kept flags were already set by wave 0 from
         * the unsorted selection.
         * Stage 1 (wave 1, column kk from selection buffer kk & 1): stable sort of the kept cells, kept lists to HBM, the
         * posterior bins of the merge cells they lead to; leaves next | prev and that bin per sorted kept cell in LDS. */
        auto lists_stage1 = [&](int kk, int64_t mcell_off) {
            const int b = kk & 1;
            const uint32_t *gsel = sel + b * 4 * S, *gnp = gsel + S, *esel = gnp + S, *enp = esel + S;
            uint32_t *snp = s1 + b * 2 * S, *sbin = snp + S;
            const int n = (int) sh[32 + 2 * b], nG = (int) sh[33 + 2 * b];
            const int64_t lcol = h.col0 + kk;
            const bool has_merge = kk + 1 < K;
            uint32_t key[2], my_np[2], my_c[2], my_src[2];
            int32_t pre_mf[2] = {0, 0}, pre_mb[2] = {0, 0};
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int i = lane + u * WAVE;
                /* stable descending sort of the kept cells (stList_sort :1071): smaller bin = larger posterior; the key
                 * bin | list position | cell is unique, so a bitonic sort in registers is stable by construction */
                key[u] = i < nG ? ((gsel[i] >> 16) << 21) | ((uint32_t) i << 14) | (gsel[i] & 0x3FFFu) : 0xFFFFFFFFu;
                /* the posteriors of the merge cells the selected cells lead to are requested now, for the selection in
                 * its unsorted order, and consumed after the sort (slot i of the selection = um[i] below) */
                if (has_merge && i < n) {
                    const uint32_t m = (i < nG ? gnp[i] : enp[i - nG]) & 0xFFFFu;
                    pre_mf[u] = d.merge_f32[mcell_off + m];
                    pre_mb[u] = d.merge_b32[mcell_off + m];
                }
            }
            wave_bitonic_sort128(key[0], key[1], lane);
            if (has_merge) {
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int i = lane + u * WAVE;
                    if (i < n) um[i] = (uint32_t) posterior_bin(pre_mf[u], pre_mb[u], total, nb, &errbits);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int i = lane + u * WAVE;
                if (i < nG) { my_c[u] = key[u] & 0x3FFFu; my_np[u] = gnp[(key[u] >> 14) & 0x7Fu]; my_src[u] = (key[u] >> 14) & 0x7Fu; }
                else if (i < n) { my_c[u] = esel[i - nG]; my_np[u] = enp[i - nG]; my_src[u] = (uint32_t) i; }
                else { my_c[u] = 0u; my_np[u] = 0u; my_src[u] = 0u; }
                if (i < n) {
                    sc.kept[lcol * S + i] = (uint16_t) my_c[u];
                    sc.kept_np[lcol * S + i] = my_np[u];
                    snp[i] = my_np[u];
                    if (has_merge) sbin[i] = um[my_src[u]];
                }
            }
            if (lane == 0) { sc.n_kept[lcol] = n; sh[36 + b] = (uint32_t) n; }
        };
        /* Stage 2 (wave 2, one column later): distinct next merge cells in order of first use, stable sort by posterior,
         * kept merge list to HBM. */
        auto lists_stage2 = [&](int kk) {
            const int b = kk & 1;
            const uint32_t *snp = s1 + b * 2 * S, *sbin = snp + S;
            const int n = (int) sh[36 + b];
            const int64_t lcol = h.col0 + kk;
            int mn = 0;
            if (kk + 1 < K) {
                /* getLinkedMergeCells :989-1004: distinct next merge cells in order of first use (sorted order of the
                 * kept cells): a 256-slot open-addressing table, merge cell -> smallest sorted index that uses it */
                for (int i = lane; i < 256; i += WAVE) { htab_key[i] = 0xFFFFFFFFu; htab_val[i] = 0xFFFFFFFFu; }
                uint32_t my_m[2] = {0u, 0u};
                int slot[2] = {0, 0};
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int i = lane + u * WAVE;
                    if (i < n) {
                        const uint32_t m = snp[i] & 0xFFFFu;
                        my_m[u] = m;
                        int q = (int) ((m * 2654435761u) >> 24);
                        for (;;) {
                            const uint32_t prev = atomicCAS(&htab_key[q], 0xFFFFFFFFu, m);
                            if (prev == 0xFFFFFFFFu || prev == m) break;
                            q = (q + 1) & 255;
                        }
                        slot[u] = q;
                        atomicMin(&htab_val[q], (uint32_t) i);
                    }
                }
                bool first[2];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int i = lane + u * WAVE;
                    first[u] = i < n && htab_val[slot[u]] == (uint32_t) i;
                }
                const uint64_t f0 = __ballot(first[0]), f1 = __ballot(first[1]);
                const int mnl = __popcll(f0) + __popcll(f1);
                uint32_t mkey[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
                int pass_thr[2] = {0, 0};
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    if (first[u]) {
                        const int pos = u == 0 ? (int) lanemask_lt_count(f0, lane) : __popcll(f0) + (int) lanemask_lt_count(f1, lane);
                        const int bin = (int) sbin[lane + u * WAVE];
                        pass_thr[u] = bin <= p.thr_bin ? 1 : 0;
                        mkey[u] = ((uint32_t) bin << 21) | ((uint32_t) pos << 14) | my_m[u];
                    }
                }
                const int gm = __popcll(__ballot(pass_thr[0] != 0)) + __popcll(__ballot(pass_thr[1] != 0));
                mn = kept_count(mnl, gm, p.min_p, p.max_p);
                /* Wave 0 has flagged EVERY distinct next merge cell.  That is what :1090-1100 keeps: a merge cell's
                 * posterior is at least that of any cell leading to it (max mode, exact integers), so whenever more
                 * than min_p cells were kept they all pass the threshold, and so do their merge cells.  Checked, not
                 * assumed: a violation discards the level (the host falls back to the per-chunk path). */
                if (mn != mnl) errbits |= MRP_ENGINE_ERR_MERGE;
                /* stable descending sort by posterior (:1090) */
                wave_bitonic_sort128(mkey[0], mkey[1], lane);
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int r = lane + u * WAVE;
                    if (r < mn) sc.keptm[lcol * S + r] = (uint16_t) (mkey[u] & 0x3FFFu);
                }
            }
            if (lane == 0) sc.n_keptm[lcol] = mn;
        };
        /* ---- stRPHmm_pruneForwards hmm.c:1049-1109 ---- */
        for (int k = 0; k < K; k++) {
            uint32_t *hk = hist + (k & 1) * nb_r;
            const int bpl = 16; /* bins per lane of the cutoff search */
            CLK(0);
            PRUNE_SHARE(col, lo, hi, nj)
            /* [A] linked cells (getLinkedCells :1021-1047) in list order.  Few cells are linked (a hundred or so kept
             * merge cells lead into the column), so the loop over the cells only tests and stages them; posterior bin,
             * candidate list and histogram are done for 64 staged cells at a time with all lanes busy. */
            uint32_t *sg = stg + wave * 4 * WAVE;
            int cnt = 0, staged = 0;
#define PRUNE_FLUSH(count_)                                                                       \
            {                                                                                     \
                if (lane < (count_)) {                                                            \
                    const uint32_t c_ = sg[lane], np_ = sg[WAVE + lane];                          \
                    const int bin_ = posterior_bin((int32_t) sg[2 * WAVE + lane], (int32_t) sg[3 * WAVE + lane], total, nb, &errbits); \
                    cand[lo + cnt + lane] = ((uint32_t) bin_ << 16) | c_;                         \
                    cand_np[lo + cnt + lane] = np_;                                               \
                    atomicAdd(&hk[(bin_ & 15) * WAVE + (bin_ >> 4)], 1u);                         \
                }                                                                                 \
                cnt += (count_);                                                                  \
            }
#pragma unroll
            for (int j = 0; j < PRUNE_CPT; j++) {
                if (j < nj) {
                    const int c = lo + j * WAVE + lane;
                    const bool linked = c < hi && (k == 0 || flags[r_np[j] >> 16] != 0);
                    const uint64_t m = __ballot(linked);
                    if (m) { /* wave-uniform */
                        const int nl = __popcll(m);
                        const int pos = staged + (int) lanemask_lt_count(m, lane);
                        if (linked && pos < WAVE) { sg[pos] = (uint32_t) c; sg[WAVE + pos] = r_np[j]; sg[2 * WAVE + pos] = (uint32_t) r_f[j]; sg[3 * WAVE + pos] = (uint32_t) r_b[j]; }
                        if (staged + nl >= WAVE) {
                            PRUNE_FLUSH(WAVE)
                            if (linked && pos >= WAVE) { sg[pos - WAVE] = (uint32_t) c; sg[pos] = r_np[j]; sg[WAVE + pos] = (uint32_t) r_f[j]; sg[2 * WAVE + pos] = (uint32_t) r_b[j]; }
                            staged = staged + nl - WAVE;
                        } else {
                            staged += nl;
                        }
                    }
                }
            }
            if (staged > 0) PRUNE_FLUSH(staged)
#undef PRUNE_FLUSH
            if (lane == 0) { sh[wave] = (uint32_t) cnt; atomicAdd(&sh[40 + (k & 1)], (uint32_t) cnt); }
            CLK(1);
            /* the next column's cells are requested now and consumed after the two barriers below */
            const SweepCol cur = col;
            if (k + 1 < K) {
                col = col_next; /* its descriptor was requested one column ago */
                if (k + 2 < K) col_next = k_load(d.scols + h.col0 + k + 2);
                PRUNE_SHARE(col, lo1, hi1, nj1)
                (void) nj1;
                PRUNE_LOAD(col, lo1, hi1, nj1)
            }
            CLK(2);
            lds_barrier();
            CLK(3);
            if (wave == 1) {
                if (k > 0) lists_stage1(k - 1, mcell_prev);
            } else if (wave == 2) {
                if (k > 1) lists_stage2(k - 2);
            } else if (wave > 2) {
                uint32_t *hn = hist + ((k + 1) & 1) * nb_r;
                for (int i = lane + (wave - 3) * WAVE; i < nb_r; i += T - 3 * WAVE) hn[i] = 0;
            }
            if (wave == 0) {
                uint32_t *gsel = sel + (k & 1) * 4 * S, *gnp = gsel + S, *esel = gnp + S, *enp = esel + S;
                /* [B] cutoff bin and quota */
                const int n_link = (int) sh[40 + (k & 1)]; /* summed by the waves at the end of [A] */
                if (lane == 0) sh[40 + ((k + 1) & 1)] = 0u;
                /* lane l owns the bpl consecutive bins [l * bpl, (l + 1) * bpl); the histogram is stored lane-major, so these
                 * reads are conflict-free and the search has a fixed, short cost wherever the cutoff lies (the posteriors of
                 * the linked cells spread over hundreds of bins) */
                int v[16];
                int tot = 0, pass = 0;
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    v[q] = q < bpl ? (int) hk[q * WAVE + lane] : 0;
                    tot += v[q];
                    if (lane * bpl + q <= p.thr_bin) pass += v[q];
                }
                const int incl = wave_incl_scan(tot, lane);
                const int g = p.thr_bin >= nb - 1 ? n_link : __shfl(wave_incl_scan(pass, lane), WAVE - 1, WAVE);
                const int n = kept_count(n_link, g, p.min_p, p.max_p);
                const int ex = incl - tot;
                int myB = -1, myQ = 0;
                const bool owner_lane = n > 0 && ex < n && n <= incl;
                if (owner_lane) {
                    int cum = ex;
#pragma unroll
                    for (int q = 0; q < 16; q++) {
                        if (myB < 0 && cum + v[q] >= n) { myB = lane * bpl + q; myQ = n - cum; }
                        cum += v[q];
                    }
                }
                const uint64_t om = __ballot(owner_lane);
                const int src = om ? __ffsll((unsigned long long) om) - 1 : 0;
                const int srcu = __builtin_amdgcn_readfirstlane(src); /* wave-uniform: v_readlane instead of ds_bpermute */
                const int B = om ? __builtin_amdgcn_readlane(myB, srcu) : -1;
                const int quota = om ? __builtin_amdgcn_readlane(myQ, srcu) : 0;
                const int nG = n - quota;
                CLK(4);
                /* ordered selection over the wave segments (list order) */
                {
                    int gc = 0, ec = 0;
                    for (int w = 0; w < W && (gc < nG || ec < quota); w++) {
                        const int per_w = ((cur.n_cells + W - 1) / W + 63) & ~63;
                        const int base = w * per_w;
                        const int cw = (int) sh[w];
                        for (int i0 = 0; i0 < cw && (gc < nG || ec < quota); i0 += WAVE) {
                            const int i = i0 + lane;
                            const bool valid = i < cw;
                            const uint32_t e = valid ? cand[base + i] : 0u;
                            const int bin = (int) (e >> 16);
                            const bool is_g = valid && bin < B, is_e = valid && bin == B;
                            const uint64_t mg = __ballot(is_g), me = __ballot(is_e);
                            if (is_g) {
                                const int pos = gc + (int) lanemask_lt_count(mg, lane);
                                gsel[pos] = e;
                                gnp[pos] = cand_np[base + i];
                            }
                            if (is_e) {
                                const int pe = ec + (int) lanemask_lt_count(me, lane);
                                if (pe < quota) { esel[pe] = e & 0xFFFFu; enp[pe] = cand_np[base + i]; }
                            }
                            gc += __popcll(mg);
                            ec += __popcll(me);
                        }
                    }
                }
                CLK(5);
                if (lane == 0) { sh[32 + 2 * (k & 1)] = (uint32_t) n; sh[33 + 2 * (k & 1)] = (uint32_t) nG; }
                /* the kept flags: those of the previous merge column go (its selection still sits in the other buffer),
                 * those of the merge column after this one are the next merge cells of the selection */
                {
                    const uint32_t *pg = sel + ((k + 1) & 1) * 4 * S + S, *pe = pg + 2 * S;
                    for (int i = lane; i < n_prev; i += WAVE) flags[(i < nG_prev ? pg[i] : pe[i - nG_prev]) & 0xFFFFu] = 0;
                    if (k + 1 < K)
                        for (int i = lane; i < n; i += WAVE) flags[(i < nG ? gnp[i] : enp[i - nG]) & 0xFFFFu] = 1;
                }
                n_prev = k + 1 < K ? n : 0;
                nG_prev = nG;
                CLK(8);
            }
            mcell_prev = cur.mcell_off;
            lds_barrier();
            CLK(9);
        }
        /* the lists of the last two columns */
        if (wave == 1) lists_stage1(K - 1, mcell_prev);
        if (wave == 2 && K > 1) lists_stage2(K - 2);
        lds_barrier();
        if (wave == 2) lists_stage2(K - 1);
        __syncthreads(); /* also makes the lists above visible in global memory */

        /* ---- stRPHmm_pruneBackwards hmm.c:1111-1158: lists of at most S entries, one wave; the lists of
         * column k - 1 are requested before column k is worked on ---- */
        if (wave == 0) {
            uint32_t pm[2] = {0u, 0u}; /* kept merge cells of the merge column after column k: they own the flags */
            bool pmk[2] = {false, false};
            /* the lists of a column are requested two columns before they are used */
            struct Lists { int nk, nm; uint32_t cc[2], cn[2], mm[2]; };
            auto fetch = [&](int k) {
                Lists L;
                L.nk = 0; L.nm = 0;
#pragma unroll
                for (int u = 0; u < 2; u++) { L.cc[u] = 0u; L.cn[u] = 0u; L.mm[u] = 0u; }
                if (k >= 0) {
                    const int64_t lc = h.col0 + k;
                    L.nk = sc.n_kept[lc];
                    L.nm = sc.n_keptm[lc];
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const int i = lane + u * WAVE;
                        /* (the counts are not known yet when the lists are requested: every slot below S is read) */
                        L.cc[u] = i < S ? sc.kept[lc * S + i] : 0u;
                        L.cn[u] = i < S ? sc.kept_np[lc * S + i] : 0u;
                        L.mm[u] = i < S ? sc.keptm[lc * S + i] : 0u;
                    }
                }
                return L;
            };
            Lists cur = fetch(K - 1), nx1 = fetch(K - 2);
            for (int k = K - 1; k >= 0; k--) {
                const int64_t lcol = h.col0 + k;
                const Lists nx2 = fetch(k - 2);
                const int nk = cur.nk;
                bool keep[2];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int i = lane + u * WAVE;
                    keep[u] = i < nk && (k + 1 == K || flags[cur.cn[u] & 0xFFFFu] != 0);
                }
                const uint64_t m0 = __ballot(keep[0]), m1 = __ballot(keep[1]);
                const int ns = __popcll(m0) + __popcll(m1);
                if (pmk[0]) flags[pm[0]] = 0;
                if (pmk[1]) flags[pm[1]] = 0;
                if (ns != nk) {
                    if (keep[0]) {
                        const int pos = (int) lanemask_lt_count(m0, lane);
                        sc.kept[lcol * S + pos] = (uint16_t) cur.cc[0];
                        sc.kept_np[lcol * S + pos] = cur.cn[0];
                    }
                    if (keep[1]) {
                        const int pos = __popcll(m0) + (int) lanemask_lt_count(m1, lane);
                        sc.kept[lcol * S + pos] = (uint16_t) cur.cc[1];
                        sc.kept_np[lcol * S + pos] = cur.cn[1];
                    }
                    if (lane == 0) sc.n_kept[lcol] = ns;
                }
                if (k == 0) break;
                /* merge column k - 1 keeps the merge cells some surviving cell comes from (:1141-1155) */
                if (keep[0]) flags[cur.cn[0] >> 16] = 1;
                if (keep[1]) flags[cur.cn[1] >> 16] = 1;
                const int nmp = nx1.nm;
                bool mk[2];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int i = lane + u * WAVE;
                    mk[u] = i < nmp && flags[nx1.mm[u]] != 0;
                }
                const uint64_t q0 = __ballot(mk[0]), q1 = __ballot(mk[1]);
                const int nms = __popcll(q0) + __popcll(q1);
                if (nms != nmp) {
                    if (mk[0]) sc.keptm[(lcol - 1) * S + (int) lanemask_lt_count(q0, lane)] = (uint16_t) nx1.mm[0];
                    if (mk[1]) sc.keptm[(lcol - 1) * S + __popcll(q0) + (int) lanemask_lt_count(q1, lane)] = (uint16_t) nx1.mm[1];
                    if (lane == 0) sc.n_keptm[lcol - 1] = nms;
                }
                /* leave flagged exactly the surviving merge cells of column k - 1 */
                if (keep[0]) flags[cur.cn[0] >> 16] = 0;
                if (keep[1]) flags[cur.cn[1] >> 16] = 0;
                if (mk[0]) flags[nx1.mm[0]] = 1;
                if (mk[1]) flags[nx1.mm[1]] = 1;
                pm[0] = nx1.mm[0]; pm[1] = nx1.mm[1];
                pmk[0] = mk[0]; pmk[1] = mk[1];
                cur = nx1;
                nx1 = nx2;
            }
        }
        __syncthreads();
    }
#undef PRUNE_SHARE
#undef PRUNE_LOAD
#ifdef PRUNE_EXP_CLOCK
    CLK(10);
    if (wave == 0 && lane == 0 && T == 1024)
        for (int i = 0; i < 12; i++) atomicAdd((unsigned long long *) (sc.err + 4) + i, (unsigned long long) clk[i]);
#endif
    if (errbits) atomicOr(sc.err, errbits);
}

hipError_t mrp_launch_prune(const MrpBatchDev &d, const PruneHmm *hmms_dev, int64_t n_hmms, PruneParams p, PruneScratch s,
                            hipStream_t stream) {
    if (n_hmms <= 0) return hipSuccess;
    if (p.S > MRP_PRUNE_MAX_S || p.max_cells > MRP_PRUNE_MAX_CELLS || p.max_merge > MRP_PRUNE_MAX_CELLS || p.n_bins > 1024) return hipErrorInvalidValue;
    const size_t cap = (size_t) ((std::max(p.max_cells, p.max_merge) + 3) & ~3);
    auto lds_for = [&](int threads) { return (size_t) (13 * p.S + 64 + (threads / 64) * 4 * 64 + 2 * 1024 + 2 * cap + 512) * 4 + (size_t) ((p.max_merge + 3) & ~3) + 16; };
    /* once per process (thread-safe static initialisation: the concurrent halves of a call launch from two host threads) */
    static const hipError_t configured = [] {
        hipError_t e = hipFuncSetAttribute((const void *) mrp_prune_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *) mrp_prune_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return e;
    }();
    if (configured != hipSuccess) return configured;
    if (lds_for(1024) > (size_t) MRP_LDS_BUDGET) return hipErrorInvalidValue;
    const dim3 grid((unsigned) (n_hmms < 65536 ? n_hmms : 65536));
    const PruneIn in{d.scols, d.cell_np, d.cell_f32, d.cell_b32, d.merge_f32, d.merge_b32, d.hmm_fb};
    /* 1 024 threads for the big columns even when fewer would hold them: the per-wave share of phase [A] is what the
     * column's critical path waits for (640 threads measured 8 % slower) */
    if (p.max_cells <= 256 * PRUNE_CPT)
        hipLaunchKernelGGL(mrp_prune_kernel<256>, grid, dim3(256), lds_for(256), stream, in, hmms_dev, n_hmms, p, s);
    else
        hipLaunchKernelGGL(mrp_prune_kernel<1024>, grid, dim3(1024), lds_for(1024), stream, in, hmms_dev, n_hmms, p, s);
    return hipGetLastError();
}

/* ------------------------------------------------------------------------------------------ */
/* compaction: the pruned hmm in the resident layout                                           */
/* ------------------------------------------------------------------------------------------ */
__global__ void __launch_bounds__(256) mrp_compact_kernel(MrpBatchDev d, const PruneHmm *__restrict__ hmms,
                                                          const int32_t *__restrict__ col_hmm, int64_t n_cols, PruneParams p,
                                                          PruneScratch sc) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x / WAVE));
    const int S = p.S;
    const int64_t stride = (int64_t) gridDim.x * (blockDim.x / WAVE);
    for (int64_t lcol = (int64_t) blockIdx.x * (blockDim.x / WAVE) + wave; lcol < n_cols; lcol += stride) {
        const PruneHmm h = k_load(hmms + col_hmm[lcol]);
        const int k = (int) (lcol - h.col0);
        const int K = h.n_cols;
        const SweepCol col = k_load(d.scols + lcol);
        const int nk = sc.n_kept[lcol];
        const int nm = k + 1 < K ? sc.n_keptm[lcol] : 0;
        const int nmp = k > 0 ? sc.n_keptm[lcol - 1] : 0;
        for (int i = lane; i < nk; i += WAVE) {
            const uint32_t c = sc.kept[lcol * S + i];
            const uint32_t np = sc.kept_np[lcol * S + i];
            const uint32_t nx = np & 0xFFFFu, pv = np >> 16;
            /* filterMergeCells keeps the merge cells in their original relative order */
            uint32_t new_next = 0, new_prev = 0;
            for (int j = 0; j < nm; j++) new_next += sc.keptm[lcol * S + j] < nx ? 1u : 0u;
            for (int j = 0; j < nmp; j++) new_prev += sc.keptm[(lcol - 1) * S + j] < pv ? 1u : 0u;
            const uint64_t part = d.partition[col.cell_off + c];
            h.out_part[(int64_t) k * S + i] = part;
            h.out_np[(int64_t) k * S + i] = new_next | (new_prev << 16);
        }
        if (lane == 0) {
            h.out_n_cells[k] = nk;
            h.out_n_merge[k] = nm;
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* trace back                                                                                  */
/* ------------------------------------------------------------------------------------------ */
__global__ void __launch_bounds__(64) mrp_traceback_kernel(MrpBatchDev d, const PruneHmm *__restrict__ hmms, int64_t n_hmms,
                                                           int32_t *__restrict__ err) {
    const int lane = threadIdx.x;
    for (int64_t hi = blockIdx.x; hi < n_hmms; hi += gridDim.x) {
        const PruneHmm h = k_load(hmms + hi);
        const int K = h.n_cols;
        uint32_t want = 0; /* merge cell the chosen cell of column k + 1 comes from */
        /* The walk is a chain of dependent steps (the chosen cell names the merge cell the next column is filtered by), but
         * what a step READS does not depend on the chain: the descriptor of a column is requested two steps ahead and its
         * cells (two per lane: a pruned column has at most 128) one step ahead, so a step is a filter, an argmax over the
         * wave and a lane broadcast. */
        struct Cells { uint32_t np[2]; int32_t f[2]; };
        auto fetch = [&](const SweepCol &c) {
            Cells r;
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int i = lane + u * WAVE;
                const bool in = i < c.n_cells && c.n_cells <= 2 * WAVE;
                r.np[u] = in ? d.cell_np[c.cell_off + i] : 0u;
                r.f[u] = in ? d.cell_f32[c.cell_off + i] : MRP_NEG_I32;
            }
            return r;
        };
        SweepCol col = k_load(d.scols + h.col0 + K - 1);
        SweepCol col_m1 = K > 1 ? k_load(d.scols + h.col0 + K - 2) : col;
        Cells cur = fetch(col);
        for (int k = K - 1; k >= 0; k--) {
            const SweepCol col_m2 = k > 1 ? k_load(d.scols + h.col0 + k - 2) : col_m1;
            Cells nxt = cur;
            if (k > 0) nxt = fetch(col_m1);
            /* hmm.c:173-186 (last column: best forward probability) / :196-214 (cells feeding the chosen merge cell) */
            int32_t best = 0, best_i = -1;
            const bool small = col.n_cells <= 2 * WAVE;
            if (small) {
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int c = lane + u * WAVE;
                    if (c >= col.n_cells) continue;
                    if (k + 1 < K && (cur.np[u] & 0xFFFFu) != want) continue;
                    const int32_t f = cur.f[u];
                    if (f == MRP_NEG_I32) continue; /* -inf never beats the initial -inf of the reference loop ... */
                    if (best_i < 0 || f > best) { best = f; best_i = c; }
                }
            } else {
                for (int c = lane; c < col.n_cells; c += WAVE) {
                    if (k + 1 < K && (d.cell_np[col.cell_off + c] & 0xFFFFu) != want) continue;
                    const int32_t f = d.cell_f32[col.cell_off + c];
                    if (f == MRP_NEG_I32) continue;
                    if (best_i < 0 || f > best) { best = f; best_i = c; }
                }
            }
            {   /* argmax with the first index winning ties, as one unsigned maximum: strides 1..8 by DPP */
                uint64_t key = best_i < 0 ? 0ull : ((uint64_t) ((uint32_t) best ^ 0x80000000u) << 32) | (uint32_t) (0x7FFFFFFF - best_i);
#define TB_STEP(J)                                                                                                      \
                {                                                                                                       \
                    const uint64_t o = ((uint64_t) lane_xor<J>((uint32_t) (key >> 32), lane) << 32) | lane_xor<J>((uint32_t) key, lane); \
                    key = o > key ? o : key;                                                                            \
                }
                TB_STEP(1) TB_STEP(2) TB_STEP(4) TB_STEP(8) TB_STEP(16) TB_STEP(32)
#undef TB_STEP
                best_i = key ? 0x7FFFFFFF - (int32_t) (uint32_t) key : -1;
            }
            if (best_i < 0) {
                /* ... except in the last column, where the reference starts from the first cell */
                if (k + 1 == K) best_i = 0;
                else {
                    if (lane == 0) atomicOr(err, MRP_ENGINE_ERR_RANGE);
                    best_i = 0;
                }
            }
            if (lane == 0) {
                h.out_n_cells[k] = best_i;
                h.out_part[k] = d.partition[col.cell_off + best_i];
            }
            if (small) {
                const uint32_t from = best_i < WAVE ? cur.np[0] : cur.np[1];
                want = (uint32_t) __shfl((int) from, best_i & (WAVE - 1), WAVE) >> 16;
            } else {
                want = d.cell_np[col.cell_off + best_i] >> 16;
            }
            col = col_m1;
            col_m1 = col_m2;
            cur = nxt;
        }
    }
}

hipError_t mrp_launch_traceback(const MrpBatchDev &d, const PruneHmm *hmms_dev, int64_t n_hmms, int32_t *err, hipStream_t stream) {
    if (n_hmms <= 0) return hipSuccess;
    hipLaunchKernelGGL(mrp_traceback_kernel, dim3((unsigned) (n_hmms < 65536 ? n_hmms : 65536)), dim3(64), 0, stream, d, hmms_dev,
                       n_hmms, err);
    return hipGetLastError();
}

hipError_t mrp_launch_compact(const MrpBatchDev &d, const PruneHmm *hmms_dev, const int32_t *col_hmm_dev, int64_t n_cols,
                              PruneParams p, PruneScratch s, hipStream_t stream) {
    if (n_cols <= 0) return hipSuccess;
    const int64_t wgs = (n_cols + 3) / 4;
    hipLaunchKernelGGL(mrp_compact_kernel, dim3((unsigned) (wgs < 65536 ? wgs : 65536)), dim3(256), 0, stream, d, hmms_dev,
                       col_hmm_dev, n_cols, p, s);
    return hipGetLastError();
}
