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
#include <cstdlib>
#include <type_traits>

#include "mrp_engine.h"
#include "mrp_internal.h"
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
/* LDS traffic of ONE wave is in order once its counter has drained: what lanes wrote is visible to the other lanes */
static __device__ __forceinline__ void wave_lds_fence() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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
                                  bool in_paired, uint32_t tid, uint32_t nt) {
    if (!part) return 0;
    int bad = 0;
    const bool cells_paired = inv && depth > 0;
    if (cells_paired && (C & 1u)) return MRP_ENGINE_ERR_STRUCTURE;
    const uint64_t acc = accept_mask(depth);
    for (uint32_t e = tid; e < C; e += nt) {
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

/* The ORDER RULE of a cross product column (the one place it is written down on the device; pair_index above is its twin
 * for merge cells): cell e of the column is the pair (c1, c2) of parent cells.  Plain mode: row-major.  With inverted
 * partitions: the reference appends the complement right after every new partition, so the cells come as
 * (2r, h), (2r + 1, partner of h), (2r, h + 1), ... */
static __device__ __forceinline__ void cross_cell(uint32_t e, uint32_t C2, bool inv, bool a_cells_paired, bool b_cells_paired,
                                                  uint32_t &c1, uint32_t &c2) {
    if (!inv) { c1 = e / C2; c2 = e - c1 * C2; }
    else if (!a_cells_paired) { c1 = 0; c2 = e; }
    else {
        const uint32_t r = e / (2u * C2), t = e - r * 2u * C2, h = t >> 1;
        if (t & 1u) { c1 = 2u * r + 1u; c2 = b_cells_paired ? (h ^ 1u) : h; }
        else { c1 = 2u * r; c2 = h; }
    }
}
/* ... and the merge cells it feeds (low 16 bits) and is fed by (high 16 bits), from the parents' transitions n1, n2 */
static __device__ __forceinline__ uint32_t cross_np(const CrossCol &c, bool inv, uint32_t c1, uint32_t c2, uint32_t n1, uint32_t n2) {
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
    return nxt | (prv << 16);
}
/* mergePartitionsOrMasks (partitions.c:21-28): side B's reads follow side A's */
static __device__ __forceinline__ uint64_t cross_partition(const CrossCol &c, uint32_t c1, uint32_t c2) {
    const uint64_t p1 = c.a_part ? c.a_part[c1] : 0ull, p2 = c.b_part ? c.b_part[c2] : 0ull;
    return c.d1 < 64 ? (p1 | (p2 << c.d1)) : p1;
}

__global__ void __launch_bounds__(256) mrp_cross_kernel(const CrossCol *__restrict__ cols, int64_t n_cols,
                                                        uint64_t *__restrict__ partition, uint32_t *__restrict__ cell_np,
                                                        int32_t *__restrict__ err, const int32_t *__restrict__ col_hmm,
                                                        int32_t *__restrict__ err_hmm) {
    for (int64_t col = blockIdx.x; col < n_cols; col += gridDim.x) {
        const CrossCol c = k_load(cols + col);
        const bool inv = (c.flags & MRP_XF_INVERTED) != 0;
        const uint32_t C1 = c.C1, C2 = c.C2, C = C1 * C2;
        const bool a_cells_paired = inv && c.a_part && c.d1 > 0, b_cells_paired = inv && c.b_part && c.d2 > 0;
        int bad = verify_side(c.a_part, c.a_np, C1, c.d1, c.Ma, c.Pa, c.out_a, c.in_a, inv,
                              (c.flags & MRP_XF_OUT_A_PAIRED) != 0, (c.flags & MRP_XF_IN_A_PAIRED) != 0, threadIdx.x, blockDim.x);
        bad |= verify_side(c.b_part, c.b_np, C2, c.d2, c.Mb, c.Pb, c.out_b, c.in_b, inv,
                           (c.flags & MRP_XF_OUT_B_PAIRED) != 0, (c.flags & MRP_XF_IN_B_PAIRED) != 0, threadIdx.x, blockDim.x);
        if ((!c.a_part && C1 != 1u) || (!c.b_part && C2 != 1u)) bad |= MRP_ENGINE_ERR_RANGE;
        if (bad) { atomicOr(err, bad); atomicOr(err_hmm + col_hmm[col], bad); }
        if (__syncthreads_or(bad)) { /* the level is discarded by the host; keep the arrays defined meanwhile */
            for (uint32_t e = threadIdx.x; e < C; e += blockDim.x) {
                partition[c.x_cell_off + e] = 0ull;
                cell_np[c.x_cell_off + e] = 0u;
            }
            continue;
        }
        for (uint32_t e = threadIdx.x; e < C; e += blockDim.x) {
            uint32_t c1, c2;
            cross_cell(e, C2, inv, a_cells_paired, b_cells_paired, c1, c2);
            const uint32_t n1 = c.a_part ? c.a_np[c1] : 0u, n2 = c.b_part ? c.b_np[c2] : 0u;
            partition[c.x_cell_off + e] = cross_partition(c, c1, c2);
            cell_np[c.x_cell_off + e] = cross_np(c, inv, c1, c2, n1, n2);
        }
    }
}

hipError_t mrp_launch_cross(const CrossCol *cols_dev, int64_t n_cols, uint64_t *partition, uint32_t *cell_np, int32_t *err,
                            const int32_t *col_hmm_dev, int32_t *err_hmm, hipStream_t stream) {
    if (n_cols <= 0) return hipSuccess;
    const int64_t grid = n_cols < 65536 ? n_cols : 65536;
    hipLaunchKernelGGL(mrp_cross_kernel, dim3((unsigned) grid), dim3(256), 0, stream, cols_dev, n_cols, partition, cell_np, err, col_hmm_dev,
                       err_hmm);
    return hipGetLastError();
}

/* ------------------------------------------------------------------------------------------ */
/* cross product + emission in one pass (merge levels, no ancestor substitution model)          */
/* ------------------------------------------------------------------------------------------ */
/*
 * emissionLogProbability (emissions.c:221-240) of a cross product cell WITHOUT its partition ever being written: the cell
 * (c1, c2) holds the reads of parent cell c1 followed by those of parent cell c2 (partitions.c:21-28), and
 * getLogProbOfAllele (emissions.c:125-138) is a sum over the reads of the partition, so per allele slot
 *      hap1(c1, c2) = tA[slot][c1] + tB[slot][c2],   hap2 = T[slot] - hap1          (emissions.c:144-154)
 * with tA / tB the per-parent-cell sums over the side's own reads: (C1 + C2) * slots dot products per column instead of
 * C1 * C2 * slots.  One wave per column.  The tables sit in LDS as packed 16-bit halves, one 16-byte aligned row per parent cell,
 *      A[c1][slot] = tA | (T - tA) << 16        B[c2][slot] = tB - (tB << 16)
 * so that ONE 32-bit add yields hap1 in the low and hap2 in the high half (no carry crosses: both are sums of at most 64
 * bytes), and one v_pk_min_u16 keeps both running minima over the alleles of a site (emissions.c:174-185, :205-207).
 * With inverted partitions cells 2q, 2q+1 are complements and share their cost (hap1 <-> hap2): a lane computes it once.
 * HBM traffic: 8 B written per cell (cost + transitions), nothing read per cell.
 */
#ifndef XE_WAVES
#define XE_WAVES 1
#endif
/* Table dwords per wave, chosen per launch (dynamic LDS): XE_CAP_NARROW for the levels whose columns all take the table-free path
 * below (200 parent cells x 4 allele slots would still fit: a column that needs the tables after all fills them several times) --
 * 6.4 KB per one-wave workgroup, the registers' 20 waves per CU fit; XE_CAP_WIDE where two pruned columns of 100 cells meet: 8 slots
 * = 4 biallelic sites per fill instead of 2 (a column of more sites walks its cells once per fill, adding to the costs already
 * stored), 9.7 KB, 16 waves per CU.  Measured per 96-chunk batch, widest level / the two above it / the three below it:
 * 832 dwords 1.39 / 0.65 + 0.48 / 0.37-0.39 ms; 1 248: 1.24 / 0.64 + 0.49 / 0.39-0.41; 1 664: 1.06 / 0.60 + 0.48 / 0.35-0.44;
 * 2 496: 1.16 / 0.78 + 0.47 / 0.48-0.59; 3 328: 1.26 / 0.84 + 0.58 / 0.54-0.69. */
#ifndef XE_CAP_NARROW
#define XE_CAP_NARROW 832
#endif
#ifndef XE_CAP_WIDE
#define XE_CAP_WIDE 1664
#endif
#ifndef XE_ROWS
#define XE_ROWS 16 /* allele slots staged per table fill */
#endif

typedef unsigned short xe_u16x2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
    const xe_u16x2 r = __builtin_elementwise_min(__builtin_bit_cast(xe_u16x2, a), __builtin_bit_cast(xe_u16x2, b));
    return __builtin_bit_cast(uint32_t, r);
}
/* sum of the profile bytes of the reads in P over one allele slot: bytes are packed four reads to a word */
template <typename RowPtr>
static __device__ __forceinline__ uint32_t slot_dot(RowPtr row, uint64_t P, int w4_lo, int w4_hi) {
    uint32_t sum = 0;
    for (int w4 = w4_lo; w4 < w4_hi; w4++) {
        const uint4 bts = *reinterpret_cast<const uint4 *>(row + 4 * w4);
        const uint32_t bits = (uint32_t) (P >> (16 * w4)) & 0xFFFFu;
        sum = __builtin_amdgcn_udot4(bts.x, ((bits & 0xFu) * 0x00204081u) & 0x01010101u, sum, false);
        sum = __builtin_amdgcn_udot4(bts.y, (((bits >> 4) & 0xFu) * 0x00204081u) & 0x01010101u, sum, false);
        sum = __builtin_amdgcn_udot4(bts.z, (((bits >> 8) & 0xFu) * 0x00204081u) & 0x01010101u, sum, false);
        sum = __builtin_amdgcn_udot4(bts.w, ((bits >> 12) * 0x00204081u) & 0x01010101u, sum, false);
    }
    return sum;
}

/* cost of one cell over the sites of the tables: per site min over alleles of hap1 plus min over alleles of hap2.
 * pa, pb: the cell's two table rows (slot-minor, 16-byte aligned): four slots per LDS read. */
static __device__ __forceinline__ uint32_t xe_cost(const uint32_t *pa, const uint32_t *pb, uint32_t nsl, uint64_t ends) {
    uint32_t cost = 0, m = 0xFFFFFFFFu;
    for (uint32_t s4 = 0; s4 < nsl; s4 += 4u) {
        const uint4 a = *reinterpret_cast<const uint4 *>(pa + s4), b = *reinterpret_cast<const uint4 *>(pb + s4);
        const uint32_t x[4] = {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w};
        const uint32_t e4 = (uint32_t) (ends >> s4) & 0xFu, left = nsl - s4; /* wave-uniform */
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if ((uint32_t) j < left) {
                m = pk_min_u16(m, x[j]);
                if (e4 & (1u << j)) { cost += (m & 0xFFFFu) + (m >> 16); m = 0xFFFFFFFFu; }
            }
        }
    }
    return cost;
}
/* Table rows of one side for columns with many parent cells: a lane owns a parent cell (its partition's read bits are
 * expanded to 0/1 bytes once); the packed bytes of a slot were staged in LDS and every lane reads the same words.
 * NW4 = 16-read groups the side's reads span; the partition is shifted up to the merged column's read positions, then down
 * by the first group (rowbuf points at that group's words). */
template <int NW4, bool SIDE_B>
static __device__ __forceinline__ void xe_fill_side(uint32_t *rows, uint32_t ST, uint64_t P0, uint64_t P1, uint32_t C, uint32_t up,
                                                    uint32_t down, uint32_t nsl, const uint32_t *rowbuf, const uint32_t *totbuf, int lane) {
    /* the partitions of parent cells lane and lane + 64 are in the lane's registers (P0, P1: requested with everything else
     * the column needs, right after its descriptor) */
    for (uint32_t u = 0; u * WAVE < C; u++) {
        const uint32_t cc = (uint32_t) lane + u * WAVE;
        const bool act = cc < C;
        uint64_t P = act ? (u ? P1 : P0) : 0ull;
        P = up < 64u ? (P << up) >> down : 0ull;
        uint32_t sel[4 * NW4];
#pragma unroll
        for (int w = 0; w < 4 * NW4; w++) sel[w] = ((uint32_t) ((P >> (4 * w)) & 0xFull) * 0x00204081u) & 0x01010101u;
        for (uint32_t slot = 0; slot < nsl; slot++) {
            uint32_t t = 0;
#pragma unroll
            for (int k = 0; k < NW4; k++) {
                const uint4 r = *reinterpret_cast<const uint4 *>(rowbuf + slot * 16 + 4 * k);
                t = __builtin_amdgcn_udot4(r.x, sel[4 * k], t, false);
                t = __builtin_amdgcn_udot4(r.y, sel[4 * k + 1], t, false);
                t = __builtin_amdgcn_udot4(r.z, sel[4 * k + 2], t, false);
                t = __builtin_amdgcn_udot4(r.w, sel[4 * k + 3], t, false);
            }
            if (act) rows[cc * ST + slot] = SIDE_B ? t - (t << 16) : (t | ((totbuf[slot] - t) << 16));
        }
    }
}

/* pair_index (the merge cell a cell feeds / is fed by) split into a term per parent cell of side A and two per parent cell of
 * side B, computed once per parent cell instead of once per cell:
 *      index(c1, c2) = base(c1) + (sel(c1) ? B1(c2) : B0(c2))
 * tra[c1] = {out: base | sel << 31, in: base | sel << 31},  trb[c2] = {out: B0 | B1 << 16, in: B0 | B1 << 16}. */
static __device__ __forceinline__ uint32_t xe_term_a(uint32_t kind, bool none, uint32_t n_idx, uint32_t c1, uint32_t Mb, bool inv, bool a_paired) {
    const uint32_t i = kind == MRP_CONN_REAL ? n_idx : (kind == MRP_CONN_IDENT ? c1 : 0u);
    if (none) return 0u;
    if (!inv) return i * Mb;
    if (!a_paired) return 0u;
    return (i & 1u) ? (((i - 1u) * Mb + 1u) | 0x80000000u) : i * Mb;
}
static __device__ __forceinline__ uint32_t xe_term_b(uint32_t kind, bool none, uint32_t n_idx, uint32_t c2, bool inv, bool a_paired, bool b_paired) {
    const uint32_t j = kind == MRP_CONN_REAL ? n_idx : (kind == MRP_CONN_IDENT ? c2 : 0u);
    if (none) return 0u;
    if (!inv || !a_paired) return j | (j << 16);
    return (2u * j) | ((2u * (b_paired ? (j ^ 1u) : j)) << 16);
}
/* np0, np1: the transitions of parent cells lane and lane + 64 of each side (registers) */
static __device__ __forceinline__ void xe_stage_transitions(const CrossCol &c, bool inv, uint2 *tra, uint2 *trb, int lane, const uint32_t *npa,
                                                            const uint32_t *npb) {
    const bool oap = (c.flags & MRP_XF_OUT_A_PAIRED) != 0, obp = (c.flags & MRP_XF_OUT_B_PAIRED) != 0;
    const bool iap = (c.flags & MRP_XF_IN_A_PAIRED) != 0, ibp = (c.flags & MRP_XF_IN_B_PAIRED) != 0;
    const bool no_out = c.out_a == MRP_CONN_NONE, no_in = c.in_a == MRP_CONN_NONE;
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const uint32_t i = (uint32_t) lane + (uint32_t) u * WAVE;
        if (i < min((uint32_t) c.C1, 128u))
            tra[i] = make_uint2(xe_term_a(c.out_a, no_out, npa[u] & 0xFFFFu, i, c.Mb, inv, oap), xe_term_a(c.in_a, no_in, npa[u] >> 16, i, c.Pb, inv, iap));
        if (i < min((uint32_t) c.C2, 128u))
            trb[i] = make_uint2(xe_term_b(c.out_b, no_out, npb[u] & 0xFFFFu, i, inv, oap, obp), xe_term_b(c.in_b, no_in, npb[u] >> 16, i, inv, iap, ibp));
    }
}
/* verify_side over a side of at most 128 parent cells held in registers (entry e = lane + 64 u; its complement e ^ 1 sits in
 * the neighbouring lane) */
template <int NU = 2>
static __device__ __forceinline__ int xe_verify_side(bool have, const uint64_t *P, const uint32_t *np, uint32_t C, uint32_t depth, uint32_t M_out,
                                                     uint32_t M_in, uint32_t out_kind, uint32_t in_kind, bool inv, bool out_paired, bool in_paired,
                                                     int lane) {
    if (!have) return 0;
    int bad = 0;
    const bool cells_paired = inv && depth > 0;
    if (cells_paired && (C & 1u)) return MRP_ENGINE_ERR_STRUCTURE;
    const uint64_t acc = accept_mask(depth);
#pragma unroll
    for (int u = 0; u < NU; u++) {
        const uint32_t e = (uint32_t) lane + (uint32_t) u * WAVE;
        const uint32_t v = np[u], nx = v & 0xFFFFu, pv = v >> 16;
        const uint32_t v_other = (uint32_t) __shfl_xor((int) v, 1, WAVE);
        const uint64_t p_other = ((uint64_t) (uint32_t) __shfl_xor((int) (uint32_t) (P[u] >> 32), 1, WAVE) << 32) |
                                 (uint32_t) __shfl_xor((int) (uint32_t) P[u], 1, WAVE);
        if (e >= C) continue;
        uint32_t vo = v;
        if (cells_paired) {
            if (p_other != (~P[u] & acc)) bad |= MRP_ENGINE_ERR_STRUCTURE;
            vo = v_other;
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
static __device__ __forceinline__ uint32_t xe_np(uint2 a, uint2 b) { /* next | prev << 16 */
    const uint32_t nxt = (a.x & 0x7FFFFFFFu) + ((int32_t) a.x < 0 ? b.x >> 16 : b.x & 0xFFFFu);
    const uint32_t prv = (a.y & 0x7FFFFFFFu) + ((int32_t) a.y < 0 ? b.y >> 16 : b.y & 0xFFFFu);
    return nxt | (prv << 16);
}

/* ---- narrow columns: at most 64 array entries and at most 64 parent cells per side (the first merge levels: a few cells per
 * column, long runs of sites) ----
 * No tables: entry `lane` of the column is a cell (c1, c2) whose merged partition P the lane holds; its cost over one allele slot is
 * getLogProbOfAllele (emissions.c:125-138) evaluated directly, Sum_w v_dot4(bytes of reads 4w..4w+3, bits of P expanded to 0/1
 * bytes).  The packed bytes of a slot are the same 64 bytes for every lane.  A chunk of up to 16 slots (whole sites) is requested
 * at once: FOUR coalesced loads -- lane l of load g holds dword l & 15 of slot 4 g + (l >> 4) -- and one for the byte sums, all in
 * flight together (one round trip per 16 slots; scalar loads, tried first, return out of order and the compiler waits for each
 * slot's row in turn); a slot's words then reach every lane through v_readlane (wave-uniform operands of the v_dot4).  Per site the
 * minima over its alleles of hap1 and of hap2 = total - hap1 (emissions.c:144-154, :174-185, :205-207).  With the table kernel a
 * column of the first merge levels cost ~1 900 instructions whatever its cells (two table fills through LDS, transition terms, a
 * grid walk), two thirds of them on the scalar unit; this path has neither LDS traffic nor a second pass. */
/* The same cost with the lanes along the SLOTS (biallelic sites, a column of few cells and many sites: the first merge level's
 * columns hold two to four cells and up to sixty sites): lane l keeps the packed bytes of slot l of a chunk of 64 -- every lane its own
 * 16 x NW4 bytes, the wave a contiguous 4 KB --; for each cell of the column (its partition broadcast through a scalar pair) one
 * v_dot4 chain per lane gives hap1 of all 64 slots at once, the two alleles of a site sit in neighbouring lanes (one DPP swap for the
 * minima) and the sites are summed across the wave.  Per cell and 32 sites some 35 instructions, where a lane per cell takes 17 per
 * slot.  Returns, in lane e, the cost of cell e over the chunk. */
static __device__ __forceinline__ uint32_t xe_wave_sum_u32(uint32_t v) {
    int x = (int) v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);  /* row_shr:1 */
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false); /* row_bcast:15 into rows 1 and 3 */
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false); /* row_bcast:31 into rows 2 and 3 */
    return (uint32_t) __builtin_amdgcn_readlane(x, WAVE - 1);
}
template <int NW4>
static __device__ __forceinline__ uint32_t xe_slots_chunk(const uint32_t *__restrict__ slot_bytes, const uint32_t *__restrict__ slot_total, int64_t slot0,
                                                          uint32_t nsl, uint32_t U, uint64_t P, int lane) {
    const bool have = (uint32_t) lane < nsl;
    const uint4 *rp = reinterpret_cast<const uint4 *>(slot_bytes + (slot0 + (have ? (uint32_t) lane : 0u)) * 16);
    uint4 row[NW4];
#pragma unroll
    for (int k = 0; k < NW4; k++) row[k] = rp[k];
    const uint32_t tot = slot_total[slot0 + (have ? (uint32_t) lane : 0u)];
    uint32_t mine = 0;
    for (uint32_t e = 0; e < U; e++) { /* wave-uniform */
        const uint64_t Pe = ((uint64_t) (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) (P >> 32), (int) e) << 32) |
                            (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) P, (int) e);
        uint32_t t = 0;
#pragma unroll
        for (int k = 0; k < NW4; k++) {
            const uint32_t bits = (uint32_t) (Pe >> (16 * k)) & 0xFFFFu; /* scalar: the expansion runs on the scalar unit */
            t = __builtin_amdgcn_udot4(row[k].x, ((bits & 0xFu) * 0x00204081u) & 0x01010101u, t, false);
            t = __builtin_amdgcn_udot4(row[k].y, (((bits >> 4) & 0xFu) * 0x00204081u) & 0x01010101u, t, false);
            t = __builtin_amdgcn_udot4(row[k].z, (((bits >> 8) & 0xFu) * 0x00204081u) & 0x01010101u, t, false);
            t = __builtin_amdgcn_udot4(row[k].w, ((bits >> 12) * 0x00204081u) & 0x01010101u, t, false);
        }
        const uint32_t h2 = tot - t;
        const uint32_t o1 = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) t, 0xB1, 0xf, 0xf, false);  /* quad_perm [1,0,3,2]: the site's other allele */
        const uint32_t o2 = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) h2, 0xB1, 0xf, 0xf, false);
        const uint32_t v = (have && (lane & 1) == 0) ? min(t, o1) + min(h2, o2) : 0u;
        const uint32_t sum = xe_wave_sum_u32(v);
        mine += (uint32_t) lane == e ? sum : 0u;
    }
    return mine;
}

template <int NW4>
static __device__ __forceinline__ uint32_t xe_narrow_chunk(const uint32_t *__restrict__ slot_bytes, const uint32_t *__restrict__ slot_total, int64_t slot0,
                                                           uint32_t nsl, uint32_t ends, const uint32_t (&sel)[16], int lane, uint32_t &m1, uint32_t &m2) {
    const uint32_t grp = (uint32_t) lane >> 4, dw = (uint32_t) lane & 15u;
    uint32_t r[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int g = 0; g < 4; g++)
        if (4u * (uint32_t) g < nsl) r[g] = slot_bytes[(slot0 + min(4u * (uint32_t) g + grp, nsl - 1u)) * 16 + dw];
    const uint32_t tot_v = slot_total[slot0 + min((uint32_t) lane, nsl - 1u)];
    uint32_t cost = 0;
#pragma unroll
    for (int g = 0; g < 4; g++) {
        if (4u * (uint32_t) g < nsl) { /* wave-uniform */
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t sidx = 4u * (uint32_t) g + (uint32_t) j;
                const bool valid = sidx < nsl;
                const bool last = valid && ((ends >> sidx) & 1u) != 0u;
                uint32_t t = 0;
#pragma unroll
                for (int k = 0; k < 4 * NW4; k++) t = __builtin_amdgcn_udot4((uint32_t) __builtin_amdgcn_readlane((int) r[g], j * 16 + k), sel[k], t, false);
                const uint32_t tt = (uint32_t) __builtin_amdgcn_readlane((int) tot_v, g * 4 + j);
                const uint32_t n1 = min(m1, t), n2 = min(m2, tt - t);
                m1 = valid ? n1 : m1;
                m2 = valid ? n2 : m2;
                cost += last ? m1 + m2 : 0u;
                m1 = last ? 0xFFFFFFFFu : m1;
                m2 = last ? 0xFFFFFFFFu : m2;
            }
        }
    }
    return cost;
}

#ifdef XE_CLOCK /* development: where a wave of mrp_cross_emit_kernel spends its time (summed over the waves of a launch, clock ticks) */
#define XE_T(i) do { const uint64_t t_ = __builtin_amdgcn_s_memtime(); xe_sec[i] += t_ - xe_t; xe_t = t_; } while (0)
#else
#define XE_T(i) do { } while (0)
#endif
struct __attribute__((packed, aligned(4))) xe_u32x4 { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(4))) xe_u32x2 { uint32_t x, y; };

#ifndef XE_MIN_WAVES
#define XE_MIN_WAVES 1
#endif
#ifndef XE_FAST2_ON
#define XE_FAST2_ON 1 /* (development: 0 keeps the general grid walk for biallelic unit levels) */
#endif
#ifndef XE_SLOTS_ON
#define XE_SLOTS_ON 1 /* (development: 0 keeps a lane per cell for every narrow column) */
#endif
#ifndef XE_NARROW_ON
#define XE_NARROW_ON 1 /* (development: 0 sends every column through the tables) */
#endif
/* MODE 1: the narrow columns only (no tables, no LDS, two thirds of the registers: eight waves per SIMD wait for their descriptors and
 * parent cells side by side); MODE 2: the columns that need the tables; a wave that meets a column of the other kind ends behind the
 * column's descriptor.  MODE 0: both kinds in one launch.  The first merge levels, whose hmms are nearly all small, take MODE 1 (and
 * MODE 2 behind it unless the static bounds say every column is narrow); the levels of wide columns take MODE 0 -- a pass of waves
 * that only skip costs them 70-90 us per 190 000 columns, more than their few narrow columns gain from the leaner kernel. */
template <int MODE>
__global__ void __launch_bounds__(XE_WAVES * WAVE, MODE == 1 ? 8 : XE_MIN_WAVES) mrp_cross_emit_kernel(const CrossCol *__restrict__ ccols, const DevCol *__restrict__ cols,
                                                                         const DevChunk *__restrict__ chunks, int64_t n_cols,
                                                                         const uint32_t *__restrict__ slot_bytes,
                                                                         const uint32_t *__restrict__ slot_total,
                                                                         uint32_t *__restrict__ cell_np, uint32_t *__restrict__ cell_cost,
                                                                         int32_t *__restrict__ err, const int32_t *__restrict__ col_hmm,
                                                                         int32_t *__restrict__ err_hmm, uint32_t cap) {
    /* per wave: tables [cap] | packed bytes of a fill [XE_ROWS * 16] | their byte sums [XE_ROWS] | transition terms of the parent cells
     * [256 uint2], see xe_stage_transitions */
    extern __shared__ __attribute__((aligned(16))) uint32_t xe_lds[];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x / WAVE));
    uint32_t *tab = xe_lds + (size_t) wave * (cap + XE_ROWS * 16 + XE_ROWS + 512), *rowbuf = tab + cap, *totbuf = rowbuf + XE_ROWS * 16;
    uint2 *tra = reinterpret_cast<uint2 *>(totbuf + XE_ROWS), *trb = tra + 128;
#ifdef XE_CLOCK
    uint64_t xe_sec[8] = {0, 0, 0, 0, 0, 0, 0, 0}, xe_t = __builtin_amdgcn_s_memtime();
#endif
    for (int64_t col = (int64_t) blockIdx.x * XE_WAVES + wave; col < n_cols; col += (int64_t) gridDim.x * XE_WAVES) {
        const CrossCol c = k_load(ccols + col);
        const DevCol dc = k_load(cols + col);
#ifdef XE_CLOCK
        if (c.x_cell_off == -0x123456789LL || dc.slot_off == -0x123456789LL) continue; /* (uses the descriptors: the wait for them is section 0) */
        XE_T(0);
        xe_sec[7] += 1;
#endif
        const bool inv = (c.flags & MRP_XF_INVERTED) != 0;
        const uint32_t C1 = c.C1, C2 = c.C2, C = C1 * C2;
        const bool a_cells_paired = inv && c.a_part && c.d1 > 0, b_cells_paired = inv && c.b_part && c.d2 > 0;
        /* MRP_XF_UNITS: one array entry per complement pair (the even cell's cost, its transitions as merge UNIT indices) */
        const bool units = (c.flags & MRP_XF_UNITS) != 0;
        const uint32_t ush = units && (a_cells_paired || b_cells_paired) ? 1u : 0u; /* entry of cell e: e >> ush, written for even e */
        const uint32_t nsh = units && (c.flags & (MRP_XF_OUT_A_PAIRED | MRP_XF_OUT_B_PAIRED)) ? 1u : 0u;
        const uint32_t psh = units && (c.flags & (MRP_XF_IN_A_PAIRED | MRP_XF_IN_B_PAIRED)) ? 1u : 0u;
        auto np_entry = [&](uint32_t v) -> uint32_t { return ((v & 0xFFFFu) >> nsh) | (((v >> 16) >> psh) << 16); };
        /* slots per table fill: rows of the tables are padded to a multiple of four slots, the staging buffer holds XE_ROWS */
        const uint32_t Cs = C1 + C2;
        const uint32_t slot_room = (Cs >= 1u && Cs <= 256u) ? min((uint32_t) XE_ROWS, (cap / Cs) & ~3u) : 0u;
        const uint32_t A_uni = (dc.flags >> 8) & 0xFFu; /* allele count shared by the column's sites, 0 if they differ (layout kernel) */
        const int w4_all = (dc.depth + 15) >> 4;
        const int w4_a = ((int) c.d1 + 15) >> 4, w4_b = (int) c.d1 >> 4;
        /* narrow column: every array entry and every parent cell has a lane of its own */
        const bool narrow = (C >> ush) <= (uint32_t) WAVE && C1 <= (uint32_t) WAVE && C2 <= (uint32_t) WAVE && XE_NARROW_ON;
        if (MODE == 1 ? !narrow : (MODE == 2 && narrow)) continue;
        if (MODE != 2 && narrow) {
            const bool have_a = c.a_part != nullptr, have_b = c.b_part != nullptr;
            const bool ia = have_a && (uint32_t) lane < C1, ib = have_b && (uint32_t) lane < C2;
            uint32_t npa[2] = {ia ? c.a_np[lane] : 0u, 0u}, npb[2] = {ib ? c.b_np[lane] : 0u, 0u};
            uint64_t Pa[2] = {ia ? c.a_part[lane] : 0ull, 0ull}, Pb[2] = {ib ? c.b_part[lane] : 0ull, 0ull};
            XE_T(1);
            int bad = xe_verify_side<1>(have_a, Pa, npa, C1, c.d1, c.Ma, c.Pa, c.out_a, c.in_a, inv,
                                        (c.flags & MRP_XF_OUT_A_PAIRED) != 0, (c.flags & MRP_XF_IN_A_PAIRED) != 0, lane);
            bad |= xe_verify_side<1>(have_b, Pb, npb, C2, c.d2, c.Mb, c.Pb, c.out_b, c.in_b, inv,
                                     (c.flags & MRP_XF_OUT_B_PAIRED) != 0, (c.flags & MRP_XF_IN_B_PAIRED) != 0, lane);
            if ((!c.a_part && C1 != 1u) || (!c.b_part && C2 != 1u) || (int) c.d1 + (int) c.d2 != dc.depth) bad |= MRP_ENGINE_ERR_RANGE;
            if (bad) { atomicOr(err, bad); atomicOr(err_hmm + col_hmm[col], bad); }
            const uint32_t U = C >> ush;
            if (__any(bad != 0)) { /* the level is discarded by the host; keep the arrays defined meanwhile */
                if ((uint32_t) lane < U) { cell_np[c.x_cell_off + lane] = 0u; cell_cost[c.x_cell_off + lane] = 0u; }
                continue;
            }
            /* the lane's cell, its parents' partitions and transitions (from the lanes that hold them) */
            uint32_t c1, c2;
            cross_cell(((uint32_t) lane < U ? (uint32_t) lane : 0u) << ush, C2, inv, a_cells_paired, b_cells_paired, c1, c2);
            const uint32_t n1 = (uint32_t) __builtin_amdgcn_ds_bpermute((int) (c1 << 2), (int) npa[0]);
            const uint32_t n2 = (uint32_t) __builtin_amdgcn_ds_bpermute((int) (c2 << 2), (int) npb[0]);
            const uint64_t p1 = ((uint64_t) (uint32_t) __builtin_amdgcn_ds_bpermute((int) (c1 << 2), (int) (uint32_t) (Pa[0] >> 32)) << 32) |
                                (uint32_t) __builtin_amdgcn_ds_bpermute((int) (c1 << 2), (int) (uint32_t) Pa[0]);
            const uint64_t p2 = ((uint64_t) (uint32_t) __builtin_amdgcn_ds_bpermute((int) (c2 << 2), (int) (uint32_t) (Pb[0] >> 32)) << 32) |
                                (uint32_t) __builtin_amdgcn_ds_bpermute((int) (c2 << 2), (int) (uint32_t) Pb[0]);
            const uint64_t P = c.d1 < 64 ? (p1 | (p2 << c.d1)) : p1; /* mergePartitionsOrMasks, partitions.c:21-28 */
            const uint32_t npv = np_entry(cross_np(c, inv, c1, c2, n1, n2));
            XE_T(2);
            /* the sites in chunks of whole sites of at most 16 slots; `ends` marks the last allele slot of every site */
            const uint32_t *aoff = A_uni ? nullptr : chunks[dc.chunk].allele_offset + dc.site_start;
            uint32_t sel[16];
#pragma unroll
            for (int w = 0; w < 16; w++) sel[w] = ((uint32_t) ((P >> (4 * w)) & 0xFull) * 0x00204081u) & 0x01010101u;
            uint32_t cost = 0, site0 = 0, sl0 = 0, m1 = 0xFFFFFFFFu, m2 = 0xFFFFFFFFu;
            if (A_uni == 2u && 2u * U < 2u * (uint32_t) dc.n_sites && XE_SLOTS_ON) { /* few cells, many sites: lanes along the slots */
                const uint32_t n_slots = 2u * (uint32_t) dc.n_sites;
                for (uint32_t s0 = 0; s0 < n_slots; s0 += WAVE) {
                    const uint32_t nsl = min((uint32_t) WAVE, n_slots - s0);
                    const int64_t sg = dc.slot_off + s0;
                    switch (w4_all) {
                    case 0: case 1: cost += xe_slots_chunk<1>(slot_bytes, slot_total, sg, nsl, U, P, lane); break;
                    case 2: cost += xe_slots_chunk<2>(slot_bytes, slot_total, sg, nsl, U, P, lane); break;
                    case 3: cost += xe_slots_chunk<3>(slot_bytes, slot_total, sg, nsl, U, P, lane); break;
                    default: cost += xe_slots_chunk<4>(slot_bytes, slot_total, sg, nsl, U, P, lane); break;
                    }
                }
                site0 = (uint32_t) dc.n_sites;
            }
            while (site0 < (uint32_t) dc.n_sites) {
                uint32_t cnt, nsl, ends = 0;
                if (A_uni && A_uni <= 16u) {
                    cnt = min((uint32_t) dc.n_sites - site0, 16u / A_uni);
                    nsl = cnt * A_uni;
                    if (A_uni == 2u) ends = 0xAAAAu & ((1u << nsl) - 1u);
                    else for (uint32_t s_ = 1; s_ <= cnt; s_++) ends |= 1u << (s_ * A_uni - 1u);
                } else { /* lane s looks at the end of site site0 + s */
                    const uint32_t *ao = A_uni ? chunks[dc.chunk].allele_offset + dc.site_start : aoff;
                    const uint32_t base = ao[site0];
                    const bool have = site0 + (uint32_t) lane < (uint32_t) dc.n_sites;
                    const uint32_t end = have ? ao[site0 + lane + 1] - base : 0xFFFFFFFFu;
                    const bool fits = have && end <= 16u;
                    cnt = (uint32_t) __popcll(__ballot(fits));
                    nsl = cnt ? (uint32_t) __builtin_amdgcn_readlane((int) end, (int) cnt - 1) : 0u;
                    ends = (fits && end >= 1u) ? 1u << (end - 1u) : 0u;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) ends |= (uint32_t) __shfl_xor((int) ends, o, WAVE);
                    if (cnt == 0) { /* a site of more than 16 alleles: slot by slot, the site ends with its last slot */
                        cnt = 1u;
                        nsl = ao[site0 + 1] - base;
                        const int64_t sg = dc.slot_off + sl0;
                        for (uint32_t a_ = 0; a_ < nsl; a_++) {
                            const uint32_t t = slot_dot(slot_bytes + (sg + a_) * 16, P, 0, w4_all);
                            m1 = min(m1, t);
                            m2 = min(m2, slot_total[sg + a_] - t);
                        }
                        cost += m1 + m2; m1 = 0xFFFFFFFFu; m2 = 0xFFFFFFFFu;
                        site0 += 1u; sl0 += nsl;
                        continue;
                    }
                }
                nsl = (uint32_t) __builtin_amdgcn_readfirstlane((int) nsl); /* (wave-uniform by construction; said again for the compiler) */
                ends = (uint32_t) __builtin_amdgcn_readfirstlane((int) ends);
                const int64_t sg = dc.slot_off + sl0;
                switch (w4_all) {
                case 0: case 1: cost += xe_narrow_chunk<1>(slot_bytes, slot_total, sg, nsl, ends, sel, lane, m1, m2); break;
                case 2: cost += xe_narrow_chunk<2>(slot_bytes, slot_total, sg, nsl, ends, sel, lane, m1, m2); break;
                case 3: cost += xe_narrow_chunk<3>(slot_bytes, slot_total, sg, nsl, ends, sel, lane, m1, m2); break;
                default: cost += xe_narrow_chunk<4>(slot_bytes, slot_total, sg, nsl, ends, sel, lane, m1, m2); break;
                }
                site0 += cnt;
                sl0 += nsl;
            }
            XE_T(3);
            if ((uint32_t) lane < U) {
                cell_np[c.x_cell_off + lane] = npv;
                cell_cost[c.x_cell_off + lane] = cost;
            }
            XE_T(4);
            continue;
        }
        if constexpr (MODE != 1) {
        /* Everything the column needs from HBM is requested HERE, in one round trip behind the two descriptors: the parents'
         * transitions and partitions (parent cells lane and lane + 64 of either side stay in the lane's registers: the pair
         * order check, the transition terms and the table rows all start from them) and, when the column's sites share their
         * allele count (so that the first table fill's slots are known from the descriptor alone), that fill's packed bytes
         * and byte sums.  Round 2 asked for them one after the other (transitions, order check, packed bytes, partitions):
         * four dependent trips to memory per column, half of the kernel's time. */
        const bool have_a = c.a_part != nullptr, have_b = c.b_part != nullptr;
        uint32_t npa[2], npb[2];
        uint64_t Pa[2], Pb[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const uint32_t i = (uint32_t) lane + (uint32_t) u * WAVE;
            const bool ia = have_a && i < C1, ib = have_b && i < C2;
            npa[u] = ia ? c.a_np[i] : 0u;
            Pa[u] = ia ? c.a_part[i] : 0ull;
            npb[u] = ib ? c.b_np[i] : 0u;
            Pb[u] = ib ? c.b_part[i] : 0ull;
        }
        constexpr int XE_PRE = XE_ROWS * 16 / WAVE;
        const uint32_t pre_cnt = A_uni ? min((uint32_t) dc.n_sites, slot_room / A_uni) : 0u;
        const uint32_t pre_nsl = pre_cnt * A_uni; /* 0: the first fill asks for its slots itself */
        uint32_t pre_rows[XE_PRE], pre_tot = 0u;
#pragma unroll
        for (int k = 0; k < XE_PRE; k++) {
            const uint32_t i = (uint32_t) lane + (uint32_t) k * WAVE;
            pre_rows[k] = i < pre_nsl * 16u ? slot_bytes[dc.slot_off * 16 + i] : 0u;
        }
        if ((uint32_t) lane < pre_nsl) pre_tot = slot_total[dc.slot_off + lane];
#ifdef XE_CLOCK
        if (__builtin_amdgcn_readfirstlane((int) (npa[0] ^ npb[0] ^ pre_rows[0] ^ (uint32_t) Pa[0] ^ (uint32_t) Pb[0])) == 0x7EADBEEF) continue; /* wait for the loads: section 1 */
        XE_T(1);
#endif
        xe_stage_transitions(c, inv, tra, trb, lane, npa, npb);
        int bad = xe_verify_side(have_a, Pa, npa, C1, c.d1, c.Ma, c.Pa, c.out_a, c.in_a, inv,
                                 (c.flags & MRP_XF_OUT_A_PAIRED) != 0, (c.flags & MRP_XF_IN_A_PAIRED) != 0, lane);
        bad |= xe_verify_side(have_b, Pb, npb, C2, c.d2, c.Mb, c.Pb, c.out_b, c.in_b, inv,
                              (c.flags & MRP_XF_OUT_B_PAIRED) != 0, (c.flags & MRP_XF_IN_B_PAIRED) != 0, lane);
        if ((!c.a_part && C1 != 1u) || (!c.b_part && C2 != 1u) || C1 > 128u || C2 > 128u || (int) c.d1 + (int) c.d2 != dc.depth) bad |= MRP_ENGINE_ERR_RANGE;
        if (bad) { atomicOr(err, bad); atomicOr(err_hmm + col_hmm[col], bad); }
        if (__any(bad != 0)) { /* the level is discarded by the host; keep the arrays defined meanwhile */
            for (uint32_t e = lane; e < (C >> ush); e += WAVE) { cell_np[c.x_cell_off + e] = 0u; cell_cost[c.x_cell_off + e] = 0u; }
            continue;
        }
        const uint32_t *aoff = A_uni ? nullptr : chunks[dc.chunk].allele_offset + dc.site_start;
        uint32_t site0 = 0, sl0 = 0;
        XE_T(2); /* transition terms, order check */
        /* one table fill and the cells' costs over its sites; the first is called with the registers above, the others (columns
         * of many sites or alleles: the low levels) ask again, so that nothing of the front end stays live over the cells */
        auto fill_and_cost = [&](const bool first_chunk, uint64_t Qa0, uint64_t Qa1, uint64_t Qb0, uint64_t Qb1) {
            /* as many whole sites as fit */
            uint32_t cnt, nsl;
            uint64_t ends = 0; /* bit (slot): the slot is the last allele of its site */
            if (A_uni) {
                cnt = min((uint32_t) dc.n_sites - site0, slot_room / A_uni);
                nsl = cnt * A_uni;
                for (uint32_t s_ = 1; s_ <= cnt; s_++) ends |= 1ull << (s_ * A_uni - 1u);
            } else { /* lane s looks at the end of site site0 + s */
                const uint32_t base = aoff[site0];
                const bool have = site0 + (uint32_t) lane < (uint32_t) dc.n_sites;
                const uint32_t end = have ? aoff[site0 + lane + 1] - base : 0xFFFFFFFFu;
                const bool fits = have && end <= slot_room;
                cnt = (uint32_t) __popcll(__ballot(fits));
                nsl = cnt ? (uint32_t) __builtin_amdgcn_readlane((int) end, (int) cnt - 1) : 0u;
                ends = (fits && end >= 1u) ? 1ull << (end - 1u) : 0ull;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) ends |= __shfl_xor(ends, o, WAVE);
                ends = ((uint64_t) (uint32_t) __builtin_amdgcn_readfirstlane((int) (ends >> 32)) << 32) |
                       (uint32_t) __builtin_amdgcn_readfirstlane((int) ends); /* wave-uniform: the site loop branches on scalars */
            }
            const int64_t slot_g = dc.slot_off + sl0;
            if (cnt == 0) {
                /* A site with more alleles than the tables hold for this many parent cells (rare): its cells are costed
                 * one by one from their merged partitions, straight from the packed bytes in HBM. */
                const uint32_t A = A_uni ? A_uni : aoff[site0 + 1] - aoff[site0];
                for (uint32_t en = lane; en < (C >> ush); en += WAVE) {
                    const uint32_t e = en << ush;
                    uint32_t c1, c2;
                    cross_cell(e, C2, inv, a_cells_paired, b_cells_paired, c1, c2);
                    const uint64_t P = cross_partition(c, c1, c2);
                    uint32_t m1 = 0xFFFFFFFFu, m2 = 0xFFFFFFFFu;
                    for (uint32_t a_ = 0; a_ < A; a_++) {
                        const uint32_t t = slot_dot(slot_bytes + (slot_g + a_) * 16, P, 0, w4_all);
                        m1 = min(m1, t);
                        m2 = min(m2, slot_total[slot_g + a_] - t);
                    }
                    const int64_t o = c.x_cell_off + en;
                    if (first_chunk) {
                        cell_np[o] = np_entry(xe_np(tra[c1], trb[c2]));
                        cell_cost[o] = m1 + m2;
                    } else cell_cost[o] += m1 + m2;
                }
                site0 += 1u;
                sl0 += A;
                return;
            }
            const uint32_t ST = (nsl + 3u) & ~3u; /* row stride: 16-byte rows */
            uint32_t *tb = tab + C1 * ST;
            {
                /* the packed bytes and byte sums of the slots: one coalesced read (the first fill's is in flight since the top of
                 * the column), then LDS */
                if (first_chunk && pre_nsl) {
#pragma unroll
                    for (int k = 0; k < XE_PRE; k++) {
                        const uint32_t i = (uint32_t) lane + (uint32_t) k * WAVE;
                        if (i < nsl * 16u) rowbuf[i] = pre_rows[k];
                    }
                    if ((uint32_t) lane < nsl) totbuf[lane] = pre_tot;
                } else {
                    for (uint32_t i = lane; i < nsl * 16u; i += WAVE) rowbuf[i] = slot_bytes[slot_g * 16 + i];
                    if ((uint32_t) lane < nsl) totbuf[lane] = slot_total[slot_g + lane];
                }
                wave_lds_fence();
                if (Cs >= 24u) { /* many parent cells: a lane per parent cell */
                    const uint32_t *rowb = rowbuf + 4 * w4_b;
                    switch (w4_a) {
                    case 0: case 1: xe_fill_side<1, false>(tab, ST, Qa0, Qa1, C1, 0u, 0u, nsl, rowbuf, totbuf, lane); break;
                    case 2: xe_fill_side<2, false>(tab, ST, Qa0, Qa1, C1, 0u, 0u, nsl, rowbuf, totbuf, lane); break;
                    case 3: xe_fill_side<3, false>(tab, ST, Qa0, Qa1, C1, 0u, 0u, nsl, rowbuf, totbuf, lane); break;
                    default: xe_fill_side<4, false>(tab, ST, Qa0, Qa1, C1, 0u, 0u, nsl, rowbuf, totbuf, lane); break;
                    }
                    switch (min(w4_all, 4) - min(w4_b, 3)) { /* d1 = 64: side B has no reads, any group of zero bits will do */
                    case 0: case 1: xe_fill_side<1, true>(tb, ST, Qb0, Qb1, C2, c.d1, 16u * min(w4_b, 3), nsl, rowbuf + 4 * min(w4_b, 3), totbuf, lane); break;
                    case 2: xe_fill_side<2, true>(tb, ST, Qb0, Qb1, C2, c.d1, 16u * w4_b, nsl, rowb, totbuf, lane); break;
                    case 3: xe_fill_side<3, true>(tb, ST, Qb0, Qb1, C2, c.d1, 16u * w4_b, nsl, rowb, totbuf, lane); break;
                    default: xe_fill_side<4, true>(tb, ST, Qb0, Qb1, C2, c.d1, 16u * w4_b, nsl, rowb, totbuf, lane); break;
                    }
                } else { /* few parent cells (the low levels: long columns of few cells): lanes along (slot, parent cell); the
                          * partition of parent cell cc comes from lane cc's register */
                    for (uint32_t i0 = 0; i0 < C1 * nsl; i0 += WAVE) {
                        const uint32_t idx = i0 + (uint32_t) lane;
                        const bool act = idx < C1 * nsl;
                        const uint32_t slot = act ? idx / C1 : 0u, cc = act ? idx - slot * C1 : 0u;
                        const uint64_t P = ((uint64_t) (uint32_t) __shfl((int) (uint32_t) (Qa0 >> 32), (int) cc, WAVE) << 32) |
                                           (uint32_t) __shfl((int) (uint32_t) Qa0, (int) cc, WAVE);
                        if (act) {
                            const uint32_t t = slot_dot(rowbuf + slot * 16, P, 0, w4_a);
                            tab[cc * ST + slot] = t | ((totbuf[slot] - t) << 16);
                        }
                    }
                    for (uint32_t i0 = 0; i0 < C2 * nsl; i0 += WAVE) {
                        const uint32_t idx = i0 + (uint32_t) lane;
                        const bool act = idx < C2 * nsl;
                        const uint32_t slot = act ? idx / C2 : 0u, cc = act ? idx - slot * C2 : 0u;
                        const uint64_t Pr = ((uint64_t) (uint32_t) __shfl((int) (uint32_t) (Qb0 >> 32), (int) cc, WAVE) << 32) |
                                            (uint32_t) __shfl((int) (uint32_t) Qb0, (int) cc, WAVE);
                        if (act) {
                            const uint64_t P = c.d1 < 64 ? Pr << c.d1 : 0ull;
                            const uint32_t t = slot_dot(rowbuf + slot * 16, P, w4_b, w4_all);
                            tb[cc * ST + slot] = t - (t << 16);
                        }
                    }
                }
            }
            /* biallelic sites, unit level (the shipped ONT parameters: every level above the first): the cells' loop below takes the
             * slots four at a time without looking at `ends`; the rows' padding up to a multiple of four slots is zeroed (a pad pair
             * then costs 0 + 0) */
            const bool fast2 = units && a_cells_paired && A_uni == 2u && XE_FAST2_ON;
            if (fast2 && ST > nsl)
                for (uint32_t i = lane; i < Cs; i += WAVE) {
                    uint32_t *rw = tab + i * ST; /* (side B's rows follow side A's: tb = tab + C1 * ST) */
                    for (uint32_t s_ = nsl; s_ < ST; s_++) rw[s_] = 0u;
                }
            wave_lds_fence();
            XE_T(3); /* table fill */
            if (fast2) {
                /* the grid walk of the general branch below -- row r = pair (2r, 2r + 1) of side A cells, lane = side B cell h -- with
                 * everything that does not depend on the row hoisted: the lane's table row (up to 16 slots in registers), its two
                 * transition terms per direction, and the store addresses, which advance by a constant.  Per unit: an add and half a
                 * packed min per slot, two adds per site, seven operations for the transitions. */
                const uint32_t W = C2 >= 33u ? 64u : (C2 >= 17u ? 32u : (C2 >= 9u ? 16u : (C2 >= 5u ? 8u : (C2 >= 3u ? 4u : (C2 >= 2u ? 2u : 1u)))));
                const uint32_t wsh = 31u - (uint32_t) __builtin_clz(W), R = WAVE >> wsh;
                const uint32_t hl = (uint32_t) lane & (W - 1u), rl = (uint32_t) lane >> wsh, rows = C1 >> 1;
                const uint32_t nq = ST >> 2;
                for (uint32_t hb = 0; hb < C2; hb += WAVE) {
                    const uint32_t h = hb + hl;
                    const bool hv = h < C2;
                    const uint32_t hc = hv ? h : 0u;
                    const uint2 tbh = trb[hc];
                    const uint32_t bn_lo = tbh.x & 0xFFFFu, bn_hi = tbh.x >> 16, bp_lo = tbh.y & 0xFFFFu, bp_hi = tbh.y >> 16;
                    const uint32_t *pb = tb + hc * ST;
                    uint4 bq[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) bq[q] = (uint32_t) q < nq ? *reinterpret_cast<const uint4 *>(pb + 4 * q) : make_uint4(0, 0, 0, 0);
                    uint32_t *pn = cell_np + c.x_cell_off + ((int64_t) rl * C2 + h), *pc = cell_cost + c.x_cell_off + ((int64_t) rl * C2 + h);
                    const uint32_t stride = R * C2;
                    /* two rows per step (r and r + R), their table rows and transition terms requested together; NQ = groups of four
                     * slots, a compile-time constant per instantiation: the body is straight-line code */
                    auto walk = [&](auto nq_tag) {
                        constexpr int NQ = decltype(nq_tag)::value;
                        constexpr bool TWO = NQ <= 2; /* (two rows of 12 or 16 slots in flight cost the fifth wave per SIMD its registers) */
                        for (uint32_t r = rl; r < rows; r += (TWO ? 2u : 1u) * R) {
                            const uint32_t r2 = r + R;
                            const bool v2 = TWO && r2 < rows;
                            const uint32_t r2c = v2 ? r2 : r;
                            const uint32_t *pa0 = tab + 2u * r * ST, *pa1 = tab + 2u * r2c * ST;
                            uint4 a0[NQ], a1[TWO ? NQ : 1];
#pragma unroll
                            for (int q = 0; q < NQ; q++) { a0[q] = *reinterpret_cast<const uint4 *>(pa0 + 4 * q); if (TWO) a1[q] = *reinterpret_cast<const uint4 *>(pa1 + 4 * q); }
                            const uint2 ta0 = tra[2u * r], ta1 = TWO ? tra[2u * r2c] : make_uint2(0u, 0u);
                            uint32_t cost0 = 0, cost1 = 0;
#pragma unroll
                            for (int q = 0; q < NQ; q++) {
                                const uint32_t m01 = pk_min_u16(a0[q].x + bq[q].x, a0[q].y + bq[q].y), m23 = pk_min_u16(a0[q].z + bq[q].z, a0[q].w + bq[q].w);
                                cost0 += (m01 & 0xFFFFu) + (m01 >> 16) + (m23 & 0xFFFFu) + (m23 >> 16);
                                if (TWO) {
                                    const uint32_t n01 = pk_min_u16(a1[q].x + bq[q].x, a1[q].y + bq[q].y), n23 = pk_min_u16(a1[q].z + bq[q].z, a1[q].w + bq[q].w);
                                    cost1 += (n01 & 0xFFFFu) + (n01 >> 16) + (n23 & 0xFFFFu) + (n23 >> 16);
                                }
                            }
                            const uint32_t nx0 = (ta0.x & 0x7FFFFFFFu) + ((int32_t) ta0.x < 0 ? bn_hi : bn_lo), pv0 = (ta0.y & 0x7FFFFFFFu) + ((int32_t) ta0.y < 0 ? bp_hi : bp_lo);
                            const uint32_t nx1 = (ta1.x & 0x7FFFFFFFu) + ((int32_t) ta1.x < 0 ? bn_hi : bn_lo), pv1 = (ta1.y & 0x7FFFFFFFu) + ((int32_t) ta1.y < 0 ? bp_hi : bp_lo);
                            if (hv) {
                                if (first_chunk) { pn[0] = (nx0 >> nsh) | ((pv0 >> psh) << 16); if (v2) pn[stride] = (nx1 >> nsh) | ((pv1 >> psh) << 16); }
                                else { cost0 += pc[0]; if (v2) cost1 += pc[stride]; }
                                pc[0] = cost0;
                                if (v2) pc[stride] = cost1;
                            }
                            pn += (TWO ? 2u : 1u) * stride; pc += (TWO ? 2u : 1u) * stride;
                        }
                    };
                    switch (nq) {
                    case 1: walk(std::integral_constant<int, 1>()); break;
                    case 2: walk(std::integral_constant<int, 2>()); break;
                    case 3: walk(std::integral_constant<int, 3>()); break;
                    default: walk(std::integral_constant<int, 4>()); break;
                    }
                }
            } else if (a_cells_paired) {
                /* The cells of the column are a grid: row r = pair (2r, 2r + 1) of side A cells, position h = side B cell;
                 * cells e = 2 (r C2 + h) and e + 1 are complements and share their cost.  A lane keeps ONE h: its side B
                 * table row and transition terms stay in registers while it walks down the rows; with C2 <= 32 a wave
                 * takes several rows per step.  Per cell that leaves an add and a packed min per allele slot. */
                const uint32_t W = C2 >= 33u ? 64u : (C2 >= 17u ? 32u : (C2 >= 9u ? 16u : (C2 >= 5u ? 8u : (C2 >= 3u ? 4u : (C2 >= 2u ? 2u : 1u)))));
                const uint32_t wsh = 31u - (uint32_t) __builtin_clz(W), R = WAVE >> wsh;
                const uint32_t hl = (uint32_t) lane & (W - 1u), rl = (uint32_t) lane >> wsh, rows = C1 >> 1;
                for (uint32_t hb = 0; hb < C2; hb += WAVE) {
                    const uint32_t h = hb + hl;
                    const bool hv = h < C2;
                    const uint32_t hc = hv ? h : 0u, g = b_cells_paired ? (hc ^ 1u) : hc;
                    const uint2 tbh = trb[hc], tbg = trb[g];
                    const uint32_t *pb = tb + hc * ST;
                    uint4 b0 = make_uint4(0, 0, 0, 0), b1 = make_uint4(0, 0, 0, 0);
                    if (nsl <= 8u) { b0 = *reinterpret_cast<const uint4 *>(pb); if (nsl > 4u) b1 = *reinterpret_cast<const uint4 *>(pb + 4); }
                    for (uint32_t r = rl; r < rows; r += R) {
                        const uint32_t *pa = tab + 2u * r * ST;
                        uint32_t cost;
                        if (nsl <= 8u) {
                            uint32_t m = 0xFFFFFFFFu;
                            cost = 0;
                            const uint4 a0 = *reinterpret_cast<const uint4 *>(pa);
                            const uint32_t x0[4] = {a0.x + b0.x, a0.y + b0.y, a0.z + b0.z, a0.w + b0.w};
#pragma unroll
                            for (int j = 0; j < 4; j++)
                                if ((uint32_t) j < nsl) {
                                    m = pk_min_u16(m, x0[j]);
                                    if (ends & (1ull << j)) { cost += (m & 0xFFFFu) + (m >> 16); m = 0xFFFFFFFFu; }
                                }
                            if (nsl > 4u) {
                                const uint4 a1 = *reinterpret_cast<const uint4 *>(pa + 4);
                                const uint32_t x1[4] = {a1.x + b1.x, a1.y + b1.y, a1.z + b1.z, a1.w + b1.w};
#pragma unroll
                                for (int j = 0; j < 4; j++)
                                    if ((uint32_t) (4 + j) < nsl) {
                                        m = pk_min_u16(m, x1[j]);
                                        if (ends & (1ull << (4 + j))) { cost += (m & 0xFFFFu) + (m >> 16); m = 0xFFFFFFFFu; }
                                    }
                            }
                        } else cost = xe_cost(pa, pb, nsl, ends);
                        if (hv && units) { /* one entry per pair: a dword per lane and array, lanes along h */
                            const int64_t o = c.x_cell_off + ((int64_t) r * C2 + h);
                            if (first_chunk) cell_np[o] = np_entry(xe_np(tra[2u * r], tbh));
                            else cost += cell_cost[o];
                            cell_cost[o] = cost;
                        } else if (hv) {
                            const int64_t o = c.x_cell_off + 2 * ((int64_t) r * C2 + h);
                            if (first_chunk) {
                                const uint4 ta = *reinterpret_cast<const uint4 *>(tra + 2u * r); /* terms of cells 2r and 2r + 1 */
                                xe_u32x2 nv = {xe_np(make_uint2(ta.x, ta.y), tbh), xe_np(make_uint2(ta.z, ta.w), tbg)};
                                *reinterpret_cast<xe_u32x2 *>(cell_np + o) = nv;
                            } else cost += cell_cost[o];
                            xe_u32x2 cv = {cost, cost};
                            *reinterpret_cast<xe_u32x2 *>(cell_cost + o) = cv;
                        }
                    }
                }
            } else {
                for (uint32_t en = lane; en < (C >> ush); en += WAVE) {
                    const uint32_t e = en << ush;
                    uint32_t c1, c2;
                    cross_cell(e, C2, inv, a_cells_paired, b_cells_paired, c1, c2);
                    const uint32_t cost = xe_cost(tab + c1 * ST, tb + c2 * ST, nsl, ends);
                    const int64_t o = c.x_cell_off + en;
                    if (first_chunk) {
                        cell_np[o] = np_entry(xe_np(tra[c1], trb[c2]));
                        cell_cost[o] = cost;
                    } else cell_cost[o] += cost;
                }
            }
            wave_lds_fence(); /* the tables are rewritten by the next chunk of sites / the next column */
            XE_T(4); /* cells */
            site0 += cnt;
            sl0 += nsl;
        };
        if ((uint32_t) dc.n_sites > 0u) fill_and_cost(true, Pa[0], Pa[1], Pb[0], Pb[1]);
        while (site0 < (uint32_t) dc.n_sites) {
            uint64_t Qa[2], Qb[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const uint32_t i = (uint32_t) lane + (uint32_t) u * WAVE;
                Qa[u] = (have_a && i < C1) ? c.a_part[i] : 0ull;
                Qb[u] = (have_b && i < C2) ? c.b_part[i] : 0ull;
            }
            fill_and_cost(false, Qa[0], Qa[1], Qb[0], Qb[1]);
        }
        } /* MODE != 1 */
    }
#ifdef XE_CLOCK
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    XE_T(5); /* the stores in flight at the end */
    if (lane == 0 && (blockIdx.x & 63u) == 0u) for (int i = 0; i < 8; i++) atomicAdd((unsigned long long *) (err + 4) + i, (unsigned long long) xe_sec[i]); /* a sample of the waves */
#endif
}

hipError_t mrp_launch_cross_emit(const CrossCol *cols_dev, const MrpBatchDev &d, int32_t *err, const int32_t *col_hmm_dev, int32_t *err_hmm,
                                 int32_t level_max_cells, bool mostly_narrow, hipStream_t stream) {
    if (d.n_cols <= 0) return hipSuccess;
    int64_t wgs = (d.n_cols + XE_WAVES - 1) / XE_WAVES;
    /* level_max_cells: the level's largest cross product column by the host's static bounds; up to 128 cells (64 array entries of a
     * unit level) every column has a lane per entry */
    uint32_t cap = level_max_cells > 2 * WAVE && !mostly_narrow ? XE_CAP_WIDE : XE_CAP_NARROW;
    static const long xe_cap = getenv("MRP_XE_CAP") ? atol(getenv("MRP_XE_CAP")) : 0; /* (development) */
    if (xe_cap >= 256 && xe_cap <= 8192) cap = (uint32_t) xe_cap & ~3u;
    const size_t lds = (size_t) XE_WAVES * (cap + XE_ROWS * 16 + XE_ROWS + 512) * sizeof(uint32_t);
    const dim3 grid((unsigned) (wgs < (1 << 20) ? wgs : (1 << 20)));
    if (mostly_narrow && XE_NARROW_ON) {
        hipLaunchKernelGGL(mrp_cross_emit_kernel<1>, grid, dim3(XE_WAVES * WAVE), 0, stream, cols_dev, d.cols, d.chunks, d.n_cols, d.slot_bytes, d.slot_total,
                           const_cast<uint32_t *>(d.cell_np), d.cell_cost, err, col_hmm_dev, err_hmm, cap);
        /* (a column of at most 64 cells is narrow whatever its parents: they have no more cells than it has) */
        if (level_max_cells > WAVE)
            hipLaunchKernelGGL(mrp_cross_emit_kernel<2>, grid, dim3(XE_WAVES * WAVE), lds, stream, cols_dev, d.cols, d.chunks, d.n_cols, d.slot_bytes, d.slot_total,
                               const_cast<uint32_t *>(d.cell_np), d.cell_cost, err, col_hmm_dev, err_hmm, cap);
    } else
        hipLaunchKernelGGL(mrp_cross_emit_kernel<0>, grid, dim3(XE_WAVES * WAVE), lds, stream, cols_dev, d.cols, d.chunks, d.n_cols, d.slot_bytes, d.slot_total,
                           const_cast<uint32_t *>(d.cell_np), d.cell_cost, err, col_hmm_dev, err_hmm, cap);
    return hipGetLastError();
}

/* ------------------------------------------------------------------------------------------ */
/* column structure of a level (mrp_engine.h "column structure of a level, on the device")      */
/* ------------------------------------------------------------------------------------------ */
/* what one side (tiling path) contributes to the child column [cs, ce): the piece of pieces_of_path / align_pieces of the
 * reference's stRPHmm_fuse (hmm.c:283-372) and stRPHmm_alignColumns (hmm.c:374-507) that holds cs */
struct SidePiece {
    const uint64_t *part; const uint32_t *np; const int32_t *ncells, *nmerge;
    const int64_t *rbo;      /* the parent column's read_byte_off entries */
    int64_t leaf_rbo;        /* leaf: the read's pool offset */
    int32_t delta_site;      /* first site of the parent column (rbo moves on by the alleles between it and cs) */
    uint8_t depth, out, paired, cont;
    bool leaf;
};
static __device__ SidePiece structure_side(const StructureIn &in, const mrp_xpar *path, int n_path, int32_t ref_end, int32_t cs, int32_t ce,
                                           int *bad) {
    SidePiece r;
    r.part = nullptr; r.np = nullptr; r.ncells = nullptr; r.nmerge = nullptr; r.rbo = nullptr; r.leaf_rbo = 0; r.delta_site = cs;
    r.depth = 0; r.out = MRP_CONN_ZERO; r.paired = 0; r.cont = 0; r.leaf = false;
    /* last hmm of the path that starts at or before cs */
    int lo = 0, hi = n_path;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (path[mid].start <= cs) lo = mid + 1; else hi = mid; }
    const int pi = lo - 1;
    if (pi < 0 || cs >= path[pi].end) { /* gap column (hmm.c:335-359, :396-462): one cell, partition 0, depth 0 */
        const int32_t gap_end = lo < n_path ? path[lo].start : ref_end;
        if (ce > gap_end) *bad = 1;
        r.out = ce < gap_end ? MRP_CONN_IDENT : MRP_CONN_ZERO; /* a gap cut in two: the accept-mask connector of a depth-0 column */
        return r;
    }
    const mrp_xpar p = path[pi];
    if (p.seg < 0) { /* stRPHmm_construct (hmm.c:97-133): one column {1, 0} over the read's sites */
        if (ce > p.end) *bad = 1;
        r.part = in.leaf_part; r.np = in.leaf_np; r.ncells = in.leaf_count;
        r.leaf = true; r.leaf_rbo = p.col0; r.delta_site = p.start; r.depth = 1;
        r.out = ce < p.end ? MRP_CONN_IDENT : MRP_CONN_ZERO;
        r.paired = r.out == MRP_CONN_IDENT ? 1 : 0;
        r.cont = r.paired;
        return r;
    }
    const SegDev sg = in.segs[p.seg];
    const ResCol *pc = sg.cols + p.col0;
    int a = 0, b = p.n_cols; /* last column of the parent that starts at or before cs */
    while (a < b) { const int mid = (a + b) >> 1; if (pc[mid].start <= cs) a = mid + 1; else b = mid; }
    const int k = a - 1;
    if (k < 0) { *bad = 1; return r; }
    const ResCol c = pc[k];
    const int32_t pe = k + 1 < p.n_cols ? pc[k + 1].start : p.end;
    if (ce > pe) *bad = 1;
    const int64_t col = p.col0 + k;
    r.part = sg.part + col * in.stride; r.np = sg.np + col * in.stride; r.ncells = sg.n_cells + col;
    r.rbo = sg.rbo + c.rbo_off; r.delta_site = c.start; r.depth = c.depth;
    if (ce < pe) { r.out = MRP_CONN_IDENT; r.paired = c.depth > 0 ? 1 : 0; }          /* column.c:86-101 */
    else if (k + 1 < p.n_cols) { r.out = MRP_CONN_REAL; r.paired = c.cont; r.nmerge = sg.n_merge + col; }
    else { r.out = MRP_CONN_ZERO; r.paired = 0; }                                       /* hmm.c:324-331 */
    r.cont = r.paired;
    return r;
}

__global__ void __launch_bounds__(256) mrp_structure_kernel(StructureIn in) {
    const int64_t col = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= in.n_cols) return;
    /* the hmm the column belongs to: last one whose first column is at or before col */
    int64_t lo = 0, hi = in.n_hmms;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (in.xd[mid].col0 <= col) lo = mid + 1; else hi = mid; }
    const XDesc x = in.xd[lo - 1];
    const int k = (int) (col - x.col0);
    const bool last = k + 1 == x.n_cols;
    int32_t cs = in.col_start[col], ce = last ? x.ref_end : in.col_start[col + 1];
    int bad = (ce <= cs || cs < x.ref_start || ce > x.ref_end) ? 1 : 0;
    if (bad) { cs = x.ref_start; ce = x.ref_end; } /* (the chunk's tables are read below whatever the flag says: inside the hmm's interval) */
    const mrp_xpar *pa = in.par + x.par0, *pb = pa + x.n_a;
    const SidePiece A = structure_side(in, pa, x.n_a, x.ref_end, cs, ce, &bad);
    const SidePiece B = structure_side(in, pb, x.n_b, x.ref_end, cs, ce, &bad);
    const DevChunk ch = in.chunks[x.chunk];
    const int depth = (int) A.depth + (int) B.depth;
    if (depth > MRP_MAX_READ_PARTITIONING_DEPTH) bad = 1;
    const int64_t read_off = x.read0 + in.col_roff[col];
    /* the host sized the column's slice of the read offsets from ITS depth: a disagreement must not run into the next column's */
    if (!last && depth != in.col_roff[col + 1] - in.col_roff[col]) bad = 1;
    if (!bad) { /* the column's reads: side A's then side B's (partitions.c:21-28); profileSeq.c:41-47 */
        int64_t *dst = in.rbo + read_off;
        const uint32_t a_cs = ch.allele_offset[cs];
        if (A.depth) {
            const int64_t delta = (int64_t) a_cs - (int64_t) ch.allele_offset[A.delta_site];
            if (A.leaf) dst[0] = A.leaf_rbo + delta;
            else for (int i = 0; i < A.depth; i++) dst[i] = A.rbo[i] + delta;
        }
        if (B.depth) {
            const int64_t delta = (int64_t) a_cs - (int64_t) ch.allele_offset[B.delta_site];
            if (B.leaf) dst[A.depth] = B.leaf_rbo + delta;
            else for (int i = 0; i < B.depth; i++) dst[A.depth + i] = B.rbo[i] + delta;
        }
    }
    PlanCol o;
    o.a_part = A.part; o.b_part = B.part; o.a_np = A.np; o.b_np = B.np;
    o.a_ncells = A.ncells; o.b_ncells = B.ncells; o.a_nmerge = A.nmerge; o.b_nmerge = B.nmerge;
    o.read_off = read_off;
    o.slot_off = x.slot0 + (int64_t) (ch.allele_offset[cs] - ch.allele_offset[x.ref_start]);
    o.site_start = cs; o.n_sites = ce - cs; o.depth = bad ? 0 : depth;
    o.n_slots = (int32_t) (ch.allele_offset[ce] - ch.allele_offset[cs]);
    o.chunk = x.chunk;
    const int32_t uniform = ch.same_until[cs] >= ce ? (int32_t) ch.allele_number[cs] : 0;
    o.uniform_alleles = uniform;
    o.d1 = bad ? 0 : A.depth; o.d2 = bad ? 0 : B.depth; o.out_a = A.out; o.out_b = B.out;
    o.out_a_paired = A.paired; o.out_b_paired = B.paired;
    o.need_planes = (!in.fused && (uniform == 0 || (x.flags & MRP_FLAG_INCLUDE_ANCESTOR_SUB_PROB))) ? 1 : 0;
    o.last = last ? 1 : 0;
    o.pad = 0;
    if (bad) { o.a_part = nullptr; o.b_part = nullptr; o.a_np = nullptr; o.b_np = nullptr; o.a_ncells = nullptr; o.b_ncells = nullptr;
               o.a_nmerge = nullptr; o.b_nmerge = nullptr; o.out_a = o.out_b = MRP_CONN_ZERO; o.out_a_paired = o.out_b_paired = 0; }
    in.plan[col] = o;
    ResCol rc;
    rc.rbo_off = read_off; rc.start = cs; rc.depth = (uint8_t) o.depth; rc.cont = last ? 0 : (uint8_t) ((A.cont | B.cont) && !bad); rc.pad = 0;
    in.cols[col] = rc;
    in.col_hmm[col] = x.prune_pos;
    if (bad) { atomicOr(in.err, MRP_ENGINE_ERR_RANGE); atomicOr(in.err_hmm + x.prune_pos, MRP_ENGINE_ERR_RANGE); }
}

hipError_t mrp_launch_structure(const StructureIn &in, hipStream_t stream) {
    if (in.n_cols <= 0) return hipSuccess;
    hipLaunchKernelGGL(mrp_structure_kernel, dim3((unsigned) ((in.n_cols + 255) / 256)), dim3(256), 0, stream, in);
    return hipGetLastError();
}

/* ------------------------------------------------------------------------------------------ */
/* layout of a level: sizes, offsets and kernel descriptors from the parents' counts           */
/* ------------------------------------------------------------------------------------------ */
static __device__ __forceinline__ int32_t layout_count(const int32_t *p, int32_t S) {
    if (!p) return 1;
    const int32_t v = *p;
    return v < 1 ? 1 : (v > S ? S : v); /* (a discarded parent may hold anything: the level stays within its static bounds) */
}
static __device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
static __device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(v, o, WAVE); v = t > v ? t : v; }
    return v;
}

/* pass 1, one wave per hmm: C1, C2, Ma, Mb of every column; per hmm the sums the scan needs */
__global__ void __launch_bounds__(256) mrp_layout_count_kernel(const PlanCol *__restrict__ plan, const PlanHmm *__restrict__ ph, int64_t n_hmms,
                                                               int32_t S, uint32_t xflags, LayoutOut o) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int64_t w = (int64_t) blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE;
    if (w >= n_hmms) return;
    const PlanHmm h = ph[w];
    const bool ancestor = (h.flags & MRP_FLAG_INCLUDE_ANCESTOR_SUB_PROB) != 0;
    int64_t cells = 0, merge = 0, acells = 0, amerge = 0;
    int tf = 0, tg = 0, mc = 1, mm = 1;
    const bool units = (xflags & MRP_XF_UNITS) != 0;
    for (int k = lane; k < h.n_cols; k += WAVE) {
        const PlanCol c = plan[h.col0 + k];
        /* Every count is kept within the bound the HOST assumes for the column (mrp_side_bound: a pruned column has at most S cells
         * and never more than the bipartitions of its reads; rphmm_host.c r_cross_build sums exactly these products): a valid parent
         * never exceeds it, a discarded one may hold anything -- and a level launched without waiting for its totals (mrp_engine.cpp,
         * deferred launch) has arrays of the bounds' size. */
        const int S1 = (int) mrp_side_bound(c.d1, S), S2 = (int) mrp_side_bound(c.d2, S);
        const int C1 = layout_count(c.a_ncells, S1), C2 = layout_count(c.b_ncells, S2);
        int Ma = 0, Mb = 0;
        if (!c.last) {
            Ma = c.out_a == MRP_CONN_REAL ? layout_count(c.a_nmerge, S1) : (c.out_a == MRP_CONN_IDENT ? C1 : 1);
            Mb = c.out_b == MRP_CONN_REAL ? layout_count(c.b_nmerge, S2) : (c.out_b == MRP_CONN_IDENT ? C2 : 1);
        }
        uint16_t *dm = o.dims + 4 * (h.col0 + k);
        dm[0] = (uint16_t) C1; dm[1] = (uint16_t) C2; dm[2] = (uint16_t) Ma; dm[3] = (uint16_t) Mb;
        const int Cf = C1 * C2, Mf = Ma * Mb;
        /* MRP_XF_UNITS: a column some side of which has reads holds its cells in complement pairs, a merge column some side of
         * which is paired likewise: one array entry per pair */
        const int C = units && ((c.a_part && c.d1 > 0) || (c.b_part && c.d2 > 0)) ? Cf >> 1 : Cf;
        const int M = units && !c.last && (c.out_a_paired || c.out_b_paired) ? Mf >> 1 : Mf;
        cells += C; merge += M; acells += Cf; amerge += Mf;
        const int nt = (C + MRP_EMIT_TILE - 1) / MRP_EMIT_TILE;
        if (c.uniform_alleles != 0 && !ancestor) tf += nt; else tg += nt;
        mc = C > mc ? C : mc;
        mm = M > mm ? M : mm;
    }
    /* (an hmm has at most 2^31 cells: checked on the host against the static bounds) */
    const int lo = wave_sum_i32((int) (cells & 0xFFFF)), hi = wave_sum_i32((int) (cells >> 16));
    const int mlo = wave_sum_i32((int) (merge & 0xFFFF)), mhi = wave_sum_i32((int) (merge >> 16));
    const int alo = wave_sum_i32((int) (acells & 0xFFFF)), ahi = wave_sum_i32((int) (acells >> 16));
    const int amlo = wave_sum_i32((int) (amerge & 0xFFFF)), amhi = wave_sum_i32((int) (amerge >> 16));
    tf = wave_sum_i32(tf); tg = wave_sum_i32(tg); mc = wave_max_i32(mc); mm = wave_max_i32(mm);
    if (lane == 0) {
        LayoutTot t;
        t.cells = ((int64_t) hi << 16) + lo; t.merge = ((int64_t) mhi << 16) + mlo;
        t.acells = ((int64_t) ahi << 16) + alo; t.amerge = ((int64_t) amhi << 16) + amlo;
        t.tiles_fast = tf; t.tiles_gen = tg; t.max_cells = mc; t.max_merge = mm;
        o.tot[w] = t;
    }
}

/* pass 2: where every hmm starts (cells padded to a multiple of 4 per hmm), the totals, the DevHmm records -- an exclusive scan over the
 * hmms' sums in two small kernels of 256-thread workgroups, a tile of 256 hmms each:
 *   mrp_layout_tiles_kernel   the six sums of every tile (coalesced: thread i reads hmm 256 t + i);
 *   mrp_layout_scan_kernel    tile t adds up the tiles before it (at most a few hundred: 37 000 hmms are 145 tiles), scans its own 256
 *                             hmms in LDS and writes their bases and DevHmm records; the last tile writes the totals.
 * Through round 4 this was ONE workgroup of 1 024 threads walking the hmms in runs (two passes of strided 48-byte reads, 128 registers
 * and 48 spilled): 20-240 us alone -- and, needing a CU's whole register file to start, 1 ms on average once the device was full of
 * other batches' chain workgroups, on the critical path of every level (62 ms of summed kernel time per 1 152-chunk step). */
#define LAYOUT_TILE 256
static __device__ __forceinline__ void layout_sums_of(const LayoutTot &x, int64_t (&v)[6]) {
    v[0] = (x.cells + 3) & ~3ll; v[1] = x.merge; v[2] = x.tiles_fast; v[3] = x.tiles_gen; v[4] = x.acells; v[5] = x.amerge;
}
static __device__ __forceinline__ int64_t wave_incl_scan_i64(int64_t v, int lane) {
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) { const int64_t t = __shfl_up(v, o, WAVE); if (lane >= o) v += t; }
    return v;
}
__global__ void __launch_bounds__(LAYOUT_TILE) mrp_layout_tiles_kernel(int64_t n_hmms, LayoutOut o, int64_t *__restrict__ tile_sums) {
    __shared__ int64_t wsum[LAYOUT_TILE / WAVE][6];
    const int t = threadIdx.x, lane = t & (WAVE - 1), wave = t / WAVE;
    const int64_t i = (int64_t) blockIdx.x * LAYOUT_TILE + t;
    int64_t v[6] = {0, 0, 0, 0, 0, 0};
    if (i < n_hmms) layout_sums_of(o.tot[i], v);
#pragma unroll
    for (int q = 0; q < 6; q++) {
        int64_t x = v[q];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, WAVE);
        if (lane == 0) wsum[wave][q] = x;
    }
    __syncthreads();
    if (t < 6) {
        int64_t x = 0;
#pragma unroll
        for (int w = 0; w < LAYOUT_TILE / WAVE; w++) x += wsum[w][t];
        tile_sums[(int64_t) blockIdx.x * 6 + t] = x;
    }
}
__global__ void __launch_bounds__(LAYOUT_TILE) mrp_layout_scan_kernel(const PlanHmm *__restrict__ ph, int64_t n_hmms, LayoutOut o,
                                                                      const int64_t *__restrict__ tile_sums) {
    __shared__ int64_t wsum[LAYOUT_TILE / WAVE][6];
    __shared__ int64_t before[6];
    const int t = threadIdx.x, lane = t & (WAVE - 1), wave = t / WAVE;
    const int64_t tile = blockIdx.x, i = tile * LAYOUT_TILE + t;
    /* the tiles before this one */
    int64_t pre[6] = {0, 0, 0, 0, 0, 0};
    for (int64_t j = t; j < tile; j += LAYOUT_TILE)
#pragma unroll
        for (int q = 0; q < 6; q++) pre[q] += tile_sums[j * 6 + q];
#pragma unroll
    for (int q = 0; q < 6; q++) {
        int64_t x = pre[q];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, WAVE);
        if (lane == 0) wsum[wave][q] = x;
    }
    __syncthreads();
    if (t < 6) {
        int64_t x = 0;
#pragma unroll
        for (int w = 0; w < LAYOUT_TILE / WAVE; w++) x += wsum[w][t];
        before[t] = x;
    }
    __syncthreads();
    /* this tile's hmms */
    LayoutTot x{};
    int64_t v[6] = {0, 0, 0, 0, 0, 0}, incl[6];
    if (i < n_hmms) { x = o.tot[i]; layout_sums_of(x, v); }
#pragma unroll
    for (int q = 0; q < 6; q++) { incl[q] = wave_incl_scan_i64(v[q], lane); if (lane == WAVE - 1) wsum[wave][q] = incl[q]; }
    __syncthreads();
    int64_t excl[6];
#pragma unroll
    for (int q = 0; q < 6; q++) {
        int64_t bw = before[q], all = before[q];
#pragma unroll
        for (int w = 0; w < LAYOUT_TILE / WAVE; w++) { const int64_t y = wsum[w][q]; if (w < wave) bw += y; all += y; }
        excl[q] = bw + incl[q] - v[q];
        if (tile == (int64_t) gridDim.x - 1 && t == 0) o.totals[q] = all;
    }
    if (i < n_hmms) {
        const PlanHmm h = ph[i];
        LayoutBase b;
        b.cell0 = excl[0]; b.mcell0 = excl[1]; b.tile_fast0 = excl[2]; b.tile_gen0 = excl[3];
        o.base[i] = b;
        DevHmm d;
        d.col0 = h.col0; d.n_cols = h.n_cols; d.flags = h.flags; d.max_merge = x.max_merge; d.max_cells = x.max_cells;
        d.wide_idx = 0; d.pad = 0; d.n_cells = x.cells; d.n_merge = x.merge; d.cost_bound = h.cost_bound;
        o.hmms[i] = d;
    }
}

/* exclusive prefix sum over the wave (shuffles: the DPP scans of the prune kernel are defined further down) */
static __device__ __forceinline__ int wave_excl_scan_shfl(int v, int lane, int *total) {
    int x = v;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) { const int t = __shfl_up(x, o, WAVE); if (lane >= o) x += t; }
    *total = __shfl(x, WAVE - 1, WAVE);
    return x - v;
}

/* pass 3, one wave per hmm: the offsets of its columns and every per-column descriptor of the level's kernels */
__global__ void __launch_bounds__(256) mrp_layout_fill_kernel(const PlanCol *__restrict__ plan, const PlanHmm *__restrict__ ph, int64_t n_hmms,
                                                              const DevChunk *__restrict__ chunks, uint32_t xflags, LayoutOut o) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int64_t w = (int64_t) blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE;
    if (w >= n_hmms) return;
    const PlanHmm h = ph[w];
    const LayoutBase base = o.base[w];
    const bool ancestor = (h.flags & MRP_FLAG_INCLUDE_ANCESTOR_SUB_PROB) != 0;
    int64_t c_off = base.cell0, m_off = base.mcell0, t_fast = base.tile_fast0, t_gen = o.totals[2] + base.tile_gen0;
    for (int k0 = 0; k0 < h.n_cols; k0 += WAVE) {
        const int k = k0 + lane;
        const bool in = k < h.n_cols;
        const int64_t col = h.col0 + (in ? k : 0);
        const PlanCol c = plan[col];
        const uint16_t *dm = o.dims + 4 * col;
        const int C1 = in ? dm[0] : 0, C2 = in ? dm[1] : 0, Ma = in ? dm[2] : 0, Mb = in ? dm[3] : 0;
        const bool units = (xflags & MRP_XF_UNITS) != 0; /* array entries per column: see mrp_layout_count_kernel */
        const int C = units && ((c.a_part && c.d1 > 0) || (c.b_part && c.d2 > 0)) ? (C1 * C2) >> 1 : C1 * C2;
        const int M = units && !c.last && (c.out_a_paired || c.out_b_paired) ? (Ma * Mb) >> 1 : Ma * Mb;
        const int nt = (C + MRP_EMIT_TILE - 1) / MRP_EMIT_TILE;
        const bool fast = c.uniform_alleles != 0 && !ancestor;
        int tc, tm, tf, tg;
        const int xc = wave_excl_scan_shfl(C, lane, &tc), xm = wave_excl_scan_shfl(M, lane, &tm);
        const int xf = wave_excl_scan_shfl(in && fast ? nt : 0, lane, &tf), xg = wave_excl_scan_shfl(in && !fast ? nt : 0, lane, &tg);
        if (in) {
            DevCol dc;
            dc.cell_off = c_off + xc; dc.mcell_off = c.last ? 0 : m_off + xm; dc.slot_off = c.slot_off; dc.read_off = c.read_off;
            dc.n_cells = C; dc.n_merge = M; dc.site_start = c.site_start; dc.n_sites = c.n_sites; dc.depth = c.depth; dc.n_slots = c.n_slots;
            dc.chunk = c.chunk; dc.flags = h.flags | ((uint32_t) (c.uniform_alleles > 255 ? 0 : c.uniform_alleles) << 8); /* bits 8..15: allele count shared by the column's sites (0: they differ) */
            o.cols[col] = dc;
            SweepCol sc;
            sc.cell_off = dc.cell_off; sc.mcell_off = dc.mcell_off; sc.n_cells = C; sc.n_merge = M; sc.pad[0] = 0; sc.pad[1] = 0;
            o.scols[col] = sc;
            PlaneCol pc;
            pc.pool = chunks[c.chunk].pool; pc.read_off = c.read_off; pc.slot_off = c.slot_off; pc.depth = c.depth; pc.n_slots = c.n_slots;
            pc.need_planes = c.need_planes; pc.pad = 0;
            o.pcols[col] = pc;
            TileCol tcl;
            tcl.first = fast ? t_fast + xf : t_gen + xg; tcl.uniform_alleles = c.uniform_alleles; tcl.pad = 0;
            o.tilecols[col] = tcl;
            CrossCol x;
            x.a_part = c.a_part; x.b_part = c.b_part; x.a_np = c.a_np; x.b_np = c.b_np;
            x.x_cell_off = dc.cell_off;
            x.C1 = (uint16_t) C1; x.C2 = (uint16_t) C2; x.Ma = (uint16_t) Ma; x.Mb = (uint16_t) Mb;
            x.d1 = c.d1; x.d2 = c.d2;
            uint8_t fl = (uint8_t) xflags;
            x.out_a = c.last ? 0 : c.out_a; x.out_b = c.last ? 0 : c.out_b;
            if (!c.last) { if (c.out_a_paired) fl |= MRP_XF_OUT_A_PAIRED; if (c.out_b_paired) fl |= MRP_XF_OUT_B_PAIRED; }
            x.Pa = 0; x.Pb = 0; x.in_a = 0; x.in_b = 0;
            if (k > 0) { /* the connector that enters the column is the one that leaves the column before */
                const PlanCol q = plan[col - 1];
                const uint16_t *qm = o.dims + 4 * (col - 1);
                x.Pa = qm[2]; x.Pb = qm[3]; x.in_a = q.out_a; x.in_b = q.out_b;
                if (q.out_a_paired) fl |= MRP_XF_IN_A_PAIRED;
                if (q.out_b_paired) fl |= MRP_XF_IN_B_PAIRED;
            }
            x.flags = fl; x.pad = 0;
            o.ccols[col] = x;
        }
        c_off += tc; m_off += tm; t_fast += tf; t_gen += tg;
    }
}

hipError_t mrp_launch_layout(const PlanCol *plan_dev, const PlanHmm *hmms_dev, int64_t n_hmms, int64_t n_cols, const DevChunk *chunks_dev,
                             int32_t S, uint32_t xflags, LayoutOut out, hipStream_t stream) {
    if (n_hmms <= 0) return hipSuccess;
    (void) n_cols;
    const unsigned g = (unsigned) ((n_hmms + 3) / 4);
    hipLaunchKernelGGL(mrp_layout_count_kernel, dim3(g), dim3(256), 0, stream, plan_dev, hmms_dev, n_hmms, S, xflags, out);
    const unsigned tiles = (unsigned) ((n_hmms + LAYOUT_TILE - 1) / LAYOUT_TILE);
    hipLaunchKernelGGL(mrp_layout_tiles_kernel, dim3(tiles), dim3(LAYOUT_TILE), 0, stream, n_hmms, out, out.tile_sums);
    hipLaunchKernelGGL(mrp_layout_scan_kernel, dim3(tiles), dim3(LAYOUT_TILE), 0, stream, hmms_dev, n_hmms, out, out.tile_sums);
    hipLaunchKernelGGL(mrp_layout_fill_kernel, dim3(g), dim3(256), 0, stream, plan_dev, hmms_dev, n_hmms, chunks_dev, xflags, out);
    return hipGetLastError();
}

/* ------------------------------------------------------------------------------------------ */
/* prune                                                                                       */
/* ------------------------------------------------------------------------------------------ */
/* n kept of n_link candidates whose first g pass the posterior threshold: the loop of hmm.c:1073-1079 /
 * :1094-1100 ("while n > min && (n > max || last.posterior < threshold) drop last") in closed form */
static __device__ __forceinline__ int kept_count(int n_link, int g, int min_p, int max_p) {
    if (n_link <= min_p) return n_link;
    int n = g < max_p ? g : max_p;
    return n > min_p ? n : min_p;
}

/* posterior bin of a cell or merge cell: total - f - b, all three exact integers below 2^30 in magnitude (the int32
 * recursion kernel only takes hmms whose cost bound is below 2^30, and f + b counts every column at most once), so the
 * difference is formed in 32 bits -- in 64 bits the compiler widens every f and b it holds in registers */
static __device__ __forceinline__ int posterior_bin(int32_t f, int32_t b, int32_t total, int n_bins, int *errbits) {
    if (f == MRP_NEG_I32 || b == MRP_NEG_I32) return n_bins - 1; /* exp(-inf) = 0 */
    const int32_t s = total - f - b;
    if (s < 0) { *errbits |= MRP_ENGINE_ERR_POSTERIOR; return 0; }
    return s < n_bins - 1 ? (int) s : n_bins - 1;
}

/* Cross-lane primitives of the single-wave sections, on the DPP path where gfx950 has one (tools/ubench/dpp_prims.hip
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
/* the same with the shifted-in lanes read as zero by the DPP unit itself (bound_ctrl, row masks): an add per step */
static __device__ __forceinline__ int wave_incl_scan_bc(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);  /* row_shr:1 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); /* row_bcast:15 into rows 1 and 3 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false); /* row_bcast:31 into rows 2 and 3 */
    return v;
}
/* number of set bits of m below this lane */
static __device__ __forceinline__ int mbcnt64(uint64_t m) {
    return (int) __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u));
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
/* the same for 64 NR keys, NR per lane (index lane + 64 u in register u): ascending */
template <int NR, int K, int J>
static __device__ __forceinline__ void bitonic_step_n(uint32_t (&k)[NR], int lane) {
    if (J >= 64) { /* the partner sits in another register of the same lane; K > J >= 64: the direction depends on u alone */
        constexpr int dj = J / 64;
#pragma unroll
        for (int u = 0; u < NR; u++)
            if ((u & dj) == 0) {
                const bool asc = ((u * 64) & K) == 0;
                const uint32_t a = k[u], b = k[u | dj], lo = a < b ? a : b, hi = a < b ? b : a;
                k[u] = asc ? lo : hi;
                k[u | dj] = asc ? hi : lo;
            }
    } else {
#pragma unroll
        for (int u = 0; u < NR; u++) {
            const uint32_t pv = lane_xor<J>(k[u], lane);
            const bool lower = (lane & J) == 0, asc = ((lane + 64 * u) & K) == 0;
            const uint32_t mn = k[u] < pv ? k[u] : pv, mx = k[u] < pv ? pv : k[u];
            k[u] = (lower == asc) ? mn : mx;
        }
    }
}
template <int NR, int K>
static __device__ __forceinline__ void bitonic_merge_n(uint32_t (&k)[NR], int lane) {
    if (K >= 512) bitonic_step_n<NR, K, 256>(k, lane);
    if (K >= 256) bitonic_step_n<NR, K, 128>(k, lane);
    if (K >= 128) bitonic_step_n<NR, K, 64>(k, lane);
    if (K >= 64) bitonic_step_n<NR, K, 32>(k, lane);
    if (K >= 32) bitonic_step_n<NR, K, 16>(k, lane);
    if (K >= 16) bitonic_step_n<NR, K, 8>(k, lane);
    if (K >= 8) bitonic_step_n<NR, K, 4>(k, lane);
    if (K >= 4) bitonic_step_n<NR, K, 2>(k, lane);
    bitonic_step_n<NR, K, 1>(k, lane);
}
template <int NR>
static __device__ __forceinline__ void wave_bitonic_sort_n(uint32_t (&k)[NR], int lane) {
    bitonic_merge_n<NR, 2>(k, lane); bitonic_merge_n<NR, 4>(k, lane); bitonic_merge_n<NR, 8>(k, lane); bitonic_merge_n<NR, 16>(k, lane);
    bitonic_merge_n<NR, 32>(k, lane); bitonic_merge_n<NR, 64>(k, lane);
    if (NR >= 2) bitonic_merge_n<NR, 128>(k, lane);
    if (NR >= 4) bitonic_merge_n<NR, 256>(k, lane);
    if (NR >= 8) bitonic_merge_n<NR, 512>(k, lane);
}
/* orders this wave's LDS traffic for the compiler (the LDS itself executes one wave's instructions in issue order) */

/*
 * stRPHmm_prune for the cross products of a level, one workgroup per hmm.
 *
 * What the forward pass (stRPHmm_pruneForwards, hmm.c:1049-1109) needs of a column is small: the cells LINKED to a kept
 * merge cell of the previous merge column -- typically a hundred or two of the column's thousands (the cells of a cross
 * product column are all pairs (c1, c2) of the parents' cells; a kept merge cell (i, j) links exactly the pairs with
 * prev(c1) = i and prev(c2) = j).  So the candidates are ENUMERATED from the <= 128 kept merge cells and the parents'
 * transition arrays (<= 128 entries per side, inverted into per-merge-cell lists), never searched for among the cells:
 *
 *   wave 0        the sequential chain, without a barrier inside a column: kept merge cells -> candidates (64 per slot, as
 *                 many slots as the column needs; cell index by the closed form of the cross product order) -> posterior
 *                 bins (LDS gathers) -> cutoff bin by histogram -> selection (ties in the cutoff bin by list order = cell
 *                 index, ranked through a bitmap) -> the distinct next merge cells of the selection = the kept merge cells
 *                 of the next column, listed with their per-side indices.
 *   wave 1, 2     one and two columns behind: stable sort of the selection, kept lists to HBM, posteriors of the merge
 *                 cells, kept merge list (stRPHmm_pruneForwards' list building; nothing on the chain reads it).
 *   wave 3        inverts the parents' prev arrays of the NEXT column (loaded one column earlier), zeroes the histogram.
 *   waves 4..     two groups alternating over the columns: stream f and b of a whole column (the only bulk traffic: 8 B
 *                 per cell, requested two columns ahead) and leave its posterior bins in LDS as 16-bit values.
 *
 * Selection, tie order and list contents are those of the reference: sorting by (bin, cell index) is the stable sort of
 * the linked cells (in list order) by descending posterior (hmm.c:1043, :1071).
 */
struct PruneIn {
    const SweepCol *scols;
    const CrossCol *ccols;
    const int32_t *cell_f32, *cell_b32, *merge_f32, *merge_b32;
    const double *hmm_fb;
};

/* buffer descriptor over [p, p + bytes): both made wave-uniform for the compiler (cdna_hip_programming.md T8 / T20) */
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t prune_rsrc(const void *p, int bytes) {
    const uint64_t a = (uint64_t) p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t) a), hi = __builtin_amdgcn_readfirstlane((uint32_t) (a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void *) (((uint64_t) hi << 32) | lo), 0, bytes, 0x00020000);
}

/* profiling aid: build with -DPRUNE_EXP_CLOCK (both mrp_engine.cpp and this file) to sum, for the first hmm of a launch, the
 * shader cycles each role spends working and waiting at the column barrier (printed by mrp_engine.cpp under MRP_TIMING) */
#ifdef PRUNE_EXP_CLOCK
#define ROLE_CLK_INIT() uint64_t clk_work = 0, clk_wait = 0, clk_t = __builtin_amdgcn_s_memtime()
#define ROLE_BARRIER() do { const uint64_t t1_ = __builtin_amdgcn_s_memtime(); lds_barrier(); const uint64_t t2_ = __builtin_amdgcn_s_memtime(); \
                            clk_work += t1_ - clk_t; clk_wait += t2_ - t1_; clk_t = t2_; } while (0)
#define ROLE_CLK_DONE(slot) do { if (hi_ == 0 && lane == 0) { atomicAdd((unsigned long long *) (sc.err + 4) + 2 * (slot), (unsigned long long) clk_work); \
                                 atomicAdd((unsigned long long *) (sc.err + 4) + 2 * (slot) + 1, (unsigned long long) clk_wait); } } while (0)
#else
#define ROLE_CLK_INIT() do { } while (0)
#define ROLE_BARRIER() lds_barrier()
#define ROLE_CLK_DONE(slot) do { } while (0)
#endif

#ifdef PRUNE_EXP_CLOCK2 /* sections of the chain wave instead: kept merge cells, candidates, histogram + cutoff, selection, ties, next merge cells */
#define SEC_INIT() uint64_t sec_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; uint64_t sec_t = __builtin_amdgcn_s_memtime()
#define SEC_COUNT(i) do { sec_[i] += 1; } while (0) /* columns by path: 8 keep-all, 9 sorted, 10 histogram (one chunk), 11 histogram (several chunks) */
#define SEC(i) do { const uint64_t t_ = __builtin_amdgcn_s_memtime(); sec_[i] += t_ - sec_t; sec_t = t_; } while (0)
#define SEC_DONE() do { if (hi_ == 0 && lane == 0) for (int i_ = 0; i_ < 12; i_++) atomicAdd((unsigned long long *) (sc.err + 4) + i_, (unsigned long long) sec_[i_]); } while (0)
#else
#define SEC_INIT() do { } while (0)
#define SEC(i) do { } while (0)
#define SEC_COUNT(i) do { } while (0)
#define SEC_DONE() do { } while (0)
#endif

#define PRUNE_SP 128  /* slots of the selection buffers (S <= MRP_PRUNE_MAX_S) */
#define PRUNE_TAB 5   /* per side and buffer: cnt, start, list, nx, pv, 128 entries each */

/* T threads; the bin-streaming waves (all but the first four) form NGRP groups (2: alternating over the columns, a column's f and
 * b sit in a group's registers for two column times; 1: one group, requested one column ahead) and hold CPT loads of VEC cells
 * per lane and array. */
template <int T, int CPT, int NGRP, int VEC, bool PAIRS>
__global__ void __launch_bounds__(T, T <= 512 ? 4 : 1) mrp_prune_kernel(PruneIn d, const PruneHmm *__restrict__ hmms, int64_t n_hmms,
                                                      PruneParams p, PruneScratch sc) {
    constexpr int W = T / WAVE;
    constexpr int NBG = (W - 4) / NGRP;  /* waves per bin-streaming group; none (T = 256): columns of at most 64 CPT cells, whose
                                          * bins the table wave writes beside its tables -- four waves per workgroup, four
                                          * workgroups per CU where the register file allows two of eight waves */
    constexpr int LG = NBG > 0 ? NBG * WAVE : WAVE; /* lanes per group */
    static_assert((W == 4 || W >= 6) && ((W - 4) % NGRP) == 0 && (NGRP == 1 || NGRP == 2) && (VEC == 1 || VEC == 4), "role layout");
    extern __shared__ uint32_t lds[];
    const int S = p.S;
    const int nb = p.n_bins;
    const bool units = PAIRS && p.pairs == 2; /* the level's f, b and merge arrays hold one entry per unit (MRP_XF_UNITS) */
    const int nb_r = 1024; /* 16 bins per lane of the cutoff search: bin b lives at (b & 15) * 64 + (b >> 4) */
    /* posterior bins in LDS, two columns: one per cell; PAIRS: one per unit (cells 2u, 2u + 1 tie) */
    const int cap_c = PAIRS ? ((p.max_cells + 1) / 2 + 3) & ~3 : (p.max_cells + 3) & ~3;
    /* LDS layout (dwords) */
    uint32_t *sel = lds;                       /* [2][2][SP] selection of column k in buffer k & 1: key = bin << 14 | cell, np = next | prev << 16 */
    uint32_t *um = sel + 4 * PRUNE_SP;         /* [SP] posterior bin of the merge cell each selected cell leads to (selection order); wave 1 */
    uint32_t *s1 = um + PRUNE_SP;              /* [2][2][SP] stage 1 -> stage 2: next | prev and merge posterior bin per sorted kept cell */
    uint32_t *sh = s1 + 4 * PRUNE_SP;          /* [64] counters: n of the selection buffers at [32, 34), of the stage-1 buffers at [36, 38), of the kept merge lists at [40, 42) */
    uint32_t *hist = sh + 64;                  /* [2][nb_r] */
    uint32_t *bmp_c = hist + 2 * nb_r;         /* [512] cells of the cutoff bin, by cell index */
    uint32_t *bmp_m = bmp_c + 512;             /* [512] next merge cells seen */
    uint32_t *kml = bmp_m + 512;               /* [2][SP] kept merge cells leading into column k, buffer k & 1 */
    uint32_t *minfo = kml + 2 * PRUNE_SP;      /* [SP][2] per kept merge cell: what its candidates need (wave 0) */
    uint32_t *heads = minfo + 2 * PRUNE_SP;    /* [512] "a range starts here" marks of one chunk of candidates, zero between uses (wave 0) */
    uint32_t *pref = heads + 512;              /* [512] marks before each word of the cutoff bin's bitmap (wave 0) */
    uint32_t *tab = pref + 512;                /* [2 buffers][2 sides][PRUNE_TAB][128] */
    uint32_t *htab_key = tab + 4 * PRUNE_TAB * 128; /* [256] merge cell -> first kept cell that uses it (open addressing); wave 2 */
    uint32_t *htab_val = htab_key + 256;       /* [256] */
    uint32_t *flagw = htab_val + 256;          /* [max_merge bits] kept flag per merge cell (backward pass) */
    const int n_flagw = ((p.max_merge + 127) >> 7) << 2; /* words, a multiple of four */
    uint16_t *bins = reinterpret_cast<uint16_t *>(flagw + n_flagw); /* [2][cap_c] posterior bin per cell */
    auto flag_get = [&](uint32_t i) -> bool { return (flagw[i >> 5] >> (i & 31u)) & 1u; };
    auto flag_set = [&](uint32_t i) { atomicOr(&flagw[i >> 5], 1u << (i & 31u)); };
    auto flag_clr = [&](uint32_t i) { atomicAnd(&flagw[i >> 5], ~(1u << (i & 31u))); };

    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    /* The four role waves of a workgroup sit on the CU's four SIMDs (wave i on SIMD i mod 4).  Workgroups that share a CU
     * rotate the roles: otherwise every chain wave -- the one wave of a workgroup that is busy all the time -- would issue
     * from SIMD 0 and the workgroups would take turns on it.  MRP_PRUNE_NO_ROT (development) switches that off. */
    const int hw_wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
#ifdef MRP_PRUNE_NO_ROT
    const int wave = hw_wave;
#else
    const int wave = hw_wave < 4 ? ((hw_wave + (int) (blockIdx.x & 3u)) & 3) : hw_wave;
#endif
    const uint64_t lt_mask = (1ull << lane) - 1ull;

    for (int64_t hi_ = blockIdx.x; hi_ < n_hmms; hi_ += gridDim.x) {
        int errbits = 0;
        const PruneHmm h = k_load(hmms + hi_);
        const int K = h.n_cols;
        const int32_t total = (int32_t) d.hmm_fb[2 * h.hmm_index]; /* max mode: the same integer for every column */
        for (int i = tid; i < 2 * nb_r; i += T) hist[i] = 0;
        for (int i = tid; i < 1024; i += T) bmp_c[i] = 0; /* both bitmaps */
        for (int i = tid; i < 512; i += T) heads[i] = 0;
        for (int i = tid; i < n_flagw; i += T) flagw[i] = 0;
        if (tid < 64) sh[tid] = 0u;
        __syncthreads();

        /* The sorted lists of a finished selection, in two stages one column apart.  Nothing later in the forward pass
         * reads them: the next column's kept merge cells were already listed by wave 0 from the unsorted selection.
         * Stage 1 (wave 1, column kk from selection buffer kk & 1): stable sort of the kept cells, kept lists to HBM, the
         * posterior bins of the merge cells they lead to; leaves next | prev and that bin per sorted kept cell in LDS. */
        auto lists_stage1 = [&](int kk, int64_t mcell_off) {
            if constexpr (PAIRS) {
                /* Complement pairs (see the chain wave): the selection holds UNITS, at most 64; a unit's two cells tie and are
                 * neighbours in list order, so the sorted units, each expanded to (cell 2u, cell 2u + 1), are the sorted cells */
                const int b = kk & 1;
                const uint2 *selb = reinterpret_cast<const uint2 *>(sel) + b * 64; /* (bin << 14 | unit, next | prev << 16) */
                uint32_t *snp = s1 + b * 2 * PRUNE_SP, *sbin = snp + PRUNE_SP;
                const int n = (int) sh[32 + b];
                const uint32_t fl = sh[44 + b]; /* bit 0: the column's cells come in pairs, 1: the merge cells after it, 2: those before it */
                const uint32_t o_pm = (fl >> 1) & 1u, i_pm = (fl >> 2) & 1u;
                const int64_t lcol = h.col0 + kk;
                const bool has_merge = kk + 1 < K;
                uint32_t key[1];
                int32_t pre_mf = 0, pre_mb = 0;
                const uint2 mine = selb[lane];
                key[0] = lane < n ? (mine.x << 7) | (uint32_t) lane : 0xFFFFFFFFu; /* bin (10) | unit (14) | slot (7) */
                if (has_merge && lane < n) {
                    const uint32_t m = mine.y & 0xFFFFu;
                    pre_mf = d.merge_f32[mcell_off + m];
                    pre_mb = d.merge_b32[mcell_off + m];
                }
                wave_bitonic_sort_n<1>(key, lane);
                if (has_merge) {
                    if (lane < n) um[lane] = (uint32_t) posterior_bin(pre_mf, pre_mb, total, nb, &errbits);
                    wave_lds_fence();
                }
                if (lane < n) {
                    const uint32_t src = key[0] & 0x7Fu, u = (key[0] >> 7) & 0x3FFFu, np_ = selb[src].y;
                    if (fl & 1u) {
                        *reinterpret_cast<uint32_t *>(sc.kept + lcol * S + 2 * lane) = (2u * u) | ((2u * u + 1u) << 16);
                        *reinterpret_cast<uint2 *>(sc.kept_np + lcol * S + 2 * lane) = make_uint2(np_, np_ ^ o_pm ^ (i_pm << 16));
                    } else { /* a column of one cell */
                        sc.kept[lcol * S + lane] = (uint16_t) u;
                        sc.kept_np[lcol * S + lane] = np_;
                    }
                    snp[lane] = np_;
                    if (has_merge) sbin[lane] = um[src];
                }
                if (lane == 0) { sc.n_kept[lcol] = (fl & 1u) ? 2 * n : n; sh[36 + b] = (uint32_t) n; sh[46 + b] = fl; }
                return;
            }
            const int b = kk & 1;
            const uint32_t *skey = sel + b * 2 * PRUNE_SP, *snp_in = skey + PRUNE_SP;
            uint32_t *snp = s1 + b * 2 * PRUNE_SP, *sbin = snp + PRUNE_SP;
            const int n = (int) sh[32 + b];
            const int64_t lcol = h.col0 + kk;
            const bool has_merge = kk + 1 < K;
            uint32_t key[2];
            int32_t pre_mf[2] = {0, 0}, pre_mb[2] = {0, 0};
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int i = lane + u * WAVE;
                /* stable descending sort of the kept cells (stList_sort :1071): smaller bin = larger posterior, then list
                 * order = cell index; the slot rides along in the low bits: bin (10) | cell (14) | slot (7) */
                key[u] = i < n ? (skey[i] << 7) | (uint32_t) i : 0xFFFFFFFFu;
                /* the posteriors of the merge cells the selected cells lead to are requested now, for the selection in
                 * its unsorted order, and consumed after the sort */
                if (has_merge && i < n) {
                    const uint32_t m = snp_in[i] & 0xFFFFu;
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
                wave_lds_fence();
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int i = lane + u * WAVE;
                if (i < n) {
                    const uint32_t src = key[u] & 0x7Fu, cell = (key[u] >> 7) & 0x3FFFu, np_ = snp_in[src];
                    sc.kept[lcol * S + i] = (uint16_t) cell;
                    sc.kept_np[lcol * S + i] = np_;
                    snp[i] = np_;
                    if (has_merge) sbin[i] = um[src];
                }
            }
            if (lane == 0) { sc.n_kept[lcol] = n; sh[36 + b] = (uint32_t) n; }
        };
        /* PAIRS: stage 1 in two halves a column apart -- the posteriors of the merge cells are a gather from HBM (two microseconds
         * under load, as long as a whole column of the chain on pairs): asked for when the selection of column kk is read (step
         * kk + 1), used a step later, with the selection in registers meanwhile (the chain reuses its buffer). */
        uint32_t p1_key = 0xFFFFFFFFu, p1_np = 0u, p1_fl = 0u;
        int32_t p1_mf = 0, p1_mb = 0;
        int p1_n = 0;
        auto stage1_ask = [&](int kk, int64_t mcell_off) {
            const int b = kk & 1;
            const uint2 *selb = reinterpret_cast<const uint2 *>(sel) + b * 64; /* (bin << 14 | unit, next | prev << 16) */
            p1_n = (int) sh[32 + b];
            p1_fl = sh[44 + b]; /* bit 0: the column's cells come in pairs, 1: the merge cells after it, 2: those before it */
            const uint2 mine = selb[lane];
            p1_key = lane < p1_n ? (mine.x << 7) | (uint32_t) lane : 0xFFFFFFFFu; /* bin (10) | unit (14) | slot (7) */
            p1_np = mine.y;
            p1_mf = 0; p1_mb = 0;
            if (kk + 1 < K && lane < p1_n) {
                const uint32_t m = (mine.y & 0xFFFFu) >> (units ? (p1_fl >> 1) & 1u : 0u); /* (units: the merge arrays hold merge units) */
                p1_mf = d.merge_f32[mcell_off + m];
                p1_mb = d.merge_b32[mcell_off + m];
            }
        };
        auto stage1_finish = [&](int kk) {
            const int b = kk & 1;
            uint32_t *snp = s1 + b * 2 * PRUNE_SP, *sbin = snp + PRUNE_SP;
            const int n = p1_n;
            const uint32_t fl = p1_fl, o_pm = (fl >> 1) & 1u, i_pm = (fl >> 2) & 1u;
            const int64_t lcol = h.col0 + kk;
            const bool has_merge = kk + 1 < K;
            uint32_t key[1] = {p1_key};
            wave_bitonic_sort_n<1>(key, lane);
            const uint32_t mbin = has_merge && lane < n ? (uint32_t) posterior_bin(p1_mf, p1_mb, total, nb, &errbits) : 0u;
            /* what rides along with a unit comes from the lane that held it before the sort */
            const uint32_t src = key[0] & 0x3Fu, u = (key[0] >> 7) & 0x3FFFu;
            const uint32_t np_ = (uint32_t) __builtin_amdgcn_ds_bpermute((int) (src << 2), (int) p1_np);
            const uint32_t sb_ = (uint32_t) __builtin_amdgcn_ds_bpermute((int) (src << 2), (int) mbin);
            if (lane < n) {
                if (fl & 1u) {
                    *reinterpret_cast<uint32_t *>(sc.kept + lcol * S + 2 * lane) = (2u * u) | ((2u * u + 1u) << 16);
                    *reinterpret_cast<uint2 *>(sc.kept_np + lcol * S + 2 * lane) = make_uint2(np_, np_ ^ o_pm ^ (i_pm << 16));
                } else { /* a column of one cell */
                    sc.kept[lcol * S + lane] = (uint16_t) u;
                    sc.kept_np[lcol * S + lane] = np_;
                }
                snp[lane] = np_;
                sbin[lane] = sb_;
            }
            if (lane == 0) { sc.n_kept[lcol] = (fl & 1u) ? 2 * n : n; sh[36 + b] = (uint32_t) n; sh[46 + b] = fl; }
        };
        /* Stage 2 (wave 2, one column later): distinct next merge cells in order of first use, stable sort by posterior,
         * kept merge list to HBM. */
        auto lists_stage2 = [&](int kk) {
            if constexpr (PAIRS) {
                const int b = kk & 1;
                const uint32_t *snp = s1 + b * 2 * PRUNE_SP, *sbin = snp + PRUNE_SP;
                const int n = (int) sh[36 + b];
                const uint32_t o_pm = (sh[46 + b] >> 1) & 1u;
                const int64_t lcol = h.col0 + kk;
                int mn_out = 0;
                if (kk + 1 < K) {
                    /* distinct next merge UNITS in order of first use; the two merge cells of a unit enter the list in the order the
                     * unit's first user reaches them: its even cell first (bit q of that cell's merge cell index), then its twin */
                    for (int i = lane; i < 256; i += WAVE) { htab_key[i] = 0xFFFFFFFFu; htab_val[i] = 0xFFFFFFFFu; }
                    wave_lds_fence();
                    uint32_t my_mu = 0u, my_q = 0u;
                    int slot = 0;
                    if (lane < n) {
                        const uint32_t mfull = snp[lane] & 0xFFFFu;
                        my_mu = mfull >> o_pm; my_q = mfull & o_pm;
                        int q = (int) ((my_mu * 2654435761u) >> 24);
                        for (;;) {
                            const uint32_t prev = atomicCAS(&htab_key[q], 0xFFFFFFFFu, my_mu);
                            if (prev == 0xFFFFFFFFu || prev == my_mu) break;
                            q = (q + 1) & 255;
                        }
                        slot = q;
                        atomicMin(&htab_val[q], (uint32_t) lane);
                    }
                    wave_lds_fence();
                    const bool first = lane < n && htab_val[slot] == (uint32_t) lane;
                    const uint64_t f0 = __ballot(first);
                    const int mnl = __popcll(f0);
                    uint32_t mkey[1] = {0xFFFFFFFFu};
                    bool pass_thr = false;
                    if (first) {
                        const int pos = (int) lanemask_lt_count(f0, lane);
                        const int bin = (int) sbin[lane];
                        pass_thr = bin <= p.thr_bin;
                        mkey[0] = ((uint32_t) bin << 20) | ((uint32_t) pos << 14) | (my_mu << 1) | my_q;
                    }
                    const int gm = __popcll(__ballot(pass_thr));
                    const int wm = o_pm ? 2 : 1;
                    const int mn = kept_count(mnl * wm, gm * wm, p.min_p, p.max_p);
                    if (mn != mnl * wm || (p.pad && hi_ == 0 && kk == 0)) errbits |= MRP_ENGINE_ERR_MERGE;
                    wave_bitonic_sort_n<1>(mkey, lane);
                    if (lane < mnl) {
                        const uint32_t mu = (mkey[0] >> 1) & 0x1FFFu, q = mkey[0] & 1u;
                        if (o_pm) *reinterpret_cast<uint32_t *>(sc.keptm + lcol * S + 2 * lane) = (2u * mu + q) | ((2u * mu + (q ^ 1u)) << 16);
                        else sc.keptm[lcol * S + lane] = (uint16_t) mu;
                    }
                    mn_out = mnl * wm;
                }
                if (lane == 0) sc.n_keptm[lcol] = mn_out;
                return;
            }
            const int b = kk & 1;
            const uint32_t *snp = s1 + b * 2 * PRUNE_SP, *sbin = snp + PRUNE_SP;
            const int n = (int) sh[36 + b];
            const int64_t lcol = h.col0 + kk;
            int mn = 0;
            if (kk + 1 < K) {
                /* getLinkedMergeCells :989-1004: distinct next merge cells in order of first use (sorted order of the
                 * kept cells): a 256-slot open-addressing table, merge cell -> smallest sorted index that uses it */
                for (int i = lane; i < 256; i += WAVE) { htab_key[i] = 0xFFFFFFFFu; htab_val[i] = 0xFFFFFFFFu; }
                wave_lds_fence();
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
                wave_lds_fence();
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
                /* Wave 0 has kept EVERY distinct next merge cell.  That is what :1090-1100 keeps: a merge cell's
                 * posterior is at least that of any cell leading to it (max mode, exact integers), so whenever more
                 * than min_p cells were kept they all pass the threshold, and so do their merge cells.  Checked, not
                 * assumed: a violation discards the hmm (the host redoes its chunk on the hashing path). */
                if (mn != mnl || (p.pad && hi_ == 0 && kk == 0)) errbits |= MRP_ENGINE_ERR_MERGE; /* p.pad: fault injection of the tests */
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

        /* ---- stRPHmm_pruneForwards hmm.c:1049-1109 ----
         * Every role runs its own loop over the columns (the registers of one role are not live in the others); all of them
         * pass the same barriers: one after the prologue, one per column, one before the last merge list. */
        if (wave == 0) {
            /* the chain wave goes first whenever it can issue: alone on the device that changes nothing (its SIMD's helper
             * waves are parked), beside the kernels of other batches on the same CU it is worth ~2 % of a call */
            __builtin_amdgcn_s_setprio(3);
            if (lane == 0) {
                kml[0] = 0u;   /* column 0: every cell is "linked" (one virtual merge cell in front of it) */
                sh[40] = 1u;
            }
            lds_barrier();
            ROLE_CLK_INIT();
            SEC_INIT();
            if constexpr (PAIRS) {
            /* ---- the chain on complement PAIRS (includeInvertedPartitions, even column limits) ----
             * Every cell of a cross product column has its complement next to it (cells 2u, 2u + 1 = "unit" u; the order rule of
             * cross_cell / pair_index), with the same cost, f, b and posterior, and the twin's merge cells are the twins of its
             * merge cells.  A kept merge cell is kept with its twin, linked cells are linked with their twins, ties between twins
             * fall in list order (2u before 2u + 1), and the limits are even, so the forward pass never separates a pair: the
             * chain carries UNITS -- half the kept merge cells (one per lane), half the candidates, half the selection -- and waves
             * 1 and 2 write both cells of every kept unit.  A unit is named by ANY of its two members where that is cheaper:
             *   unit(x, y)   = (x >> 1) * Y + ((y ^ (x & xm)) >> sb)      x, y: indices on side A / B, Y: side B's count
             *   parity(x, y) = (x & pa) | (y & sb)                         which member of its unit the pair (x, y) is
             * with xm = both sides paired, sb = only side B paired, pa = side A paired (all wave-uniform).  A column or merge
             * column with neither side paired has one self-complementary entry: a unit of one cell.  The cells linked to ONE
             * member of a kept merge unit are one member of every linked unit, so the enumeration is that of the general
             * kernel over half the entries; only behind a merge column of one cell both members turn up, and the odd ones are
             * dropped. */
            /* One slot of 64 candidates at a time, start to finish (owner, parent cells, unit, posterior bin, next merge unit):
             * the slots of a column are independent, but a column links 65 units on average -- one or two slots -- and code that
             * keeps eight slots in flight at once spends more instructions moving its register tuples through the "has this slot
             * work" branches than on the work.  Records of the kept merge units (16 bytes, with the reciprocal for the division
             * by the side-B count), the selection (key and transitions as one 8-byte entry) and the marks sit in LDS. */
            uint4 *rec = reinterpret_cast<uint4 *>(minfo);       /* [64] per kept merge unit of the column */
            for (int k = 0; k < K; k++) {
                SEC(7);
                const int b = k & 1;
                const uint4 cd = *reinterpret_cast<const uint4 *>(sh + 48 + 4 * b);
                const int nkm = (int) sh[40 + b];
                const uint32_t ent_ = kml[b * PRUNE_SP + lane];
                const uint32_t cd0 = (uint32_t) __builtin_amdgcn_readfirstlane((int) cd.x), cd1 = (uint32_t) __builtin_amdgcn_readfirstlane((int) cd.y);
                const uint32_t cd2 = (uint32_t) __builtin_amdgcn_readfirstlane((int) cd.z); /* the table wave's bits */
                uint2 *selb = reinterpret_cast<uint2 *>(sel) + b * 64; /* selection of column k: (bin << 14 | unit, next | prev << 16 of the unit's even cell) */
                const uint32_t *tA = tab + b * 2 * PRUNE_TAB * 128, *tB = tA + PRUNE_TAB * 128;
                const uint32_t *csA = tA + 128, *listA = tA + 256, *nxA = tA + 384;
                const uint32_t *csB = tB + 128, *listB = tB + 256, *nxB = tB + 384;
                const uint16_t *bin_k = bins + b * cap_c;
                uint32_t *hk = hist + b * nb_r;
                uint32_t *kmn = kml + (b ^ 1) * PRUNE_SP; /* the kept merge units leading into column k + 1 */
                const uint32_t cflags = cd1 & 0xFFu;
                const uint32_t C2 = (cd0 & 0xFFFFu) > 128u ? 128u : (cd0 & 0xFFFFu), Mb = cd0 >> 16, Pb = cd1 >> 16;
                const bool a_cp = (cd1 & 0x100u) != 0, b_cp = (cd1 & 0x200u) != 0;
                const bool in_ap = (cflags & MRP_XF_IN_A_PAIRED) != 0, in_bp = (cflags & MRP_XF_IN_B_PAIRED) != 0;
                const bool has_next = k + 1 < K;
                const uint32_t c_xm = cd2 & 1u, c_sb = (cd2 >> 1) & 1u, c_pa = (cd2 >> 2) & 1u;
                const uint32_t o_xm = (cd2 >> 3) & 1u, o_sb = (cd2 >> 4) & 1u, o_pa = (cd2 >> 5) & 1u;
                const uint32_t o_pm = (cd2 >> 6) & 1u, i_pm = (cd2 >> 7) & 1u;
                const uint32_t i_pa = (cd2 >> 8) & 1u, i_sb = (cd2 >> 9) & 1u;
                const int w_sh = (int) ((cd2 >> 10) & 1u), w_c = 1 << w_sh;  /* cells per unit of this column: 1 << w_sh */
                const bool filt = ((cd2 >> 11) & 1u) != 0;  /* behind a merge column of one cell: both members are enumerated */
                /* this lane's kept merge unit: the parents' cells one of its members links, as list ranges.  An entry of the
                 * list is unit | i << 14 | j << 21: the unit and the indices of that member on either side */
                const bool has = lane < nkm;
                const uint32_t ent = has ? ent_ : 0u;
                const uint32_t e_i = (ent >> 14) & 127u, e_j = (ent >> 21) & 127u;
                const uint32_t e_csa = csA[e_i], e_csb = csB[e_j]; /* first entry | size << 8 of the group in the inverted lists */
                const uint32_t e_nb = has ? e_csb >> 8 : 0u;
                const int tot = has ? (int) ((e_csa >> 8) * e_nb) : 0;
                if (has && tot == 0) errbits |= MRP_ENGINE_ERR_RANGE; /* (a kept merge cell links at least one cell: the owners below count on it) */
                const int incl = wave_incl_scan_bc(tot);
                const int L = __builtin_amdgcn_readlane(incl, WAVE - 1); /* pairs of parent cells enumerated */
                const int off = incl - tot;
                {
                    const uint32_t e_par = (e_i & i_pa) | (e_j & i_sb);     /* which member of its unit the entry names */
                    const float rn = __builtin_amdgcn_rcpf((float) (e_nb ? e_nb : 1u));
                    if (has) rec[lane] = make_uint4((ent & 0x3FFFu) | (e_nb << 14) | (e_par << 22), (uint32_t) off | ((e_csa & 0xFFu) << 14) | ((e_csb & 0xFFu) << 21),
                                                    __float_as_uint(rn), 0u);
                }
                const int Lu = filt ? L >> 1 : L;  /* linked units */
                const int Lc = Lu << w_sh;         /* linked cells */
                SEC(0);
                const bool thr_all = p.thr_bin >= nb - 1;
                const bool keep_all = Lc <= p.min_p || (thr_all && Lc <= p.max_p); /* the loop of :1073-1079 drops nothing */
                auto kept_units = [&](int g_units) -> int { return kept_count(Lc, g_units << w_sh, p.min_p, p.max_p) >> w_sh; };
                /* candidates q0 .. q0 + 63: key = bin << 14 | unit (0xFFFFFFFF: none), the parent cells the enumeration met, and the
                 * merge cell the unit's EVEN cell comes from */
                auto slot = [&](int q0, uint32_t &c1, uint32_t &c2, uint32_t &prv_even) -> uint32_t {
                    /* marks: index + 1 of the kept merge unit whose range starts (or, at position 0, continues) here.  The
                     * ranges follow each other without gaps, so the owner of a position is the first owner plus the marks
                     * up to it: no prefix maximum */
                    if (tot > 0 && off + tot > q0 && off < q0 + WAVE) heads[off > q0 ? off - q0 : 0] = (uint32_t) lane + 1u;
                    wave_lds_fence();
                    const uint32_t own = heads[lane];
                    heads[lane] = 0u; /* left clean for the next use */
                    const uint64_t marks = __ballot(own != 0u);
                    const int first_owner = __builtin_amdgcn_readfirstlane((int) own) - 1;
                    const int owner = first_owner + mbcnt64(marks) + (own != 0u ? 1 : 0) - 1;
                    const bool in = q0 + lane < L;
                    const uint4 r = rec[in ? owner & 63 : 0];
                    const uint32_t nbq = (r.x >> 14) & 0xFFu, sa = (r.y >> 14) & 0x7Fu, sb = (r.y >> 21) & 0x7Fu;
                    const uint32_t t = (uint32_t) (q0 + lane) - (r.y & 0x3FFFu);
                    /* t = x * nb + y: (t + 0.5) / nb is at least 0.5 / nb away from an integer and the float product is off by less */
                    const uint32_t x = (uint32_t) (((float) t + 0.5f) * __uint_as_float(r.z));
                    const uint32_t y = t - x * nbq;
                    c1 = listA[(sa + x) & 127u];
                    c2 = listB[(sb + y) & 127u];
                    const uint32_t par_c = (c1 & c_pa) | (c2 & c_sb);
                    const uint32_t u = in ? (c1 >> 1) * C2 + ((c2 ^ (c1 & c_xm)) >> c_sb) : 0u;
                    prv_even = ((r.x & 0x3FFFu) << 1) + ((((r.x >> 22) & 1u) ^ par_c) & i_pm);
                    const uint32_t bin_ = bin_k[u];
                    return (!in || (filt && par_c != 0u)) ? 0xFFFFFFFFu : ((bin_ << 14) | u);
                };
                /* a selected candidate: its slot of the selection, and -- the first time its next merge unit is seen -- that unit as
                 * a kept merge unit of the next column.  Called in wave-uniform control flow, `take` per lane. */
                int cm = 0;
                auto emit = [&](bool take, int pos, uint32_t key_, uint32_t c1, uint32_t c2, uint32_t prv_even) {
                    const uint32_t ii = nxA[c1 & 127u], jj = nxB[c2 & 127u];
                    const uint32_t par_c = (c1 & c_pa) | (c2 & c_sb);
                    const uint32_t mu2 = has_next ? (ii >> 1) * Mb + ((jj ^ (ii & o_xm)) >> o_sb) : 0u;
                    const uint32_t par_m = (ii & o_pa) | (jj & o_sb);
                    const uint32_t nxt_even = (mu2 << 1) + ((par_m ^ par_c) & o_pm); /* the merge cell the unit's EVEN cell leads to */
                    if (take) selb[pos & 63] = make_uint2(key_, nxt_even | (prv_even << 16));
                    if (has_next) {
                        bool first = false;
                        if (take) {
                            const uint32_t bit = 1u << (mu2 & 31u);
                            first = (atomicOr(&bmp_m[(mu2 >> 5) & 511u], bit) & bit) == 0u;
                        }
                        const uint64_t fm = __ballot(first);
                        if (first) kmn[(cm + mbcnt64(fm)) & 127] = mu2 | (ii << 14) | (jj << 21);
                        cm += __popcll(fm);
                    }
                };
                int n = 0;
                /* Columns of up to 64 NS candidates (NS = 1, 2, 4: nine columns in ten): the NS slots side by side in straight-line
                 * code -- one fence for all their marks, their LDS round trips overlapped -- then either every candidate is kept
                 * (emit in enumeration order; the list stage sorts) or the kept units are the first n of the sorted keys. */
                auto small_col = [&](auto ns_tag) {
                    constexpr int NS = decltype(ns_tag)::value;
                    if (tot > 0) { /* marks: where this unit's range starts, and the slots it continues into */
                        if (off < NS * WAVE) heads[off] = (uint32_t) lane + 1u;
#pragma unroll
                        for (int j = 1; j < NS; j++)
                            if (off < j * WAVE && off + tot > j * WAVE) heads[j * WAVE] = (uint32_t) lane + 1u;
                    }
                    wave_lds_fence();
                    uint32_t own[NS], key[NS], c1[NS], c2[NS], pe[NS];
#pragma unroll
                    for (int j = 0; j < NS; j++) { own[j] = heads[j * WAVE + lane]; heads[j * WAVE + lane] = 0u; }
                    uint4 r[NS];
                    bool in[NS];
#pragma unroll
                    for (int j = 0; j < NS; j++) {
                        const uint64_t marks = __ballot(own[j] != 0u);
                        const int first_owner = __builtin_amdgcn_readfirstlane((int) own[j]) - 1;
                        const int owner = first_owner + mbcnt64(marks) + (own[j] != 0u ? 1 : 0) - 1;
                        in[j] = j * WAVE + lane < L;
                        r[j] = rec[in[j] ? owner & 63 : 0];
                    }
#pragma unroll
                    for (int j = 0; j < NS; j++) {
                        const uint32_t nbq = (r[j].x >> 14) & 0xFFu, sa = (r[j].y >> 14) & 0x7Fu, sb = (r[j].y >> 21) & 0x7Fu;
                        const uint32_t t = (uint32_t) (j * WAVE + lane) - (r[j].y & 0x3FFFu);
                        const uint32_t x = (uint32_t) (((float) t + 0.5f) * __uint_as_float(r[j].z)); /* exact: see slot() */
                        const uint32_t y = t - x * nbq;
                        c1[j] = listA[(sa + x) & 127u];
                        c2[j] = listB[(sb + y) & 127u];
                    }
                    uint32_t ii[NS], jj[NS];
#pragma unroll
                    for (int j = 0; j < NS; j++) {
                        const uint32_t par_c = (c1[j] & c_pa) | (c2[j] & c_sb);
                        const uint32_t u = in[j] ? (c1[j] >> 1) * C2 + ((c2[j] ^ (c1[j] & c_xm)) >> c_sb) : 0u;
                        pe[j] = ((r[j].x & 0x3FFFu) << 1) + ((((r[j].x >> 22) & 1u) ^ par_c) & i_pm);
                        const uint32_t bin_ = bin_k[u];
                        if (keep_all) { ii[j] = nxA[c1[j] & 127u]; jj[j] = nxB[c2[j] & 127u]; } /* (wave-uniform: asked for together with the bins) */
                        key[j] = (!in[j] || (filt && par_c != 0u)) ? 0xFFFFFFFFu : ((bin_ << 14) | u);
                    }
                    if (keep_all) {
                        SEC_COUNT(8);
                        uint32_t mu2[NS], ent2[NS], old[NS];
#pragma unroll
                        for (int j = 0; j < NS; j++) {
                            const bool take = key[j] != 0xFFFFFFFFu;
                            const uint32_t par_c = (c1[j] & c_pa) | (c2[j] & c_sb);
                            mu2[j] = has_next ? (ii[j] >> 1) * Mb + ((jj[j] ^ (ii[j] & o_xm)) >> o_sb) : 0u;
                            const uint32_t par_m = (ii[j] & o_pa) | (jj[j] & o_sb);
                            const uint32_t nxt_even = (mu2[j] << 1) + ((par_m ^ par_c) & o_pm);
                            ent2[j] = mu2[j] | (ii[j] << 14) | (jj[j] << 21);
                            const uint64_t am = __ballot(take);
                            if (take) selb[(n + mbcnt64(am)) & 63] = make_uint2(key[j], nxt_even | (pe[j] << 16));
                            n += __popcll(am);
                            old[j] = 0xFFFFFFFFu;
                            if (has_next && take) old[j] = atomicOr(&bmp_m[(mu2[j] >> 5) & 511u], 1u << (mu2[j] & 31u));
                        }
                        if (has_next) {
#pragma unroll
                            for (int j = 0; j < NS; j++) {
                                const bool first = ((old[j] >> (mu2[j] & 31u)) & 1u) == 0u;
                                const uint64_t fm = __ballot(first);
                                if (first) kmn[(cm + mbcnt64(fm)) & 127] = ent2[j];
                                cm += __popcll(fm);
                            }
                        }
                        SEC(1);
                    } else {
                        SEC_COUNT(9);
                        int g = Lu;
                        if (!thr_all) {
                            g = 0;
#pragma unroll
                            for (int j = 0; j < NS; j++) g += __popcll(__ballot(key[j] != 0xFFFFFFFFu && (int) (key[j] >> 14) <= p.thr_bin));
                        }
                        n = kept_units(g);
                        wave_bitonic_sort_n<NS>(key, lane);
                        const uint32_t *pvA = tA + 512, *pvB = tB + 512;
                        /* n <= 64 units: the kept ones are in the first register.  The even cell of unit u and where it comes from: */
                        const bool take = lane < n;
                        const uint32_t u = take ? key[0] & 0x3FFFu : 0u;
                        uint32_t d1, d2;
                        if (a_cp) {
                            const uint32_t q = (uint32_t) (((float) u + 0.5f) * __builtin_amdgcn_rcpf((float) (C2 ? C2 : 1u)));
                            d1 = 2u * q; d2 = u - q * C2;
                        } else if (b_cp) { d1 = 0u; d2 = 2u * u; }
                        else { d1 = 0u; d2 = 0u; }
                        d1 &= 127u; d2 &= 127u;
                        const uint32_t prv = k > 0 ? pair_index(pvA[d1], pvB[d2], Pb, true, in_ap, in_bp) : 0u;
                        emit(take, lane, key[0], d1, d2, prv);
                        SEC(4);
                    }
                };
                if (L <= WAVE) small_col(std::integral_constant<int, 1>());
                else if (L <= 2 * WAVE) small_col(std::integral_constant<int, 2>());
                else if (L <= 4 * WAVE) small_col(std::integral_constant<int, 4>());
                else if (keep_all) {
                    SEC_COUNT(8);
#pragma unroll 1
                    for (int q0 = 0; q0 < L; q0 += WAVE) {
                        uint32_t c1, c2, pe;
                        const uint32_t key_ = slot(q0, c1, c2, pe);
                        const uint64_t am = __ballot(key_ != 0xFFFFFFFFu);
                        emit(key_ != 0xFFFFFFFFu, n + mbcnt64(am), key_, c1, c2, pe);
                        n += __popcll(am);
                    }
                    SEC(1);
                } else {
                    /* pass 1: histogram of the posterior bins */
                    if (lane == 0) sh[56 + b] = 1u; /* the table wave wipes this histogram before its next use */
                    SEC_COUNT(L > 512 ? 11 : 10);
#pragma unroll 1
                    for (int q0 = 0; q0 < L; q0 += WAVE) {
                        uint32_t c1, c2, pe;
                        const uint32_t key_ = slot(q0, c1, c2, pe);
                        if (key_ != 0xFFFFFFFFu) {
                            const int bin_ = (int) (key_ >> 14);
                            atomicAdd(&hk[(bin_ & 15) * WAVE + (bin_ >> 4)], 1u);
                        }
                    }
                    wave_lds_fence();
                    SEC(2);
                    /* cutoff bin and quota: lane l owns the 16 consecutive bins [16 l, 16 l + 16) */
                    int v[16];
                    int tot_l = 0, pass = 0;
                    const int thr_rel = p.thr_bin - lane * 16; /* bins q <= thr_rel of this lane pass the threshold */
#pragma unroll
                    for (int q = 0; q < 16; q++) {
                        v[q] = (int) hk[q * WAVE + lane];
                        tot_l += v[q];
                        if (q <= thr_rel) pass += v[q];
                    }
                    const int incl_h = wave_incl_scan(tot_l, lane);
                    const int g = thr_all ? Lu : __builtin_amdgcn_readlane(wave_incl_scan(pass, lane), WAVE - 1);
                    n = kept_units(g);
                    const int ex = incl_h - tot_l;
                    int myq = -1, myQ = 0, myV = 0;
                    const bool owner_lane = n > 0 && ex < n && n <= incl_h;
                    if (owner_lane) {
                        int cum = ex;
#pragma unroll
                        for (int q = 0; q < 16; q++) {
                            if (myq < 0 && cum + v[q] >= n) { myq = q; myQ = n - cum; myV = v[q]; }
                            cum += v[q];
                        }
                    }
                    const int myB = lane * 16 + myq;
                    const uint64_t om = __ballot(owner_lane);
                    const int srcu = __builtin_amdgcn_readfirstlane(om ? __ffsll((unsigned long long) om) - 1 : 0);
                    const int B = om ? __builtin_amdgcn_readlane(myB, srcu) : -1;
                    const int quota = om ? __builtin_amdgcn_readlane(myQ, srcu) : 0;
                    const int in_B = om ? __builtin_amdgcn_readlane(myV, srcu) : 0;
                    const int nG = n - quota;
                    const bool whole_bin = in_B == quota; /* the cutoff bin is kept entirely: no ranking needed */
                    SEC(3);
                    /* pass 2: the units above the cutoff bin; those in it are kept directly or marked by unit index */
                    int gc = 0, ec = 0;
#pragma unroll 1
                    for (int q0 = 0; q0 < L; q0 += WAVE) {
                        uint32_t c1, c2, pe;
                        const uint32_t key_ = slot(q0, c1, c2, pe);
                        const int bin_ = key_ != 0xFFFFFFFFu ? (int) (key_ >> 14) : nb;
                        const bool is_g = bin_ < B, is_e = bin_ == B;
                        const uint64_t mg = __ballot(is_g), me = __ballot(is_e);
                        const bool take = is_g || (is_e && whole_bin);
                        emit(take, is_g ? gc + mbcnt64(mg) : nG + ec + mbcnt64(me), key_, c1, c2, pe);
                        if (is_e && !whole_bin) {
                            const uint32_t e = key_ & 0x3FFFu;
                            atomicOr(&bmp_c[e >> 5], 1u << (e & 31u));
                        }
                        gc += __popcll(mg);
                        ec += __popcll(me);
                    }
                    SEC(4);
                    if (!whole_bin && quota > 0) {
                        /* the first `quota` units of the cutoff bin in list order (= by unit index): a unit's rank is the number
                         * of marks below it */
                        wave_lds_fence();
                        {
                            uint32_t w[8];
                            int cnt_l = 0;
#pragma unroll
                            for (int q = 0; q < 8; q++) { w[q] = bmp_c[lane * 8 + q]; cnt_l += __popc(w[q]); }
                            int run = wave_incl_scan(cnt_l, lane) - cnt_l;
#pragma unroll
                            for (int q = 0; q < 8; q++) { pref[lane * 8 + q] = (uint32_t) run; run += __popc(w[q]); }
                        }
                        wave_lds_fence();
#pragma unroll 1
                        for (int q0 = 0; q0 < L; q0 += WAVE) {
                            uint32_t c1, c2, pe;
                            const uint32_t key_ = slot(q0, c1, c2, pe);
                            const bool is_e = key_ != 0xFFFFFFFFu && (int) (key_ >> 14) == B;
                            const uint32_t e = key_ & 0x3FFFu, wd = is_e ? e >> 5 : 0u;
                            const int rank = (int) pref[wd] + __popc(bmp_c[wd] & ((1u << (e & 31u)) - 1u));
                            emit(is_e && rank < quota, nG + rank, key_, c1, c2, pe);
                        }
                        wave_lds_fence();
#pragma unroll 1
                        for (int q0 = 0; q0 < L; q0 += WAVE) { /* the marks go */
                            uint32_t c1, c2, pe;
                            const uint32_t key_ = slot(q0, c1, c2, pe);
                            if (key_ != 0xFFFFFFFFu && (int) (key_ >> 14) == B) bmp_c[(key_ & 0x3FFFu) >> 5] = 0u;
                        }
                    }
                }
                wave_lds_fence();
                SEC(5);
                if (lane == 0) {
                    sh[32 + b] = (uint32_t) n; sh[40 + (b ^ 1)] = (uint32_t) cm;
                    sh[44 + b] = (w_c == 2 ? 1u : 0u) | (o_pm << 1) | (i_pm << 2); /* for the list stages */
                }
                /* the marks of the next merge units go */
                if (has_next && lane < cm) bmp_m[((kmn[lane] & 0x3FFFu) >> 5) & 511u] = 0u;
                SEC(6);
                ROLE_BARRIER();
            }
            } else {
            for (int k = 0; k < K; k++) {
                SEC(7);
                /* the descriptor of the column after next: requested first, so that its latency hides behind this column's work
                 * (a scalar load issued just before the barrier would be waited for there: the barrier drains lgkmcnt) */
                const int b = k & 1;
                /* what the chain needs of the column's CrossCol comes through LDS from the table wave (sh[48 + 2 b ..]): a scalar
                 * load here would sit in the same counter as the LDS traffic, and every LDS wait of the column would wait for it */
                const uint32_t cd0 = sh[48 + 4 * b], cd1 = sh[49 + 4 * b];
                uint32_t *skey = sel + b * 2 * PRUNE_SP, *snp = skey + PRUNE_SP;
                const uint32_t *tA = tab + b * 2 * PRUNE_TAB * 128, *tB = tA + PRUNE_TAB * 128;
                const uint32_t *cntA = tA, *startA = tA + 128, *listA = tA + 256, *nxA = tA + 384;
                const uint32_t *cntB = tB, *startB = tB + 128, *listB = tB + 256, *nxB = tB + 384;
                const uint16_t *bin_k = bins + b * cap_c;
                uint32_t *hk = hist + b * nb_r;
                uint32_t *kmn = kml + (b ^ 1) * PRUNE_SP; /* the kept merge cells leading into column k + 1 */
                const uint32_t cflags = cd1 & 0xFFu;
                const bool inv = (cflags & MRP_XF_INVERTED) != 0;
                const uint32_t C2 = (cd0 & 0xFFFFu) > 128u ? 128u : (cd0 & 0xFFFFu), Mb = cd0 >> 16;
                const bool a_cp = inv && (cd1 & 0x100u), b_cp = inv && (cd1 & 0x200u);
                const bool out_ap = (cflags & MRP_XF_OUT_A_PAIRED) != 0, out_bp = (cflags & MRP_XF_OUT_B_PAIRED) != 0;
                const bool has_next = k + 1 < K;
                const int nkm = (int) sh[40 + b];
                /* this lane's kept merge cells (two at most): the parents' cells each links, as list ranges.  An entry of the
                 * list is m | i << 14 | j << 21: the merge cell and its index on either side (what links the parents' cells) */
                uint32_t off_[2];
                int tot[2];
                {
                    uint32_t m_[2], sa[2], sb_[2];
                    int nbb[2];
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const int idx = lane + u * WAVE;
                        const bool has = idx < nkm;
                        const uint32_t ent = has ? kml[b * PRUNE_SP + idx] : 0u;
                        m_[u] = ent & 0x3FFFu;
                        const uint32_t i = (ent >> 14) & 127u, j = (ent >> 21) & 127u;
                        sa[u] = startA[i]; sb_[u] = startB[j];
                        const int na = has ? (int) cntA[i] : 0;
                        nbb[u] = has ? (int) cntB[j] : 0;
                        tot[u] = na * nbb[u];
                    }
                    /* candidate q of the column belongs to the kept merge cell whose range [off, off + tot) holds it: ranges in
                     * list order of the kept merge cells (first all of this wave's "u = 0" entries, then the "u = 1" ones) */
                    const int i0 = wave_incl_scan(tot[0], lane);
                    const int t0 = __builtin_amdgcn_readlane(i0, WAVE - 1);
                    off_[0] = (uint32_t) (i0 - tot[0]);
                    off_[1] = (uint32_t) t0;
                    if (nkm > WAVE) {
                        const int i1 = wave_incl_scan(tot[1], lane);
                        off_[1] = (uint32_t) (t0 + i1 - tot[1]);
                    }
#pragma unroll
                    for (int u = 0; u < 2; u++) { /* what a candidate needs of its kept merge cell: two dwords */
                        const int idx = lane + u * WAVE;
                        if (idx < nkm) *reinterpret_cast<uint2 *>(minfo + 2 * idx) = make_uint2(m_[u] | ((uint32_t) nbb[u] << 14), off_[u] | (sa[u] << 14) | (sb_[u] << 21));
                    }
                }
                int L;
                {
                    const int l0 = wave_incl_scan(tot[0] + tot[1], lane);
                    L = __builtin_amdgcn_readlane(l0, WAVE - 1);
                }
                SEC(0);
                const bool thr_all = p.thr_bin >= nb - 1;
                const bool keep_all = L <= p.min_p || (thr_all && L <= p.max_p); /* the loop of :1073-1079 drops nothing */
                /* One chunk = up to 512 consecutive candidates in eight slots of 64 (slot j, lane l: candidate q0 + 64 j + l);
                 * only the ns slots the chunk needs are worked on (a column links 130 cells on average: two or three slots).
                 * Per slot: the owner (the kept merge cell whose range holds the candidate) by a prefix maximum over "a range
                 * starts here" marks, the owner's record, the two parent cells from the inverted lists, the cell index by
                 * the closed form of the cross product order, its posterior bin.  The slots' chains are independent. */
                uint32_t key[8], aux[8]; /* bin << 14 | cell (0xFFFFFFFF: none);  c1 | c2 << 8 | prev merge cell << 16 */
                auto load_chunk = [&](int q0) -> int {
                    const int left = L - q0;
                    const int ns = left >= 512 ? 8 : (left + 63) >> 6;
#pragma unroll
                    for (int u = 0; u < 2; u++) { /* marks: index + 1 of the kept merge cell whose range starts (or continues) here */
                        const int idx = lane + u * WAVE;
                        const int lo = (int) off_[u], hi = lo + tot[u];
                        if (tot[u] > 0 && hi > q0 && lo < q0 + 512) heads[lo > q0 ? lo - q0 : 0] = (uint32_t) idx + 1u;
                    }
                    wave_lds_fence();
                    uint32_t own[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        own[j] = 0u;
                        if (j < ns) { own[j] = heads[j * WAVE + lane]; heads[j * WAVE + lane] = 0u; } /* left clean for the next use */
                    }
                    uint32_t carry = 0u; /* the latest mark of the slots before */
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        if (j < ns) { /* inclusive prefix maximum over the lanes (marks grow with the position: max = latest) */
                            uint32_t v = own[j], t;
                            t = dpp_mov<0x111>(v); if ((lane & 15) >= 1) v = t > v ? t : v;
                            t = dpp_mov<0x112>(v); if ((lane & 15) >= 2) v = t > v ? t : v;
                            t = dpp_mov<0x114>(v); if ((lane & 15) >= 4) v = t > v ? t : v;
                            t = dpp_mov<0x118>(v); if ((lane & 15) >= 8) v = t > v ? t : v;
                            t = dpp_mov<0x142, 0xa>(v); if ((lane & 31) >= 16) v = t > v ? t : v;
                            t = dpp_mov<0x143, 0xc>(v); if (lane >= 32) v = t > v ? t : v;
                            v = v > carry ? v : carry;
                            own[j] = v;
                            carry = (uint32_t) __builtin_amdgcn_readlane((int) v, WAVE - 1);
                        }
                    }
                    uint32_t w0[8], w1[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        key[j] = 0xFFFFFFFFu;
                        if (j < ns) {
                            const bool act = j * WAVE + lane < left && own[j] > 0u;
                            const uint2 r = act ? *reinterpret_cast<const uint2 *>(minfo + 2 * (own[j] - 1u)) : make_uint2(0u, 0u);
                            w0[j] = r.x; w1[j] = r.y;
                            if (act) key[j] = 0u;
                        }
                    }
                    uint32_t c1_[8], c2_[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        if (j < ns) {
                            const uint32_t nbq = (w0[j] >> 14) & 0xFFu, off = w1[j] & 0x3FFFu, sa = (w1[j] >> 14) & 0x7Fu, sb = (w1[j] >> 21) & 0x7Fu;
                            const uint32_t t = (uint32_t) (q0 + j * WAVE + lane) - off;
                            /* t = x * nb + y, nb <= 128: exact through a float reciprocal (t < 2^14) */
                            uint32_t x = (uint32_t) ((float) t * __builtin_amdgcn_rcpf((float) (nbq ? nbq : 1u)));
                            if (x * nbq > t) x--;
                            if ((x + 1u) * nbq <= t) x++;
                            const uint32_t y = t - x * nbq;
                            c1_[j] = listA[(sa + x) & 127u];
                            c2_[j] = listB[(sb + y) & 127u];
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        if (j < ns) {
                            const uint32_t e = key[j] == 0u ? pair_index(c1_[j], c2_[j], C2, inv, a_cp, b_cp) : 0u;
                            aux[j] = c1_[j] | (c2_[j] << 8) | ((w0[j] & 0x3FFFu) << 16);
                            const uint32_t bin_ = bin_k[e];
                            if (key[j] == 0u) key[j] = (bin_ << 14) | e;
                        }
                    }
                    return ns;
                };
                /* a selected candidate: its slot of the selection, and -- the first time its next merge cell is seen -- that merge
                 * cell as a kept merge cell of the next column.  Called in wave-uniform control flow, `take` per lane. */
                int cm = 0;
                auto emit = [&](bool take, int pos, uint32_t key_, uint32_t aux_) {
                    const uint32_t ii = nxA[aux_ & 0x7Fu], jj = nxB[(aux_ >> 8) & 0x7Fu];
                    const uint32_t nxt = has_next ? pair_index(ii, jj, Mb, inv, out_ap, out_bp) : 0u;
                    if (take) { skey[pos] = key_; snp[pos] = nxt | (aux_ & 0xFFFF0000u); }
                    if (has_next) {
                        bool first = false;
                        if (take) {
                            const uint32_t bit = 1u << (nxt & 31u);
                            first = (atomicOr(&bmp_m[(nxt >> 5) & 511u], bit) & bit) == 0u;
                        }
                        const uint64_t fm = __ballot(first);
                        if (first) kmn[cm + __popcll(fm & lt_mask)] = nxt | (ii << 14) | (jj << 21);
                        cm += __popcll(fm);
                    }
                };
                const int n_chunks = (L + 511) >> 9;
                int n = 0;
                if (keep_all) {
                    SEC_COUNT(8);
                    int at = 0;
                    for (int c = 0; c < n_chunks; c++) {
                        const int ns = load_chunk(c << 9);
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            if (j < ns) {
                                const bool act = key[j] != 0xFFFFFFFFu;
                                const uint64_t am = __ballot(act);
                                emit(act, at + __popcll(am & lt_mask), key[j], aux[j]);
                                at += __popcll(am);
                            }
                        }
                    }
                    n = at;
                    SEC(1);
                } else if (L <= 4 * WAVE) {
                    /* Up to four slots of candidates: selection by SORTING.  The keys bin << 14 | cell are distinct and their ascending
                     * order is the stable descending-posterior order of the reference (hmm.c:1043, :1071: smaller bin = larger
                     * posterior, ties in list order = cell index): the kept cells are the first n of the sorted candidates.  One
                     * bitonic sort in registers over as many slots as the column needs replaces histogram, cutoff search,
                     * selection pass and tie ranking (measured: 29 000 cycles per such column against 10 000 for a column that
                     * keeps all its candidates).  What rides along with a candidate (its parent cells, the merge cell it comes
                     * from) is re-derived from the cell index for the kept ones. */
                    SEC_COUNT(9);
                    const int ns = load_chunk(0);
                    int g = L;
                    if (!thr_all) {
                        g = 0;
#pragma unroll
                        for (int j = 0; j < 8; j++)
                            if (j < ns) g += __popcll(__ballot(key[j] != 0xFFFFFFFFu && (int) (key[j] >> 14) <= p.thr_bin));
                    }
                    n = kept_count(L, g, p.min_p, p.max_p);
                    if (ns <= 2) { uint32_t kk[2] = {key[0], key[1]}; wave_bitonic_sort_n<2>(kk, lane); key[0] = kk[0]; key[1] = kk[1]; }
                    else { uint32_t kk[4] = {key[0], key[1], key[2], key[3]}; wave_bitonic_sort_n<4>(kk, lane); key[0] = kk[0]; key[1] = kk[1]; key[2] = kk[2]; key[3] = kk[3]; }
                    const uint32_t Pb = cd1 >> 16;
                    const bool in_ap = (cflags & MRP_XF_IN_A_PAIRED) != 0, in_bp = (cflags & MRP_XF_IN_B_PAIRED) != 0;
                    const uint32_t *pvA = tA + 512, *pvB = tB + 512;
                    const float rc2 = __builtin_amdgcn_rcpf((float) (C2 ? C2 : 1u)), rc22 = __builtin_amdgcn_rcpf((float) (C2 ? 2u * C2 : 1u));
#pragma unroll
                    for (int j = 0; j < 2; j++) { /* n <= S <= 128: the kept cells are in the first two registers */
                        if (j * WAVE < n) {
                            const bool take = j * WAVE + lane < n;
                            const uint32_t e = take ? key[j] & 0x3FFFu : 0u;
                            uint32_t c1, c2; /* cross_cell(e), the divisions through float reciprocals (exact: e < 2^14, one correction step) */
                            if (!inv) {
                                uint32_t q = (uint32_t) ((float) e * rc2);
                                if (q * C2 > e) q--;
                                if ((q + 1u) * C2 <= e) q++;
                                c1 = q; c2 = e - q * C2;
                            } else if (!a_cp) { c1 = 0u; c2 = e; }
                            else {
                                uint32_t r = (uint32_t) ((float) e * rc22);
                                if (r * 2u * C2 > e) r--;
                                if ((r + 1u) * 2u * C2 <= e) r++;
                                const uint32_t t = e - r * 2u * C2, hh = t >> 1;
                                if (t & 1u) { c1 = 2u * r + 1u; c2 = b_cp ? (hh ^ 1u) : hh; } else { c1 = 2u * r; c2 = hh; }
                            }
                            c1 &= 127u; c2 &= 127u;
                            const uint32_t prv = k > 0 ? pair_index(pvA[c1], pvB[c2], Pb, inv, in_ap, in_bp) : 0u;
                            emit(take, j * WAVE + lane, key[j], c1 | (c2 << 8) | (prv << 16));
                        }
                    }
                    SEC(4);
                } else {
                    /* pass 1: histogram of the posterior bins */
                    SEC_COUNT(n_chunks > 1 ? 11 : 10);
                    int ns = 0;
                    for (int c = 0; c < n_chunks; c++) {
                        ns = load_chunk(c << 9);
#pragma unroll
                        for (int j = 0; j < 8; j++)
                            if (j < ns && key[j] != 0xFFFFFFFFu) {
                                const int bin_ = (int) (key[j] >> 14);
                                atomicAdd(&hk[(bin_ & 15) * WAVE + (bin_ >> 4)], 1u);
                            }
                    }
                    wave_lds_fence();
                    SEC(2);
                    /* cutoff bin and quota: lane l owns the 16 consecutive bins [16 l, 16 l + 16); the histogram is stored
                     * lane-major, so these reads are conflict-free and the search has a fixed, short cost */
                    int v[16];
                    int tot_l = 0, pass = 0;
                    const int thr_rel = p.thr_bin - lane * 16; /* bins q <= thr_rel of this lane pass the threshold */
#pragma unroll
                    for (int q = 0; q < 16; q++) {
                        v[q] = (int) hk[q * WAVE + lane];
                        tot_l += v[q];
                        if (q <= thr_rel) pass += v[q];
                    }
                    const int incl = wave_incl_scan(tot_l, lane);
                    const int g = thr_all ? L : __builtin_amdgcn_readlane(wave_incl_scan(pass, lane), WAVE - 1);
                    n = kept_count(L, g, p.min_p, p.max_p);
                    const int ex = incl - tot_l;
                    int myq = -1, myQ = 0, myV = 0;
                    const bool owner_lane = n > 0 && ex < n && n <= incl;
                    if (owner_lane) {
                        int cum = ex;
#pragma unroll
                        for (int q = 0; q < 16; q++) {
                            if (myq < 0 && cum + v[q] >= n) { myq = q; myQ = n - cum; myV = v[q]; }
                            cum += v[q];
                        }
                    }
                    const int myB = lane * 16 + myq;
                    const uint64_t om = __ballot(owner_lane);
                    const int srcu = __builtin_amdgcn_readfirstlane(om ? __ffsll((unsigned long long) om) - 1 : 0);
                    const int B = om ? __builtin_amdgcn_readlane(myB, srcu) : -1;
                    const int quota = om ? __builtin_amdgcn_readlane(myQ, srcu) : 0;
                    const int in_B = om ? __builtin_amdgcn_readlane(myV, srcu) : 0;
                    const int nG = n - quota;
                    const bool whole_bin = in_B == quota; /* the cutoff bin is kept entirely: no ranking needed */
                    SEC(3);
                    /* pass 2: the cells above the cutoff bin; those in it are kept directly or marked by cell index */
                    int gc = 0, ec = 0;
                    for (int c = 0; c < n_chunks; c++) {
                        if (n_chunks > 1) ns = load_chunk(c << 9); /* a single chunk is still in the registers */
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            if (j >= ns) continue;
                            const int bin_ = key[j] != 0xFFFFFFFFu ? (int) (key[j] >> 14) : nb;
                            const bool is_g = bin_ < B, is_e = bin_ == B;
                            const uint64_t mg = __ballot(is_g), me = __ballot(is_e);
                            const bool take = is_g || (is_e && whole_bin);
                            emit(take, is_g ? gc + __popcll(mg & lt_mask) : nG + ec + __popcll(me & lt_mask), key[j], aux[j]);
                            if (is_e && !whole_bin) {
                                const uint32_t e = key[j] & 0x3FFFu;
                                atomicOr(&bmp_c[e >> 5], 1u << (e & 31u));
                            }
                            gc += __popcll(mg);
                            ec += __popcll(me);
                        }
                    }
                    SEC(4);
                    if (!whole_bin && quota > 0) {
                        /* the first `quota` cells of the cutoff bin in list order (= by cell index): a cell's rank is the number
                         * of marks below it -- marks before its bitmap word (prefix counts, lane l owns words 8 l .. 8 l + 7) plus
                         * those below it in the word */
                        wave_lds_fence();
                        {
                            uint32_t w[8];
                            int cnt_l = 0;
#pragma unroll
                            for (int q = 0; q < 8; q++) { w[q] = bmp_c[lane * 8 + q]; cnt_l += __popc(w[q]); }
                            int run = wave_incl_scan(cnt_l, lane) - cnt_l;
#pragma unroll
                            for (int q = 0; q < 8; q++) { pref[lane * 8 + q] = (uint32_t) run; run += __popc(w[q]); }
                        }
                        wave_lds_fence();
                        for (int c = 0; c < n_chunks; c++) {
                            if (n_chunks > 1) ns = load_chunk(c << 9);
#pragma unroll
                            for (int j = 0; j < 8; j++) {
                                if (j >= ns) continue;
                                const bool is_e = key[j] != 0xFFFFFFFFu && (int) (key[j] >> 14) == B;
                                const uint32_t e = key[j] & 0x3FFFu, wd = is_e ? e >> 5 : 0u;
                                const int rank = (int) pref[wd] + __popc(bmp_c[wd] & ((1u << (e & 31u)) - 1u));
                                emit(is_e && rank < quota, nG + rank, key[j], aux[j]);
                            }
                        }
                        wave_lds_fence();
                        for (int c = 0; c < n_chunks; c++) { /* the marks go */
                            if (n_chunks > 1) ns = load_chunk(c << 9);
#pragma unroll
                            for (int j = 0; j < 8; j++)
                                if (j < ns && key[j] != 0xFFFFFFFFu && (int) (key[j] >> 14) == B) bmp_c[(key[j] & 0x3FFFu) >> 5] = 0u;
                        }
                    }
                }
                wave_lds_fence();
                SEC(5);
                if (lane == 0) { sh[32 + b] = (uint32_t) n; sh[40 + (b ^ 1)] = (uint32_t) cm; }
                /* the marks of the next merge cells go */
                if (has_next) {
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const int idx = lane + u * WAVE;
                        if (idx < cm) bmp_m[((kmn[idx] & 0x3FFFu) >> 5) & 511u] = 0u;
                    }
                }
                SEC(6);
                ROLE_BARRIER();
            }
            } /* PAIRS or not */
            ROLE_CLK_DONE(0);
            SEC_DONE();
            lds_barrier();
            if (PAIRS) lds_barrier(); /* (the list stages of the chain on pairs run three columns deep) */
        } else if (wave == 1) {
            int64_t mcell_prev = 0; /* first merge cell of the merge column after column k - 1 */
            SweepCol scur = k_load(d.scols + h.col0);
            lds_barrier();
            ROLE_CLK_INIT();
            if constexpr (PAIRS) {
                for (int k = 0; k < K; k++) {
                    if (k > 1) stage1_finish(k - 2);
                    if (k > 0) stage1_ask(k - 1, mcell_prev);
                    mcell_prev = scur.mcell_off;
                    if (k + 1 < K) scur = k_load(d.scols + h.col0 + k + 1);
                    ROLE_BARRIER();
                }
                ROLE_CLK_DONE(1);
                if (K > 1) stage1_finish(K - 2);
                stage1_ask(K - 1, mcell_prev);
                lds_barrier();
                stage1_finish(K - 1);
                lds_barrier();
            } else {
                for (int k = 0; k < K; k++) {
                    if (k > 0) lists_stage1(k - 1, mcell_prev);
                    mcell_prev = scur.mcell_off;
                    if (k + 1 < K) scur = k_load(d.scols + h.col0 + k + 1);
                    ROLE_BARRIER();
                }
                ROLE_CLK_DONE(1);
                lists_stage1(K - 1, mcell_prev);
                lds_barrier();
            }
        } else if (wave == 2) {
            lds_barrier();
            ROLE_CLK_INIT();
            if constexpr (PAIRS) { /* a column later than the general chain: stage 1 takes two steps */
                for (int k = 0; k < K; k++) {
                    if (k > 2) lists_stage2(k - 3);
                    ROLE_BARRIER();
                }
                ROLE_CLK_DONE(2);
                if (K > 2) lists_stage2(K - 3);
                lds_barrier();
                if (K > 1) lists_stage2(K - 2);
                lds_barrier();
                lists_stage2(K - 1);
            } else {
                for (int k = 0; k < K; k++) {
                    if (k > 1) lists_stage2(k - 2);
                    ROLE_BARRIER();
                }
                ROLE_CLK_DONE(2);
                if (K > 1) lists_stage2(K - 2);
                lds_barrier();
                lists_stage2(K - 1);
            }
        } else if (wave == 3) {
            /* wave 3: the parents' transitions of the column after next.  Three columns are in the pipe: the tables of column
             * t are built from the registers while the transitions of column t + 1 and the descriptor of column t + 2 are in
             * flight.  The descriptors come by VECTOR loads (lane l = dword l of the CrossCol, lanes 16.. = the SweepCol) and
             * are taken apart with v_readlane a step later: a scalar load shares lgkmcnt with the LDS traffic of the table
             * build, and a wave that first has to wait for its descriptor before it can ask for the transitions it points to
             * pays two memory latencies per column -- with the chain on complement pairs this wave had become the slowest role. */
            uint32_t r_na[2] = {0u, 0u}, r_nb[2] = {0u, 0u};
            int32_t t_f[W == 4 ? CPT : 1], t_b[W == 4 ? CPT : 1]; /* T = 256: f and b of the column whose tables are built next */
            int t_n = 0;
            CrossCol tcc = {};
            int tcol = 0;
            auto desc_load = [&](int col) -> uint32_t {
                uint32_t v = 0u;
                if (col < K) {
                    if (lane < 16) v = reinterpret_cast<const uint32_t *>(d.ccols + h.col0 + col)[lane];
                    else if (W == 4 && lane < 24) v = reinterpret_cast<const uint32_t *>(d.scols + h.col0 + col)[lane - 16];
                }
                return v;
            };
            auto fld = [&](uint32_t v, int l) -> uint32_t { return (uint32_t) __builtin_amdgcn_readlane((int) v, l); };
            /* the transitions (and, T = 256, f and b) of the column the descriptor vector dv describes: requests only */
            auto tab_load = [&](uint32_t dv) {
                CrossCol c = {};
                c.a_part = reinterpret_cast<const uint64_t *>(((uint64_t) fld(dv, 1) << 32) | fld(dv, 0));
                c.b_part = reinterpret_cast<const uint64_t *>(((uint64_t) fld(dv, 3) << 32) | fld(dv, 2));
                c.a_np = reinterpret_cast<const uint32_t *>(((uint64_t) fld(dv, 5) << 32) | fld(dv, 4));
                c.b_np = reinterpret_cast<const uint32_t *>(((uint64_t) fld(dv, 7) << 32) | fld(dv, 6));
                const uint32_t w10 = fld(dv, 10), w11 = fld(dv, 11), w12 = fld(dv, 12), w13 = fld(dv, 13), w14 = fld(dv, 14);
                c.C1 = (uint16_t) w10; c.C2 = (uint16_t) (w10 >> 16); c.Ma = (uint16_t) w11; c.Mb = (uint16_t) (w11 >> 16);
                c.Pa = (uint16_t) w12; c.Pb = (uint16_t) (w12 >> 16);
                c.d1 = (uint8_t) w13; c.d2 = (uint8_t) (w13 >> 8); c.out_a = (uint8_t) (w13 >> 16); c.out_b = (uint8_t) (w13 >> 24);
                c.in_a = (uint8_t) w14; c.in_b = (uint8_t) (w14 >> 8); c.flags = (uint8_t) (w14 >> 16);
                tcc = c;
                if (tcol < K) {
                    if (W == 4) {
                        const int64_t cell_off = (int64_t) (((uint64_t) fld(dv, 17) << 32) | fld(dv, 16));
                        t_n = (int) fld(dv, 20);
#pragma unroll
                        for (int j = 0; j < (W == 4 ? CPT : 1); j++) {
                            const int cell = lane + j * WAVE;
                            t_f[j] = cell < t_n ? d.cell_f32[cell_off + cell] : MRP_NEG_I32;
                            t_b[j] = cell < t_n ? d.cell_b32[cell_off + cell] : MRP_NEG_I32;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const uint32_t cidx = (uint32_t) (lane + u * WAVE);
                        r_na[u] = (tcc.a_np && cidx < tcc.C1) ? tcc.a_np[cidx] : 0u;
                        r_nb[u] = (tcc.b_np && cidx < tcc.C2) ? tcc.b_np[cidx] : 0u;
                    }
                }
            };
            /* inverted transition lists of both sides (cells grouped by the merge cell they come from: count, first entry, list),
             * the two sides side by side between the same three LDS fences */
            auto tab_build_sides = [&](uint32_t *tA5, uint32_t CA, uint32_t PA, uint32_t in_a, uint32_t out_a, uint32_t CB, uint32_t PB, uint32_t in_b, uint32_t out_b) {
                uint32_t *t5[2] = {tA5, tA5 + PRUNE_TAB * 128};
                const uint32_t C_[2] = {CA, CB}, in_[2] = {in_a, in_b}, out_[2] = {out_a, out_b};
                const uint32_t G_[2] = {PA < 1u ? 1u : (PA > 128u ? 128u : PA), PB < 1u ? 1u : (PB > 128u ? 128u : PB)};
#pragma unroll
                for (int sd = 0; sd < 2; sd++) { t5[sd][lane] = 0u; t5[sd][lane + WAVE] = 0u; }
                wave_lds_fence();
                uint32_t rank[2][2] = {{0u, 0u}, {0u, 0u}}, grp_[2][2] = {{0u, 0u}, {0u, 0u}};
#pragma unroll
                for (int sd = 0; sd < 2; sd++) {
                    uint32_t *cnt = t5[sd], *nx = t5[sd] + 384, *pv = t5[sd] + 512;
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const uint32_t c = (uint32_t) (lane + u * WAVE);
                        const uint32_t rn = sd == 0 ? r_na[u] : r_nb[u];
                        if (c < C_[sd]) {
                            uint32_t pvi = in_[sd] == MRP_CONN_REAL ? (rn >> 16) : (in_[sd] == MRP_CONN_IDENT ? c : 0u);
                            const uint32_t nxi = out_[sd] == MRP_CONN_REAL ? (rn & 0xFFFFu) : (out_[sd] == MRP_CONN_IDENT ? c : 0u);
                            if (pvi >= G_[sd]) { errbits |= MRP_ENGINE_ERR_RANGE; pvi = 0u; }
                            pv[c] = pvi; nx[c] = nxi;
                            grp_[sd][u] = pvi;
                            rank[sd][u] = atomicAdd(&cnt[pvi], 1u);
                        }
                    }
                }
                wave_lds_fence();
                /* exclusive scan of the group sizes (two per lane and side: groups lane and lane + 64) */
                int c0[2], c1[2], i0[2], i1[2];
#pragma unroll
                for (int sd = 0; sd < 2; sd++) { c0[sd] = (int) t5[sd][lane]; c1[sd] = (int) t5[sd][lane + WAVE]; }
#pragma unroll
                for (int sd = 0; sd < 2; sd++) { i0[sd] = wave_incl_scan_bc(c0[sd]); i1[sd] = wave_incl_scan_bc(c1[sd]); }
#pragma unroll
                for (int sd = 0; sd < 2; sd++) {
                    uint32_t *start = t5[sd] + 128;
                    const int t0 = __builtin_amdgcn_readlane(i0[sd], WAVE - 1);
                    /* (the chain on pairs reads a group's first entry and size as one word) */
                    start[lane] = (uint32_t) (i0[sd] - c0[sd]) | (PAIRS ? (uint32_t) c0[sd] << 8 : 0u);
                    start[lane + WAVE] = (uint32_t) (t0 + i1[sd] - c1[sd]) | (PAIRS ? (uint32_t) c1[sd] << 8 : 0u);
                }
                wave_lds_fence();
#pragma unroll
                for (int sd = 0; sd < 2; sd++) {
                    uint32_t *start = t5[sd] + 128, *list = t5[sd] + 256;
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const uint32_t c = (uint32_t) (lane + u * WAVE);
                        if (c < C_[sd]) list[(start[grp_[sd][u]] & 0xFFu) + rank[sd][u]] = c;
                    }
                }
            };
            auto tab_build = [&]() { /* tables of column tcol into buffer tcol & 1, from the registers loaded last time */
                if (tcol < K) {
                    uint32_t *tb = tab + (tcol & 1) * 2 * PRUNE_TAB * 128;
                    if (lane == 0) { /* the chain's view of the column (read after the barrier that ends this step) */
                        sh[48 + 4 * (tcol & 1)] = (uint32_t) tcc.C2 | ((uint32_t) tcc.Mb << 16);
                        sh[49 + 4 * (tcol & 1)] = (uint32_t) tcc.flags | ((tcc.a_part && tcc.d1 > 0) ? 0x100u : 0u) | ((tcc.b_part && tcc.d2 > 0) ? 0x200u : 0u) |
                                                  ((uint32_t) tcc.Pb << 16);
                        if (PAIRS) { /* what the chain on pairs derives from these flags, as bits (see its unit() / parity() rule) */
                            const bool a_cp = tcc.a_part && tcc.d1 > 0, b_cp = tcc.b_part && tcc.d2 > 0;
                            const bool o_a = (tcc.flags & MRP_XF_OUT_A_PAIRED) != 0, o_b = (tcc.flags & MRP_XF_OUT_B_PAIRED) != 0;
                            const bool i_a = (tcc.flags & MRP_XF_IN_A_PAIRED) != 0, i_b = (tcc.flags & MRP_XF_IN_B_PAIRED) != 0;
                            sh[50 + 4 * (tcol & 1)] = ((a_cp && b_cp) ? 1u : 0u) | ((!a_cp && b_cp) ? 2u : 0u) | (a_cp ? 4u : 0u) | ((o_a && o_b) ? 8u : 0u) |
                                                      ((!o_a && o_b) ? 16u : 0u) | (o_a ? 32u : 0u) | ((o_a || o_b) ? 64u : 0u) | ((i_a || i_b) ? 128u : 0u) |
                                                      (i_a ? 256u : 0u) | ((!i_a && i_b) ? 512u : 0u) | ((a_cp || b_cp) ? 1024u : 0u) |
                                                      ((!(i_a || i_b) && (a_cp || b_cp)) ? 2048u : 0u);
                        }
                    }
                    if (W == 4) { /* the column's posterior bins */
                        uint16_t *dst = bins + (tcol & 1) * cap_c;
                        if (t_n > CPT * WAVE) { errbits |= MRP_ENGINE_ERR_RANGE; t_n = CPT * WAVE; }
#pragma unroll
                        for (int j = 0; j < (W == 4 ? CPT : 1); j++) {
                            const int cell = lane + j * WAVE;
                            if (PAIRS && !units) { if (cell < t_n && (cell & 1) == 0) dst[cell >> 1] = (uint16_t) posterior_bin(t_f[j], t_b[j], total, nb, &errbits); }
                            else if (cell < t_n) dst[cell] = (uint16_t) posterior_bin(t_f[j], t_b[j], total, nb, &errbits);
                        }
                    }
                    uint32_t C1 = tcc.C1, C2 = tcc.C2;
                    if (C1 > 128u || C2 > 128u) { errbits |= MRP_ENGINE_ERR_RANGE; C1 = C1 > 128u ? 128u : C1; C2 = C2 > 128u ? 128u : C2; }
                    tab_build_sides(tb, C1, tcc.Pa, tcc.in_a, tcc.out_a, C2, tcc.Pb, tcc.in_b, tcc.out_b);
                }
            };

            /* the parents' transitions: tables of column 0, requests for column 1, descriptor of column 2 */
            uint32_t dv_next;
            tcol = 0; tab_load(desc_load(0)); tab_build();
            tcol = 1; tab_load(desc_load(1));
            dv_next = desc_load(2);
            lds_barrier();
            ROLE_CLK_INIT();
            for (int k = 0; k < K; k++) {
                /* tables of column k + 1 (requested while column k - 1 was worked on), then the requests for k + 2 (its
                 * descriptor was requested a step ago) and the descriptor of k + 3 */
                tab_build();
                tcol++;
                tab_load(dv_next);
                dv_next = desc_load(k + 3);
                uint32_t *hn = hist + ((k + 1) & 1) * nb_r;
                if (!PAIRS || sh[56 + ((k + 1) & 1)] != 0u) { /* (the chain on pairs says when it has used a histogram: one column in a hundred) */
                    for (int i = lane; i < nb_r; i += WAVE) hn[i] = 0u;
                    if (PAIRS && lane == 0) sh[56 + ((k + 1) & 1)] = 0u;
                }
                ROLE_BARRIER();
            }
            ROLE_CLK_DONE(3);
            lds_barrier();
            if (PAIRS) lds_barrier();
        } else {
            /* bins groups (waves 4..): group g handles the columns g, g + 2, ...  A step of its loop stores the bins of the
             * column whose f and b sit in the registers and requests the column after next into the same registers; the
             * group is idle during the other group's step.  Loads are issued for every register slot whatever the column's
             * size (beyond the column -- and beyond the last column -- the buffer descriptor's range check returns 0 without
             * touching memory): a load under a condition would make every slot a loop-carried merge of old and new value,
             * which costs a second register set. */
            int32_t r_f[CPT * VEC], r_b[CPT * VEC];
            const int bw = wave - 4, grp = NGRP == 2 ? (bw & 1) : 0, gidx = NGRP == 2 ? (bw >> 1) : bw;
            const int base_c = gidx * WAVE + lane, voff = base_c * 4 * VEC;
            int n_have = 0; /* cells of the column in the registers */
            auto bins_load = [&](int col) {
                const bool valid = col < K;
                const SweepCol c = k_load(d.scols + h.col0 + (valid ? col : 0));
                n_have = valid ? c.n_cells : 0;
                /* (VEC = 4: the last load of a column reads up to three cells of what follows it -- the hmm's cells are padded
                 * to a multiple of four, the arrays end with slack -- and bins_store leaves them out) */
                const int bytes_ = __builtin_amdgcn_readfirstlane(((n_have + VEC - 1) & ~(VEC - 1)) * 4);
                const auto rf_ = prune_rsrc(d.cell_f32 + c.cell_off, bytes_);
                const auto rb_ = prune_rsrc(d.cell_b32 + c.cell_off, bytes_);
#pragma unroll
                for (int j = 0; j < CPT; j++) { /* one lane offset for all loads; the step from load to load rides in the scalar offset */
                    if (VEC == 1) {
                        r_f[j] = (int32_t) __builtin_amdgcn_raw_buffer_load_b32(rf_, voff, j * LG * 4, 0);
                        r_b[j] = (int32_t) __builtin_amdgcn_raw_buffer_load_b32(rb_, voff, j * LG * 4, 0);
                    } else {
                        const auto vf = __builtin_amdgcn_raw_buffer_load_b128(rf_, voff, j * LG * 16, 0);
                        const auto vb = __builtin_amdgcn_raw_buffer_load_b128(rb_, voff, j * LG * 16, 0);
#pragma unroll
                        for (int q = 0; q < VEC; q++) { r_f[j * VEC + q] = (int32_t) vf[q]; r_b[j * VEC + q] = (int32_t) vb[q]; }
                    }
                }
            };
            auto bins_store = [&](int col) { /* the bins of column col into buffer col & 1 */
                uint16_t *dst = bins + (col & 1) * cap_c;
                const int nj = (n_have + LG * VEC - 1) / (LG * VEC);
#pragma unroll
                for (int j = 0; j < CPT; j++) {
                    if (j < nj) { /* (the lane's cell against a scalar bound, the step in the store's immediate offset: no
                                   * per-slot index registers) */
                        if (VEC == 1) {
                            if (PAIRS && !units) { /* the even cell of a unit writes the unit's bin */
                                if (base_c < n_have - j * LG && (base_c & 1) == 0) dst[(base_c + j * LG) >> 1] = (uint16_t) posterior_bin(r_f[j], r_b[j], total, nb, &errbits);
                            } else if (base_c < n_have - j * LG) dst[base_c + j * LG] = (uint16_t) posterior_bin(r_f[j], r_b[j], total, nb, &errbits);
                        } else {
                            const int left = n_have - (j * LG + base_c) * VEC; /* cells of this load inside the column */
                            if (left > 0) {
                                if (PAIRS && !units) { /* four cells = two units: the bins of cells 0 and 2 as one dword */
                                    const uint32_t b0 = (uint32_t) posterior_bin(r_f[j * VEC], r_b[j * VEC], total, nb, &errbits);
                                    const uint32_t b1 = left > 2 ? (uint32_t) posterior_bin(r_f[j * VEC + (VEC > 2 ? 2 : 0)], r_b[j * VEC + (VEC > 2 ? 2 : 0)], total, nb, &errbits) : 0u;
#ifdef MRP_PRUNE_CHECK_PAIRS /* development: the twins' f and b really are equal */
                                    if (left > 1 && (r_f[j * VEC] != r_f[j * VEC + 1] || r_b[j * VEC] != r_b[j * VEC + 1])) errbits |= MRP_ENGINE_ERR_STRUCTURE;
                                    if (left > 3 && (r_f[j * VEC + 2] != r_f[j * VEC + 3] || r_b[j * VEC + 2] != r_b[j * VEC + 3])) errbits |= MRP_ENGINE_ERR_STRUCTURE;
#endif
                                    *reinterpret_cast<uint32_t *>(dst + (j * LG + base_c) * (VEC / 2)) = b0 | (b1 << 16);
                                } else {
                                    uint32_t bq[VEC];
#pragma unroll
                                    for (int q = 0; q < VEC; q++) bq[q] = q < left ? (uint32_t) posterior_bin(r_f[j * VEC + q], r_b[j * VEC + q], total, nb, &errbits) : 0u;
                                    *reinterpret_cast<uint2 *>(dst + (j * LG + base_c) * VEC) = make_uint2(bq[0] | (bq[VEC > 1 ? 1 : 0] << 16), bq[VEC > 2 ? 2 : 0] | (bq[VEC > 3 ? 3 : 0] << 16));
                                }
                            }
                        }
                    }
                }
            };
            ROLE_CLK_INIT();
            if (NGRP == 2) {
                /* barrier schedule: one after the prologue (step -1), one per column (steps 0 .. K - 1), one before the last merge
                 * list.  Group 0 acts at steps -1, 1, 3, ... (columns 0, 2, 4, ...), group 1 at steps 0, 2, ... */
                bins_load(grp);
                if (grp == 1) lds_barrier(); /* step -1 */
                for (int st = grp - 1; st < K; st += 2) {
                    bins_store(st + 1);
                    __builtin_amdgcn_sched_barrier(0); /* the new column's loads reuse the registers of the one just stored */
                    bins_load(st + 3);
                    ROLE_BARRIER();                 /* end of step st */
                    if (st + 1 < K) ROLE_BARRIER(); /* step st + 1: the other group's */
                }
            } else {
                /* one group: at step st the bins of column st + 1 (requested during step st - 1) are stored and column st + 2
                 * is requested: a column's f and b have one column time to arrive */
                bins_load(0);
                for (int st = -1; st < K; st++) {
                    bins_store(st + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    bins_load(st + 2);
                    ROLE_BARRIER(); /* end of step st */
                }
            }
            if (gidx == 0) ROLE_CLK_DONE(4 + grp);
            lds_barrier();
            if (PAIRS) lds_barrier();
        }
        __syncthreads(); /* also makes the lists above visible in global memory */

        /* (stRPHmm_pruneBackwards runs as a kernel of its own, mrp_prune_back_kernel: one wave per hmm instead of one wave of eight for a
         * quarter of this workgroup's life) */
        if (errbits) { atomicOr(sc.err, errbits); atomicOr(sc.err_hmm + hi_, errbits); }
        __syncthreads();
    }
}

/*
 * stRPHmm_pruneBackwards (hmm.c:1111-1158) of a level, one WAVE per hmm.  It used to be the tail of mrp_prune_kernel, run by the chain
 * wave while the workgroup's other seven waves -- and their registers, half a CU's -- waited: 1.3-1.9 M of a top-level workgroup's 7 M
 * cycles (`-DPRUNE_EXP_CLOCK`), a quarter of the slot-time of the kernel that holds most of it.  Here it holds 64 lanes and a few hundred
 * bytes of LDS per hmm (the kept flag per merge cell).  The forward pass's lists (sc.kept, kept_np, keptm, their counts) are complete when
 * this kernel starts: same stream, behind mrp_prune_kernel.
 */
template <bool PAIRS>
__global__ void __launch_bounds__(256) mrp_prune_back_kernel(const PruneHmm *__restrict__ hmms, int64_t n_hmms, PruneParams p, PruneScratch sc) {
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x / WAVE));
    const int S = p.S;
    const int n_flagw = ((p.max_merge + 127) >> 7) << 2; /* words, a multiple of four */
    uint32_t *flagw = lds + wave * n_flagw;              /* [max_merge bits] kept flag per merge cell, this wave's */
    auto flag_get = [&](uint32_t i) -> bool { return (flagw[i >> 5] >> (i & 31u)) & 1u; };
    auto flag_set = [&](uint32_t i) { atomicOr(&flagw[i >> 5], 1u << (i & 31u)); };
    auto flag_clr = [&](uint32_t i) { atomicAnd(&flagw[i >> 5], ~(1u << (i & 31u))); };
    for (int i = lane; i < n_flagw; i += WAVE) flagw[i] = 0u;
    wave_lds_fence();
    for (int64_t hi_ = (int64_t) blockIdx.x * 4 + wave; hi_ < n_hmms; hi_ += (int64_t) gridDim.x * 4) {
        const PruneHmm h = k_load(hmms + hi_);
        const int K = h.n_cols;
        /* ---- stRPHmm_pruneBackwards hmm.c:1111-1158: lists of at most S entries, one wave; the lists of
         * column k - 1 are requested before column k is worked on ---- */
        if constexpr (PAIRS) {
            /* The lists in UNITS: entry i = cells 2 i, 2 i + 1 of the kept list (the list stages wrote a unit's cells next to each
             * other; a column or merge column of one cell has a list of one -- an odd count), one entry per lane, the flags one
             * per merge unit.  The merge cells the surviving cells come from are all in the forward pass's kept merge list (it
             * keeps every merge cell a kept cell leads to, checked), so the flags set from the cells are the survivors of that list. */
            struct ListsU { int nk, nm; uint32_t cc; uint2 cn; uint32_t mm; };
            auto fetch = [&](int k) {
                ListsU L;
                L.nk = 0; L.nm = 0; L.cc = 0u; L.cn = make_uint2(0u, 0u); L.mm = 0u;
                if (k >= 0) {
                    const int64_t lc = h.col0 + k;
                    L.nk = sc.n_kept[lc];
                    L.nm = sc.n_keptm[lc];
                    if (2 * lane < S) { /* (the counts are not known yet when the lists are requested: every slot below S is read) */
                        L.cc = *reinterpret_cast<const uint32_t *>(sc.kept + lc * S + 2 * lane);
                        L.cn = *reinterpret_cast<const uint2 *>(sc.kept_np + lc * S + 2 * lane);
                        L.mm = *reinterpret_cast<const uint32_t *>(sc.keptm + lc * S + 2 * lane);
                    }
                }
                return L;
            };
            uint32_t pm = 0u; /* merge unit of the merge column after column k this lane has flagged */
            bool pmk = false;
            (void) fetch;
            /* The lists of four columns are in flight at a time, in a ring of registers loaded by inline asm and waited for by count (as the
             * recursion kernel's ring, mrp_kernels.hip): through the compiler's own waits -- the lists are loop-carried registers, so it
             * waits with vmcnt(0) at the loop head -- every column paid the whole trip to HBM / L2 of the lists asked for a column earlier
             * (1 450 cycles per column for some 500 of LDS flag operations and ballots).  Memory operations retire in issue order and
             * stores share the counter: a step issues exactly FIVE ring loads (unconditionally: beyond column 0 they read column 0 and
             * are ignored) plus stores that can only make a wait stricter; when a step starts, the two entries it needs are followed
             * by the loads of two younger entries: vmcnt(10). */
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            int32_t bnk0 = 0, bnk1 = 0, bnk2 = 0, bnk3 = 0, bnm0 = 0, bnm1 = 0, bnm2 = 0, bnm3 = 0;
            uint32_t bcc0 = 0u, bcc1 = 0u, bcc2 = 0u, bcc3 = 0u, bmm0 = 0u, bmm1 = 0u, bmm2 = 0u, bmm3 = 0u;
            u32x2 bcn0 = {0u, 0u}, bcn1 = {0u, 0u}, bcn2 = {0u, 0u}, bcn3 = {0u, 0u};
            const int lane2 = 2 * lane < S ? 2 * lane : 0; /* (lanes beyond the lists read entry 0 and never use it: lane < nku <= S / 2) */
#define BK_LOAD(i, k_)                                                                                                                  \
            {                                                                                                                           \
                const int kk_ = (k_) < 0 ? 0 : (k_);                                                                                    \
                const int64_t lc_ = h.col0 + kk_;                                                                                       \
                asm volatile("global_load_dword %0, %5, off\n\tglobal_load_dword %1, %6, off\n\tglobal_load_dword %2, %7, off\n\t"       \
                             "global_load_dwordx2 %3, %8, off\n\tglobal_load_dword %4, %9, off"                                        \
                             : "=&v"(bnk##i), "=&v"(bnm##i), "=&v"(bcc##i), "=&v"(bcn##i), "=&v"(bmm##i)                                \
                             : "v"(sc.n_kept + lc_), "v"(sc.n_keptm + lc_), "v"(sc.kept + lc_ * S + lane2), "v"(sc.kept_np + lc_ * S + lane2), \
                               "v"(sc.keptm + lc_ * S + lane2)                                                                          \
                             : "memory");                                                                                               \
            }
#ifdef MRP_SAFE_WAIT /* debugging aid: drain everything */
#define BK_WAIT(i, j) asm volatile("s_waitcnt vmcnt(0)" : "+v"(bnk##i), "+v"(bnm##i), "+v"(bcc##i), "+v"(bcn##i), "+v"(bmm##i), "+v"(bnk##j), "+v"(bnm##j), "+v"(bcc##j), "+v"(bcn##j), "+v"(bmm##j)::"memory");
#else
#define BK_WAIT(i, j) asm volatile("s_waitcnt vmcnt(10)" : "+v"(bnk##i), "+v"(bnm##i), "+v"(bcc##i), "+v"(bcn##i), "+v"(bmm##i), "+v"(bnk##j), "+v"(bnm##j), "+v"(bcc##j), "+v"(bcn##j), "+v"(bmm##j)::"memory");
#endif
            /* one column: entry i holds its lists, entry j those of the column before it; entry i is then asked for column k_ - 4 */
#define BK_STEP(i, j, k_)                                                                                                               \
            {                                                                                                                           \
                const int k = (k_);                                                                                                     \
                BK_WAIT(i, j)                                                                                                           \
                const int64_t lcol = h.col0 + k;                                                                                        \
                const int c_nk = bnk##i, c_nm = bnm##i, n_nm = k > 0 ? bnm##j : 0;                                                      \
                const uint32_t c_cc = bcc##i, c_cnx = bcn##i.x, c_cny = bcn##i.y, n_mm = bmm##j;                                        \
                const int sk = (c_nk & 1) ? 0 : 1, so = (c_nm & 1) ? 0 : 1; /* cells / merge cells after the column in pairs */          \
                const int nku = c_nk >> sk;                                                                                             \
                const bool keep = lane < nku && (k + 1 == K || flag_get((c_cnx & 0xFFFFu) >> so));                                      \
                const uint64_t m0 = __ballot(keep);                                                                                     \
                const int ns = __popcll(m0);                                                                                            \
                if (pmk) flag_clr(pm);                                                                                                  \
                pmk = false;                                                                                                            \
                if (ns != nku) {                                                                                                        \
                    if (keep) {                                                                                                         \
                        const int pos = mbcnt64(m0);                                                                                    \
                        *reinterpret_cast<uint32_t *>(sc.kept + lcol * S + 2 * pos) = c_cc;                                             \
                        *reinterpret_cast<uint2 *>(sc.kept_np + lcol * S + 2 * pos) = make_uint2(c_cnx, c_cny);                         \
                    }                                                                                                                   \
                    if (lane == 0) sc.n_kept[lcol] = ns << sk;                                                                          \
                }                                                                                                                       \
                if (k > 0) { /* merge column k - 1 keeps the merge cells some surviving cell comes from (:1141-1155) */                  \
                    const int si = (n_nm & 1) ? 0 : 1;                                                                                  \
                    if (keep) flag_set((c_cnx >> 16) >> si);                                                                            \
                    const int nmu = n_nm >> si;                                                                                         \
                    const uint32_t mu_ = (n_mm & 0xFFFFu) >> si;                                                                        \
                    const bool mk = lane < nmu && flag_get(mu_);                                                                        \
                    const uint64_t q0 = __ballot(mk);                                                                                   \
                    const int nms = __popcll(q0);                                                                                       \
                    if (nms != nmu) {                                                                                                   \
                        if (mk) *reinterpret_cast<uint32_t *>(sc.keptm + (lcol - 1) * S + 2 * mbcnt64(q0)) = n_mm;                      \
                        if (lane == 0) sc.n_keptm[lcol - 1] = nms << si;                                                                \
                    }                                                                                                                   \
                    pm = mu_; pmk = mk;                                                                                                 \
                }                                                                                                                       \
                BK_LOAD(i, k - 4)                                                                                                       \
            }
            BK_LOAD(0, K - 1) BK_LOAD(1, K - 2) BK_LOAD(2, K - 3) BK_LOAD(3, K - 4)
            for (int kq = K - 1; kq >= 0; kq -= 4) {
                BK_STEP(0, 1, kq)
                if (kq - 1 >= 0) BK_STEP(1, 2, kq - 1)
                if (kq - 2 >= 0) BK_STEP(2, 3, kq - 2)
                if (kq - 3 >= 0) BK_STEP(3, 0, kq - 3)
            }
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(bnk0), "+v"(bnk1), "+v"(bnk2), "+v"(bnk3), "+v"(bmm0), "+v"(bmm1), "+v"(bmm2), "+v"(bmm3)::"memory"); /* the ring drains before its registers are used again */
#undef BK_STEP
#undef BK_WAIT
#undef BK_LOAD
            if (pmk) flag_clr(pm);
        } else {
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
                    keep[u] = i < nk && (k + 1 == K || flag_get(cur.cn[u] & 0xFFFFu));
                }
                const uint64_t m0 = __ballot(keep[0]), m1 = __ballot(keep[1]);
                const int ns = __popcll(m0) + __popcll(m1);
                if (pmk[0]) flag_clr(pm[0]);
                if (pmk[1]) flag_clr(pm[1]);
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
                if (keep[0]) flag_set(cur.cn[0] >> 16);
                if (keep[1]) flag_set(cur.cn[1] >> 16);
                const int nmp = nx1.nm;
                bool mk[2];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int i = lane + u * WAVE;
                    mk[u] = i < nmp && flag_get(nx1.mm[u]);
                }
                const uint64_t q0 = __ballot(mk[0]), q1 = __ballot(mk[1]);
                const int nms = __popcll(q0) + __popcll(q1);
                if (nms != nmp) {
                    if (mk[0]) sc.keptm[(lcol - 1) * S + (int) lanemask_lt_count(q0, lane)] = (uint16_t) nx1.mm[0];
                    if (mk[1]) sc.keptm[(lcol - 1) * S + __popcll(q0) + (int) lanemask_lt_count(q1, lane)] = (uint16_t) nx1.mm[1];
                    if (lane == 0) sc.n_keptm[lcol - 1] = nms;
                }
                /* leave flagged exactly the surviving merge cells of column k - 1 */
                if (keep[0]) flag_clr(cur.cn[0] >> 16);
                if (keep[1]) flag_clr(cur.cn[1] >> 16);
                wave_lds_fence();
                if (mk[0]) flag_set(nx1.mm[0]);
                if (mk[1]) flag_set(nx1.mm[1]);
                pm[0] = nx1.mm[0]; pm[1] = nx1.mm[1];
                pmk[0] = mk[0]; pmk[1] = mk[1];
                cur = nx1;
                nx1 = nx2;
            }
            if (pmk[0]) flag_clr(pm[0]);
            if (pmk[1]) flag_clr(pm[1]);
        }
        wave_lds_fence(); /* (the flags are left clean for the wave's next hmm) */
    }
}

static size_t prune_lds_bytes(const PruneParams &p) {
    const size_t cap = p.pairs ? (size_t) (((p.max_cells + 1) / 2 + 3) & ~3) : (size_t) ((p.max_cells + 3) & ~3);
    const size_t dwords = 4 * PRUNE_SP + PRUNE_SP + 4 * PRUNE_SP + 64 + 2 * 1024 + 512 + 512 + 2 * PRUNE_SP + 2 * PRUNE_SP + 512 + 512 + 4 * PRUNE_TAB * 128 + 512;
    return dwords * 4 + (size_t) (((p.max_merge + 127) >> 7) << 2) * 4 + 2 * cap * 2 + 16;
}

/* columns of up to this many cells: one group of four bin-streaming waves, ten 16-byte loads per lane and array */
#define MRP_PRUNE_MID_CELLS (4 * WAVE * 10 * 4)

hipError_t mrp_launch_prune(const MrpBatchDev &d, const CrossCol *ccols_dev, const PruneHmm *hmms_dev, int64_t n_hmms, PruneParams p,
                            PruneScratch s, hipStream_t stream) {
    if (n_hmms <= 0) return hipSuccess;
    if (p.S > MRP_PRUNE_MAX_S || p.max_cells > MRP_PRUNE_MAX_CELLS || p.max_merge > MRP_PRUNE_MAX_CELLS || p.n_bins > 1024) return hipErrorInvalidValue;
    /* once per device (thread-safe: the concurrent halves of a call launch from two host threads) */
    static PerDeviceOnce once;
    const hipError_t configured = once.run([] {
        hipError_t e = hipSuccess;
        auto big_lds = [&](const void *f) { if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); };
        big_lds((const void *) mrp_prune_kernel<512, 32, 2, 1, false>); big_lds((const void *) mrp_prune_kernel<512, 32, 2, 1, true>);
        big_lds((const void *) mrp_prune_kernel<256, 4, 1, 1, false>); big_lds((const void *) mrp_prune_kernel<256, 4, 1, 1, true>);
        big_lds((const void *) mrp_prune_kernel<512, 10, 1, 4, false>); big_lds((const void *) mrp_prune_kernel<512, 10, 1, 4, true>);
        big_lds((const void *) mrp_prune_kernel<512, 10, 2, 4, false>); big_lds((const void *) mrp_prune_kernel<512, 10, 2, 4, true>);
        big_lds((const void *) mrp_prune_kernel<1024, 36, 2, 1, false>); big_lds((const void *) mrp_prune_kernel<1024, 36, 2, 1, true>);
        return e;
    });
    if (configured != hipSuccess) return configured;
    size_t lds = prune_lds_bytes(p);
    if (const char *pad = getenv("MRP_PRUNE_LDS_PAD_KB")) lds += (size_t) atol(pad) << 10; /* (development: fewer workgroups per CU) */
    if (lds > (size_t) MRP_LDS_BUDGET) return hipErrorInvalidValue;
    const dim3 grid((unsigned) (n_hmms < 65536 ? n_hmms : 65536));
    const PruneIn in{d.scols, ccols_dev, d.cell_f32, d.cell_b32, d.merge_f32, d.merge_b32, d.hmm_fb};
    /* The bin-streaming waves hold a column's f and b in registers: two groups of 2 waves x 32 cells per lane at 512 threads;
     * one group of 4 waves x 40 cells per lane (16-byte loads) at 512 threads for columns of up to 10 240 cells -- the
     * 100 x 100 cells of the shipped parameters: half the waves and 77 KB of LDS let two workgroups share a CU where the
     * 1 024-thread variant (two groups of 6 waves x 36 cells, up to 13 824 cells) fills it alone.
     * p.pairs (includeInvertedPartitions and even column limits): the chain runs on complement pairs. */
    const char *force = getenv("MRP_PRUNE_VARIANT"); /* development: "big" sends mid-sized columns to the 1 024-thread variant */
    const bool pairs = p.pairs != 0;
#define PRUNE_LAUNCH(T_, CPT_, NGRP_, VEC_)                                                                                               \
    do {                                                                                                                                  \
        if (pairs) hipLaunchKernelGGL((mrp_prune_kernel<T_, CPT_, NGRP_, VEC_, true>), grid, dim3(T_), lds, stream, in, hmms_dev, n_hmms, p, s);  \
        else hipLaunchKernelGGL((mrp_prune_kernel<T_, CPT_, NGRP_, VEC_, false>), grid, dim3(T_), lds, stream, in, hmms_dev, n_hmms, p, s);       \
    } while (0)
    /* what the bin-streaming waves hold per column: cells, or units when the level's arrays hold units (p.pairs == 2) */
    const int held = p.pairs == 2 ? (p.max_cells + 1) / 2 : p.max_cells;
    /* columns of at most 256 entries (the first merge levels: a few reads per hmm): four waves, the table wave writes the bins */
    if (held <= 4 * WAVE && !(force && force[0] == 's')) PRUNE_LAUNCH(256, 4, 1, 1);
    else if (held <= 2 * WAVE * 32) PRUNE_LAUNCH(512, 32, 2, 1);
    /* up to 5 120 entries -- the unit levels of the shipped parameters (100 x 100 cells = 5 000 units): the same four streaming waves as
     * two groups that alternate over the columns, so that a column's f and b have TWO column times to arrive (MRP_PRUNE_VARIANT=1: one group) */
    else if (held <= 2 * WAVE * 10 * 4 && !(force && (force[0] == 'b' || force[0] == '1'))) PRUNE_LAUNCH(512, 10, 2, 4);
    else if (held <= MRP_PRUNE_MID_CELLS && !(force && force[0] == 'b')) PRUNE_LAUNCH(512, 10, 1, 4);
    else PRUNE_LAUNCH(1024, 36, 2, 1);
#undef PRUNE_LAUNCH
    {
        const hipError_t fe = hipGetLastError();
        if (fe != hipSuccess) return fe;
    }
    {   /* the backward pass: one wave per hmm, four to a workgroup */
        const size_t back_lds = (size_t) 4 * ((size_t) (((p.max_merge + 127) >> 7) << 2)) * sizeof(uint32_t);
        const int64_t wgs = (n_hmms + 3) / 4;
        const dim3 bgrid((unsigned) (wgs < 65536 ? wgs : 65536));
        if (pairs) hipLaunchKernelGGL((mrp_prune_back_kernel<true>), bgrid, dim3(256), back_lds, stream, hmms_dev, n_hmms, p, s);
        else hipLaunchKernelGGL((mrp_prune_back_kernel<false>), bgrid, dim3(256), back_lds, stream, hmms_dev, n_hmms, p, s);
    }
    return hipGetLastError();
}

/* ------------------------------------------------------------------------------------------ */
/* small hmms: recursion, prune and compaction by ONE wave                                      */
/* ------------------------------------------------------------------------------------------ */
/*
 * The first merge levels (coordination.c:341-409: 2 x 2 -> 4, 4 x 4 -> 16 cells) hold a per cent of the cells and a
 * quarter of a batch's kernel time: tens of thousands of hmms of a handful of columns of a handful of cells, for which a
 * workgroup per hmm and direction (recursion), a four-wave role pipeline with a barrier per column (prune) and a wave per
 * column (compaction), each its own launch over its own descriptors, are almost pure fixed cost.  Here ONE wave takes an
 * hmm whose columns hold at most 64 units (MRP_XF_UNITS: complement pairs) and at most 64 merge units through
 *   stRPHmm_forward / stRPHmm_backward  hmm.c:827-929   lane = unit, the merge column in 64 words of LDS (ds_max)
 *   stRPHmm_pruneForwards               hmm.c:1049-1109 kept merge units as ONE 64-bit mask; a unit is linked iff its bit is
 *                                                        set; selection = the first n of the 64 sorted keys (bin, unit)
 *   stRPHmm_pruneBackwards              hmm.c:1111-1158 the same masks from the right
 *   filterMergeCells / relinkCells      hmm.c:964-1019  a merge cell's new index = the surviving merge cells below it
 * with no barrier and no second launch; what a pass leaves for the next (f, posterior bins, the kept units) goes through
 * the level's arrays in HBM / L2 (cell_f32, cell_b32 as the bins, the prune's kept list).  Same results as the three
 * kernels it replaces (the test suite compares every level's pruned hmms with the oracle's).  Cell-level transition indices
 * (which twin of a merge unit the even cell leads to) come from the parents' transitions, as in the prune chain.
 */
__global__ void __launch_bounds__(256) mrp_mini_kernel(MrpBatchDev d, const CrossCol *__restrict__ ccols, const PruneHmm *__restrict__ hmms,
                                                       int64_t n_hmms, int64_t hmm0, PruneParams p, PruneScratch sc) {
    __shared__ int32_t lds_m[4][2][WAVE];
    __shared__ uint32_t lds_flag[4][WAVE];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x / WAVE));
    const int S = p.S, nb = p.n_bins;
    const bool thr_all = p.thr_bin >= nb - 1;
    uint32_t *flag = lds_flag[wave];
    flag[lane] = 0u;
    /* which merge units some lane of `who` names: a scatter through 64 words of LDS (left clean) */
    auto units_named = [&](bool who, uint32_t unit) -> uint64_t {
        if (who) flag[unit & 63u] = 1u;
        wave_lds_fence();
        const uint64_t m = __ballot(flag[lane] != 0u);
        flag[lane] = 0u;
        wave_lds_fence();
        return m;
    };
    for (int64_t hi_ = (int64_t) blockIdx.x * 4 + wave; hi_ < n_hmms; hi_ += (int64_t) gridDim.x * 4) {
        const PruneHmm h = k_load(hmms + hi_);
        const int K = h.n_cols;
        int errbits = 0;
        int32_t *mcur = lds_m[wave][0], *mnxt = lds_m[wave][1];
        /* The columns' descriptors 64 at a time, one column per lane (first cell relative to the hmm's, units, and the bits of the
         * CrossCol the passes branch on); a pass gets a column's by v_readlane and asks for the NEXT column's data before it works
         * on the current one: a column step then waits for LDS, not for HBM. */
        const int64_t cell0 = k_load(d.scols + h.col0).cell_off;
        uint32_t t_off = 0u, t_nu = 0u, t_fl = 0u;
        auto load_block = [&](int k0) {
            const int k = k0 + lane;
            t_off = 0u; t_nu = 0u; t_fl = 0u;
            if (k < K) {
                const SweepCol *sc_ = d.scols + h.col0 + k;
                const CrossCol *cc_ = ccols + h.col0 + k;
                t_off = (uint32_t) (sc_->cell_off - cell0);
                t_nu = (uint32_t) sc_->n_cells | ((uint32_t) sc_->n_merge << 16);
                const bool paired = (cc_->a_part && cc_->d1 > 0) || (cc_->b_part && cc_->d2 > 0);
                t_fl = (paired ? 1u : 0u) | ((k + 1 < K && (cc_->flags & (MRP_XF_OUT_A_PAIRED | MRP_XF_OUT_B_PAIRED))) ? 2u : 0u) |
                       ((k > 0 && (cc_->flags & (MRP_XF_IN_A_PAIRED | MRP_XF_IN_B_PAIRED))) ? 4u : 0u);
            }
        };
        auto col_off = [&](int k, int k0) -> int64_t { return cell0 + (int64_t) (uint32_t) __builtin_amdgcn_readlane((int) t_off, k - k0); };
        auto col_nu = [&](int k, int k0) -> uint32_t { return (uint32_t) __builtin_amdgcn_readlane((int) t_nu, k - k0); };
        auto col_fl = [&](int k, int k0) -> uint32_t { return (uint32_t) __builtin_amdgcn_readlane((int) t_fl, k - k0); };
        {   /* every column within the single wave's reach */
            bool bad = false;
            for (int k0 = 0; k0 < K; k0 += WAVE) { load_block(k0); bad |= (t_nu & 0xFFFFu) > WAVE || (t_nu >> 16) > WAVE; }
            if (__any(bad)) { if (lane == 0) { atomicOr(sc.err, MRP_ENGINE_ERR_RANGE); atomicOr(sc.err_hmm + hmm0 + hi_, MRP_ENGINE_ERR_RANGE); } continue; }
        }
        struct ColData { uint32_t cost, np; int32_t f, bin; };
#ifdef PRUNE_EXP_CLOCK2 /* development: shader cycles of the four passes of the level's LONGEST hmm of this class (first record) */
        uint64_t mt_[5]; mt_[0] = __builtin_amdgcn_s_memtime();
#define MINI_T(i) mt_[i] = __builtin_amdgcn_s_memtime()
#else
#define MINI_T(i) do { } while (0)
#endif
        /* ---- stRPHmm_forward hmm.c:827-879 ---- */
        int32_t total = MRP_NEG_I32;
        for (int k0 = 0; k0 < K; k0 += WAVE) {
            load_block(k0);
            const int k1 = K < k0 + WAVE ? K : k0 + WAVE;
            auto ask = [&](int k) -> ColData {
                ColData r; r.f = 0; r.bin = 0;
                const uint32_t nu = col_nu(k, k0) & 0xFFFFu;
                const int64_t g = col_off(k, k0) + ((uint32_t) lane < nu ? lane : 0);
                r.cost = d.cell_cost[g]; r.np = d.cell_np[g];
                return r;
            };
            ColData nx = ask(k0);
            for (int k = k0; k < k1; k++) {
                const ColData cu = nx;
                if (k + 1 < k1) nx = ask(k + 1);
                const uint32_t nu = col_nu(k, k0) & 0xFFFFu;
                const bool act = (uint32_t) lane < nu;
                mnxt[lane] = MRP_NEG_I32;
                const int32_t mfp = k == 0 ? 0 : mcur[(cu.np >> 16) & 63u];
                const int32_t f = (k > 0 && mfp == MRP_NEG_I32) ? MRP_NEG_I32 : mfp - (int32_t) cu.cost; /* forwardCellCalc1 :791 */
                if (act) d.cell_f32[col_off(k, k0) + lane] = f;
                if (k + 1 < K) {
                    wave_lds_fence();
                    if (act && f != MRP_NEG_I32) atomicMax(&mnxt[cu.np & 63u], f);                      /* forwardCellCalc2 :814 */
                    wave_lds_fence();
                } else {
                    int32_t v = act ? f : MRP_NEG_I32;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) { const int32_t t = __shfl_xor(v, o, WAVE); v = t > v ? t : v; }
                    total = v;
                }
                int32_t *t_ = mcur; mcur = mnxt; mnxt = t_;
            }
        }
        MINI_T(1);
        /* ---- stRPHmm_backward hmm.c:910-929; a unit's posterior bin (total - f - b) replaces b in cell_b32 ---- */
        for (int k0 = ((K - 1) / WAVE) * WAVE; k0 >= 0; k0 -= WAVE) {
            load_block(k0);
            const int k1 = K < k0 + WAVE ? K : k0 + WAVE;
            auto ask = [&](int k) -> ColData {
                ColData r; r.bin = 0;
                const uint32_t nu = col_nu(k, k0) & 0xFFFFu;
                const int64_t g = col_off(k, k0) + ((uint32_t) lane < nu ? lane : 0);
                r.cost = d.cell_cost[g]; r.np = d.cell_np[g]; r.f = d.cell_f32[g];
                return r;
            };
            ColData nx = ask(k1 - 1);
            for (int k = k1 - 1; k >= k0; k--) {
                const ColData cu = nx;
                if (k > k0) nx = ask(k - 1);
                const uint32_t nu = col_nu(k, k0) & 0xFFFFu;
                const bool act = (uint32_t) lane < nu;
                mnxt[lane] = MRP_NEG_I32;
                const int32_t bv = k + 1 == K ? 0 : mcur[cu.np & 63u];
                const int bin = posterior_bin(cu.f, bv, total, nb, &errbits);
                if (act) d.cell_b32[col_off(k, k0) + lane] = bin;
                if (k > 0) {
                    wave_lds_fence();
                    if (act && bv != MRP_NEG_I32) atomicMax(&mnxt[(cu.np >> 16) & 63u], bv - (int32_t) cu.cost); /* backwardCellCalc :881 */
                    wave_lds_fence();
                }
                int32_t *t_ = mcur; mcur = mnxt; mnxt = t_;
            }
        }
        MINI_T(2);
        /* ---- stRPHmm_pruneForwards hmm.c:1049-1109 ---- */
        uint64_t keptM = 0ull; /* kept merge units of the merge column before column k */
        for (int k0 = 0; k0 < K; k0 += WAVE) {
            load_block(k0);
            const int k1 = K < k0 + WAVE ? K : k0 + WAVE;
            auto ask = [&](int k) -> ColData {
                ColData r; r.cost = 0u; r.f = 0;
                const uint32_t nu = col_nu(k, k0) & 0xFFFFu;
                const int64_t g = col_off(k, k0) + ((uint32_t) lane < nu ? lane : 0);
                r.np = d.cell_np[g]; r.bin = d.cell_b32[g];
                return r;
            };
            ColData nx = ask(k0);
            for (int k = k0; k < k1; k++) {
                const ColData cu = nx;
                if (k + 1 < k1) nx = ask(k + 1);
                const int64_t lcol = h.col0 + k;
                const uint32_t nu = col_nu(k, k0) & 0xFFFFu;
                const int w_sh = (int) (col_fl(k, k0) & 1u); /* cells per unit: 1 << w_sh */
                const bool act = (uint32_t) lane < nu;
                const bool linked = act && (k == 0 || ((keptM >> ((cu.np >> 16) & 63u)) & 1ull));
                uint32_t key[1] = {linked ? ((uint32_t) cu.bin << 14) | (uint32_t) lane : 0xFFFFFFFFu};
                const int L = __popcll(__ballot(linked));
                const int gp = thr_all ? L : __popcll(__ballot(linked && cu.bin <= p.thr_bin));
                const int n = kept_count(L << w_sh, gp << w_sh, p.min_p, p.max_p) >> w_sh;
                wave_bitonic_sort_n<1>(key, lane); /* stable descending posterior: smaller bin first, then list order (:1043, :1071) */
                const bool take = lane < n;
                const uint32_t ku = key[0] & 63u;
                const uint32_t np_k = (uint32_t) __builtin_amdgcn_ds_bpermute((int) (ku << 2), (int) cu.np);
                /* the kept units in posterior order, each with its transitions (units), for the pass from the right */
                if (take) *reinterpret_cast<uint2 *>(sc.kept_np + lcol * S + 2 * lane) = make_uint2(ku, np_k);
                if (lane == 0) sc.n_kept[lcol] = n;
                if (k + 1 < K) keptM = units_named(take, np_k & 0xFFFFu); /* getLinkedMergeCells :989: every merge cell a kept cell leads to */
            }
        }
        __threadfence_block();
        MINI_T(3);
        /* ---- stRPHmm_pruneBackwards hmm.c:1111-1158 + the pruned hmm in the resident layout; three columns in the pipe: the kept
         * list of column k - 2 and the parents' cells of the kept units of column k - 1 are in flight while column k is written ---- */
        struct Kept { int n; uint32_t ku, np; uint32_t dv; };
        struct Par { CrossCol cc; uint32_t rna, rnb, c1, c2; uint64_t pa, pb; };
        /* the kept units of a column and -- as a vector load, lane l = dword l: a scalar load would be waited for on the spot,
         * it shares its counter with the LDS traffic -- the column's CrossCol, both asked for two columns ahead */
        auto ask_kept = [&](int k) -> Kept {
            Kept r; r.n = 0; r.ku = 0u; r.np = 0u; r.dv = 0u;
            if (k >= 0) {
                const int64_t lcol = h.col0 + k;
                r.n = sc.n_kept[lcol];
                const uint2 v = 2 * lane < S ? *reinterpret_cast<const uint2 *>(sc.kept_np + lcol * S + 2 * lane) : make_uint2(0u, 0u);
                r.ku = v.x & 63u; r.np = v.y;
                if (lane < 16) r.dv = reinterpret_cast<const uint32_t *>(ccols + lcol)[lane];
            }
            return r;
        };
        auto fld = [&](uint32_t v, int l) -> uint32_t { return (uint32_t) __builtin_amdgcn_readlane((int) v, l); };
        auto ask_par = [&](int k, const Kept &kp) -> Par {
            Par r; r.cc = CrossCol{}; r.rna = r.rnb = r.c1 = r.c2 = 0u; r.pa = r.pb = 0ull;
            if (k >= 0) {
                const uint32_t dv = kp.dv;
                CrossCol c = {};
                c.a_part = reinterpret_cast<const uint64_t *>(((uint64_t) fld(dv, 1) << 32) | fld(dv, 0));
                c.b_part = reinterpret_cast<const uint64_t *>(((uint64_t) fld(dv, 3) << 32) | fld(dv, 2));
                c.a_np = reinterpret_cast<const uint32_t *>(((uint64_t) fld(dv, 5) << 32) | fld(dv, 4));
                c.b_np = reinterpret_cast<const uint32_t *>(((uint64_t) fld(dv, 7) << 32) | fld(dv, 6));
                const uint32_t w10 = fld(dv, 10), w11 = fld(dv, 11), w12 = fld(dv, 12), w13 = fld(dv, 13), w14 = fld(dv, 14);
                c.C1 = (uint16_t) w10; c.C2 = (uint16_t) (w10 >> 16); c.Ma = (uint16_t) w11; c.Mb = (uint16_t) (w11 >> 16);
                c.Pa = (uint16_t) w12; c.Pb = (uint16_t) (w12 >> 16);
                c.d1 = (uint8_t) w13; c.d2 = (uint8_t) (w13 >> 8); c.out_a = (uint8_t) (w13 >> 16); c.out_b = (uint8_t) (w13 >> 24);
                c.in_a = (uint8_t) w14; c.in_b = (uint8_t) (w14 >> 8); c.flags = (uint8_t) (w14 >> 16);
                r.cc = c;
                const bool a_cp = r.cc.a_part && r.cc.d1 > 0, b_cp = r.cc.b_part && r.cc.d2 > 0;
                /* the unit's even cell (c1, c2): the order rule of cross_cell */
                if (a_cp) { const uint32_t C2 = r.cc.C2 ? r.cc.C2 : 1u; const uint32_t q = (uint32_t) (((float) kp.ku + 0.5f) * __builtin_amdgcn_rcpf((float) C2)); r.c1 = 2u * q; r.c2 = kp.ku - q * C2; }
                else if (b_cp) { r.c1 = 0u; r.c2 = 2u * kp.ku; }
                const bool in = lane < kp.n;
                if (in && r.cc.a_np) { r.rna = r.cc.a_np[r.c1 & 127u]; r.pa = r.cc.a_part[r.c1 & 127u]; }
                if (in && r.cc.b_np) { r.rnb = r.cc.b_np[r.c2 & 127u]; r.pb = r.cc.b_part[r.c2 & 127u]; }
            }
            return r;
        };
        uint64_t aliveM = 0ull; /* surviving merge units of the merge column after column k */
        Kept kp0 = ask_kept(K - 1);
        Par pr0 = ask_par(K - 1, kp0);
        Kept kp1 = ask_kept(K - 2);
        for (int k = K - 1; k >= 0; k--) {
            const Kept kp = kp0;
            const Par pr = pr0;
            kp0 = kp1;
            pr0 = ask_par(k - 1, kp0);
            kp1 = ask_kept(k - 2);
            const CrossCol &cc = pr.cc;
            const bool a_cp = cc.a_part && cc.d1 > 0, b_cp = cc.b_part && cc.d2 > 0;
            const int w_sh = (a_cp || b_cp) ? 1 : 0;
            const uint32_t o_pm = (k + 1 < K && (cc.flags & (MRP_XF_OUT_A_PAIRED | MRP_XF_OUT_B_PAIRED))) ? 1u : 0u;
            const uint32_t i_pm = (k > 0 && (cc.flags & (MRP_XF_IN_A_PAIRED | MRP_XF_IN_B_PAIRED))) ? 1u : 0u;
            const bool in = lane < kp.n;
            const bool keep = in && (k + 1 == K || ((aliveM >> (kp.np & 63u)) & 1ull));
            const uint64_t km = __ballot(keep);
            const int ns = __popcll(km), pos = mbcnt64(km);
            const uint64_t aliveP = k > 0 ? units_named(keep, kp.np >> 16) : 0ull; /* :1141-1155 */
            if (keep) {
                /* the even cell's transitions as CELL indices (which twin of the merge unit it leads to / comes from) */
                uint32_t nxt = 0u, prv = 0u;
                if (k + 1 < K) {
                    const uint32_t ii = cc.out_a == MRP_CONN_REAL ? (pr.rna & 0xFFFFu) : (cc.out_a == MRP_CONN_IDENT ? pr.c1 : 0u);
                    const uint32_t jj = cc.out_b == MRP_CONN_REAL ? (pr.rnb & 0xFFFFu) : (cc.out_b == MRP_CONN_IDENT ? pr.c2 : 0u);
                    nxt = pair_index(ii, jj, cc.Mb, true, (cc.flags & MRP_XF_OUT_A_PAIRED) != 0, (cc.flags & MRP_XF_OUT_B_PAIRED) != 0);
                }
                if (k > 0) {
                    const uint32_t ii = cc.in_a == MRP_CONN_REAL ? (pr.rna >> 16) : (cc.in_a == MRP_CONN_IDENT ? pr.c1 : 0u);
                    const uint32_t jj = cc.in_b == MRP_CONN_REAL ? (pr.rnb >> 16) : (cc.in_b == MRP_CONN_IDENT ? pr.c2 : 0u);
                    prv = pair_index(ii, jj, cc.Pb, true, (cc.flags & MRP_XF_IN_A_PAIRED) != 0, (cc.flags & MRP_XF_IN_B_PAIRED) != 0);
                }
                if ((nxt >> o_pm) != (kp.np & 0xFFFFu) || (prv >> i_pm) != (kp.np >> 16)) errbits |= MRP_ENGINE_ERR_RANGE; /* (the cross product kernel's view) */
                /* filterMergeCells keeps the merge cells in their original relative order: new index = survivors below */
                const uint32_t mu = nxt >> o_pm, pmu = prv >> i_pm;
                const uint32_t new_next = k + 1 < K ? ((uint32_t) __popcll(aliveM & ((1ull << (mu & 63u)) - 1ull)) << o_pm) + (nxt & o_pm) : 0u;
                const uint32_t new_prev = k > 0 ? ((uint32_t) __popcll(aliveP & ((1ull << (pmu & 63u)) - 1ull)) << i_pm) + (prv & i_pm) : 0u;
                const uint64_t part = cc.d1 < 64 ? (pr.pa | (pr.pb << cc.d1)) : pr.pa; /* mergePartitionsOrMasks partitions.c:21-28 */
                if (w_sh) {
                    const int64_t o = (int64_t) k * S + 2 * pos;
                    h.out_part[o] = part;
                    h.out_part[o + 1] = ~part & accept_mask((uint32_t) cc.d1 + (uint32_t) cc.d2);
                    h.out_np[o] = new_next | (new_prev << 16);
                    h.out_np[o + 1] = (new_next ^ o_pm) | ((new_prev ^ i_pm) << 16);
                } else {
                    h.out_part[(int64_t) k * S + pos] = part;
                    h.out_np[(int64_t) k * S + pos] = new_next | (new_prev << 16);
                }
            }
            if (lane == 0) {
                h.out_n_cells[k] = ns << w_sh;
                h.out_n_merge[k] = k + 1 < K ? __popcll(aliveM) << o_pm : 0;
            }
            aliveM = aliveP;
        }
        MINI_T(4);
#ifdef PRUNE_EXP_CLOCK2
        if (hi_ == 0 && lane == 0) {
            for (int i_ = 0; i_ < 4; i_++) atomicAdd((unsigned long long *) (sc.err + 4) + i_, (unsigned long long) (mt_[i_ + 1] - mt_[i_]));
            atomicAdd((unsigned long long *) (sc.err + 4) + 8, (unsigned long long) K);
        }
#endif
        if (p.pad && hmm0 + hi_ == 0) errbits |= MRP_ENGINE_ERR_MERGE; /* fault injection of the tests (mrp_context_set_test_hooks bit 0) */
        if (errbits) { atomicOr(sc.err, errbits); atomicOr(sc.err_hmm + hmm0 + hi_, errbits); }
    }
}

hipError_t mrp_launch_mini(const MrpBatchDev &d, const CrossCol *ccols_dev, const PruneHmm *hmms_dev, int64_t n_hmms, int64_t hmm0, PruneParams p,
                           PruneScratch s, hipStream_t stream) {
    if (n_hmms <= 0) return hipSuccess;
    if (p.pairs != 2 || p.n_bins > 1024) return hipErrorInvalidValue;
    const int64_t wgs = (n_hmms + 3) / 4;
    hipLaunchKernelGGL(mrp_mini_kernel, dim3((unsigned) (wgs < (1 << 20) ? wgs : (1 << 20))), dim3(256), 0, stream, d, ccols_dev, hmms_dev, n_hmms, hmm0, p, s);
    return hipGetLastError();
}

/* ------------------------------------------------------------------------------------------ */
/* compaction: the pruned hmm in the resident layout                                           */
/* ------------------------------------------------------------------------------------------ */
__global__ void __launch_bounds__(256) mrp_compact_kernel(MrpBatchDev d, const CrossCol *__restrict__ ccols, const PruneHmm *__restrict__ hmms,
                                                          const int32_t *__restrict__ col_hmm, int64_t n_cols, int32_t n_hmms_here, PruneParams p,
                                                          PruneScratch sc) {
    __shared__ uint32_t km_all[4][2 * PRUNE_SP]; /* per wave: the kept merge cells after and before the column (ascending) */
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x / WAVE));
    uint32_t *km_next = km_all[wave], *km_prev = km_all[wave] + PRUNE_SP;
    const int S = p.S;
    const int64_t stride = (int64_t) gridDim.x * (blockDim.x / WAVE);
    /* number of entries below x in an ascending list of n <= 128 entries */
    auto rank_of = [&](const uint32_t *list, int n, uint32_t x) -> uint32_t {
        int lo = 0, hi = n; /* first index with list[index] >= x */
#pragma unroll
        for (int step = 0; step < 8; step++) {
            if (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (list[mid] < x) lo = mid + 1; else hi = mid;
            }
        }
        return (uint32_t) lo;
    };
    for (int64_t lcol = (int64_t) blockIdx.x * (blockDim.x / WAVE) + wave; lcol < n_cols; lcol += stride) {
        if (col_hmm[lcol] >= n_hmms_here) continue; /* (an hmm of the single-wave kernel: it wrote its pruned hmm itself) */
        const PruneHmm h = k_load(hmms + col_hmm[lcol]);
        const int k = (int) (lcol - h.col0);
        const int K = h.n_cols;
        const SweepCol col = k_load(d.scols + lcol);
        /* the level wrote no partitions (cross product and emission in one pass): a kept cell's partition comes from its
         * two parent cells */
        CrossCol cc = {};
        if (!d.partition) cc = k_load(ccols + lcol);
        const bool inv = (cc.flags & MRP_XF_INVERTED) != 0;
        const bool a_cells_paired = inv && cc.a_part && cc.d1 > 0, b_cells_paired = inv && cc.b_part && cc.d2 > 0;
        const int nk = sc.n_kept[lcol];
        const int nm = k + 1 < K ? sc.n_keptm[lcol] : 0;
        const int nmp = k > 0 ? sc.n_keptm[lcol - 1] : 0;
        /* filterMergeCells keeps the merge cells in their original relative order: a merge cell's new index is the number of
         * kept merge cells before it, i.e. its position in the (ascending) kept list */
        {   /* the kept lists are in posterior order (hmm.c:1090); sorted by merge cell index here, 128 keys per wave in registers */
            uint32_t a0 = lane < nm ? (uint32_t) sc.keptm[lcol * S + lane] : 0xFFFF0000u + (uint32_t) lane;
            uint32_t a1 = lane + WAVE < nm ? (uint32_t) sc.keptm[lcol * S + lane + WAVE] : 0xFFFF0040u + (uint32_t) lane;
            uint32_t b0 = lane < nmp ? (uint32_t) sc.keptm[(lcol - 1) * S + lane] : 0xFFFF0000u + (uint32_t) lane;
            uint32_t b1 = lane + WAVE < nmp ? (uint32_t) sc.keptm[(lcol - 1) * S + lane + WAVE] : 0xFFFF0040u + (uint32_t) lane;
            if (nm > 1) wave_bitonic_sort128(a0, a1, lane);
            if (nmp > 1) wave_bitonic_sort128(b0, b1, lane);
            km_next[lane] = a0; km_next[lane + WAVE] = a1;
            km_prev[lane] = b0; km_prev[lane + WAVE] = b1;
        }
        wave_lds_fence();
        for (int i = lane; i < nk; i += WAVE) {
            const uint32_t c = sc.kept[lcol * S + i];
            const uint32_t np = sc.kept_np[lcol * S + i];
            const uint32_t nx = np & 0xFFFFu, pv = np >> 16;
            const uint32_t new_next = rank_of(km_next, nm, nx), new_prev = rank_of(km_prev, nmp, pv);
            uint64_t part;
            if (d.partition) part = d.partition[col.cell_off + c];
            else {
                uint32_t c1, c2;
                cross_cell(c, cc.C2, inv, a_cells_paired, b_cells_paired, c1, c2);
                part = cross_partition(cc, c1, c2);
            }
            h.out_part[(int64_t) k * S + i] = part;
            h.out_np[(int64_t) k * S + i] = new_next | (new_prev << 16);
        }
        if (lane == 0) {
            h.out_n_cells[k] = nk;
            h.out_n_merge[k] = nm;
        }
        wave_lds_fence(); /* the lists are rewritten for the wave's next column */
    }
}

/* ------------------------------------------------------------------------------------------ */
/* trace back                                                                                  */
/* ------------------------------------------------------------------------------------------ */
__global__ void __launch_bounds__(64) mrp_traceback_kernel(MrpBatchDev d, const PruneHmm *__restrict__ hmms, int64_t n_hmms,
                                                           int32_t *__restrict__ err, int32_t *__restrict__ err_hmm) {
    /* The chosen cells of the last TB_TILE columns: the walk itself neither stores to global memory nor gathers the chosen cell's
     * partition (a load whose address is the step's own result: with it in the loop every step waited a memory latency for the one
     * before); both happen a tile at a time, a column per lane, coalesced. */
    constexpr int TB_TILE = 1024;
    __shared__ int32_t chosen[TB_TILE];
    const int lane = threadIdx.x;
    for (int64_t hi = blockIdx.x; hi < n_hmms; hi += gridDim.x) {
        const PruneHmm h = k_load(hmms + hi);
        const int K = h.n_cols;
        auto flush = [&](int k_lo, int k_hi) { /* columns [k_lo, k_hi): chosen[k - k_lo] -> out_n_cells, out_part */
            wave_lds_fence();
            for (int k = k_lo + lane; k < k_hi; k += WAVE) {
                const int32_t bi = chosen[k - k_lo];
                h.out_n_cells[k] = bi;
                h.out_part[k] = d.partition[d.scols[h.col0 + k].cell_off + bi];
            }
            wave_lds_fence();
        };
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
                    if (lane == 0) { atomicOr(err, MRP_ENGINE_ERR_RANGE); atomicOr(err_hmm + hi, MRP_ENGINE_ERR_RANGE); }
                    best_i = 0;
                }
            }
            if (lane == 0) chosen[k & (TB_TILE - 1)] = best_i;
            if (small) {
                const uint32_t from = best_i < WAVE ? cur.np[0] : cur.np[1];
                want = (uint32_t) __shfl((int) from, best_i & (WAVE - 1), WAVE) >> 16;
            } else {
                want = d.cell_np[col.cell_off + best_i] >> 16;
            }
            if ((k & (TB_TILE - 1)) == 0) flush(k, min(K, k + TB_TILE)); /* (the tile [k, k + TB_TILE) is complete: the walk goes downwards) */
            col = col_m1;
            col_m1 = col_m2;
            cur = nxt;
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* genome fragments                                                                            */
/* ------------------------------------------------------------------------------------------ */
/*
 * stGenomeFragment_construct (genomeFragment.c:40-69), fillInPredictedGenome (emissions.c:246-343),
 * stGenomeFragment_refineGenomeFragment (genomeFragment.c:165-232) and the re-adding of the coverage-filtered reads
 * (bubbleGraph.c:2772-2779) for the final hmm of every chunk, one workgroup per chunk, right behind the trace back: through round 3
 * this was a quarter of the host's CPU time (a flat copy of the final hmm rebuilt per chunk, then byte loops over columns x
 * reads and reads x sites, up to ten rounds).  Everything is integer arithmetic on profile bytes and the site tables (the three
 * "probabilities" are -(float) of an integer), so the results are the host's bit for bit; what needs care is ORDER:
 *   reads1 / reads2      first sighting along the path (column, then slot) per set -- a read that inconsistent columns put on both
 *                        sides is in both; a round keeps the stayers in order and appends the arrivals in the other list's order.
 *                        First sightings are counted per column and prefix-summed; moves are stable compactions (block scans).
 *   a column names its reads by the offsets of their profile bytes (read_byte_off); the read index comes from a binary search in
 *   the chunk's reads sorted by pool offset.
 */
#define FRAG_T 1024
#define FRAG_LDS_READS 6144 /* pool offsets of up to this many reads of a chunk are searched in LDS */
static __device__ int frag_block_excl_scan(int v, int *total, int *lds /* [FRAG_T / 64 + 1] */) {
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    const int incl = wave_incl_scan_bc(v);
    if (lane == WAVE - 1) lds[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < FRAG_T / WAVE; w++) { const int x = lds[w]; if (w < wave) base += x; tot += x; }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}
/* emissions.c:263-321 for the sites of one column with the given partition */
static __device__ void frag_fill_column(const DevChunk &ch, const DevCol &c, const int64_t *rbo, uint64_t partition, FragSite *out, int32_t frag_start) {
    const uint32_t first_allele = ch.allele_offset[c.site_start];
    const uint32_t sup1 = (uint32_t) __popcll(partition), sup2 = (uint32_t) c.depth - sup1;
    for (int s_ = 0; s_ < c.n_sites; s_++) {
        const int site = c.site_start + s_;
        const uint32_t A = ch.allele_number[site], so = ch.allele_offset[site] - first_allele;
        const uint16_t *sub = ch.sub + ch.sub_offset[site], *prior = ch.prior + ch.allele_offset[site];
        uint32_t h1[MRP_MAX_ALLELES], h2[MRP_MAX_ALLELES];
        const uint32_t An = A <= MRP_MAX_ALLELES ? A : MRP_MAX_ALLELES;
        for (uint32_t a = 0; a < An; a++) { h1[a] = 0u; h2[a] = 0u; }
        for (int i = 0; i < c.depth; i++) { /* alleleLogHapProbabilities :144-154: byte sums over the reads of either side */
            const uint8_t *b = ch.pool + rbo[i] + so;
            const bool in1 = (partition >> i) & 1ull;
            for (uint32_t a = 0; a < An; a++) { const uint32_t v = b[a]; if (in1) h1[a] += v; else h2[a] += v; }
        }
        uint32_t best = 0xFFFFFFFFu, anc = 0u; /* ancestorHapProbabilities :156-172, the ml ancestor allele :283-292 */
        for (uint32_t i = 0; i < An; i++) {
            uint32_t x = h1[0] + sub[i * A], y = h2[0] + sub[i * A];
            for (uint32_t q = 1; q < An; q++) { x = min(x, h1[q] + sub[i * A + q]); y = min(y, h2[q] + sub[i * A + q]); }
            const uint32_t j = x + y + prior[i];
            if (j < best) { best = j; anc = i; }
        }
        uint32_t hap1 = 0u, hap2 = 0u, m1 = h1[0] + sub[anc * A], m2 = h2[0] + sub[anc * A]; /* getMLAllele :246-261 */
        for (uint32_t i = 1; i < An; i++) {
            if (h1[i] + sub[anc * A + i] < m1) { m1 = h1[i] + sub[anc * A + i]; hap1 = i; }
            if (h2[i] + sub[anc * A + i] < m2) { m2 = h2[i] + sub[anc * A + i]; hap2 = i; }
        }
        FragSite o;
        o.ancestor = (uint8_t) anc; o.hap1 = (uint8_t) hap1; o.hap2 = (uint8_t) hap2; o.support1 = (uint8_t) sup1; o.support2 = (uint8_t) sup2;
        o.pad[0] = o.pad[1] = o.pad[2] = 0;
        o.genotype_prob = -((float) best); o.hap_prob1 = -(float) h1[hap1]; o.hap_prob2 = -(float) h2[hap2];
        out[site - frag_start] = o;
    }
}
/* getLogProbOfReadGivenHaplotype genomeFragment.c:71-89 for both haplotypes: the byte sums (the log probabilities are minus these
 * over 30; comparing the sums the other way round compares the doubles) */
static __device__ void frag_read_sums(const DevChunk &ch, const FragRead &r, const FragSite *sites, int32_t start, int32_t length, uint32_t *s1, uint32_t *s2) {
    int lo = start - r.ref_start, hi = start + length - r.ref_start;
    if (lo < 0) lo = 0;
    if (hi > r.length) hi = r.length;
    const uint8_t *pool = ch.pool + r.pool_offset;
    const uint32_t *ao = ch.allele_offset + r.ref_start;
    const uint32_t first = ao[0];
    const FragSite *st = sites + (r.ref_start - start);
    uint32_t a = 0u, b = 0u;
    for (int i = lo; i < hi; i++) {
        const uint32_t o = ao[i] - first;
        a += pool[o + st[i].hap1];
        b += pool[o + st[i].hap2];
    }
    *s1 = a; *s2 = b;
}
__global__ void __launch_bounds__(FRAG_T) mrp_fragment_kernel(MrpBatchDev d, const PruneHmm *__restrict__ hmms, int64_t n_hmms, FragArrays fa,
                                                              int32_t *__restrict__ err, int32_t *__restrict__ err_hmm) {
    __shared__ int scan_lds[FRAG_T / WAVE + 1];
    __shared__ int moved[2];
    __shared__ int64_t pool_sorted[FRAG_LDS_READS]; /* the reads' pool offsets in ascending order */
    const int tid = threadIdx.x;
    for (int64_t hi_ = blockIdx.x; hi_ < n_hmms; hi_ += gridDim.x) {
        const PruneHmm h = hmms[hi_];
        const FragHmm f = fa.hmms[hi_];
        const int K = h.n_cols, nr = f.n_reads, cap = 2 * nr + 2;
        const DevCol c0 = d.cols[h.col0];
        const DevChunk ch = d.chunks[c0.chunk];
        const FragRead *reads = fa.reads + f.reads0;
        const int32_t *by_pool = fa.by_pool + f.reads0;
        FragSite *sites = fa.sites + f.site0;
        int32_t *l1 = fa.lists + 2 * f.list0, *l2 = l1 + cap, *w1 = fa.work + 2 * f.list0, *w2 = w1 + cap;
        uint32_t *key1 = fa.read_key + 2 * f.reads0, *key2 = key1 + nr;
        uint64_t *part = fa.col_part + h.col0;
        int32_t *cnt1 = fa.col_cnt + 2 * h.col0, *cnt2 = cnt1 + K;
        int bad = 0;
        const bool in_lds = nr <= FRAG_LDS_READS;
        for (int r = tid; r < nr; r += FRAG_T) { key1[r] = 0xFFFFFFFFu; key2[r] = 0xFFFFFFFFu; if (in_lds) pool_sorted[r] = reads[by_pool[r]].pool_offset; }
        __syncthreads();
        /* the reads of every column (binary search of the byte offset among the chunk's reads), the path's partitions, and for
         * every read where it is first seen on either side */
        for (int k = tid; k < K; k += FRAG_T) {
            const DevCol c = d.cols[h.col0 + k];
            const uint64_t p = h.out_part[k];
            part[k] = p;
            const int64_t *rbo = d.read_byte_off + c.read_off;
            const int64_t shift = (int64_t) ch.allele_offset[c.site_start];
            for (int i = 0; i < c.depth; i++) {
                const int64_t v = rbo[i]; /* = pool_offset + (allele_offset[site_start] - allele_offset[read start]) */
                int lo = 0, hi = nr; /* last read (by pool offset) that starts at or before v */
                if (in_lds) { while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (pool_sorted[mid] <= v) lo = mid; else hi = mid; } }
                else { while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (reads[by_pool[mid]].pool_offset <= v) lo = mid; else hi = mid; } }
                const int rid = nr > 0 ? by_pool[lo] : 0;
                const FragRead r = reads[rid];
                if (nr <= 0 || r.pool_offset + (shift - (int64_t) ch.allele_offset[r.ref_start]) != v || c.site_start < r.ref_start || c.site_start >= r.ref_start + r.length) bad = 1;
                fa.col_read[c.read_off + i] = rid;
                atomicMin(((p >> i) & 1ull) ? &key1[rid] : &key2[rid], ((uint32_t) k << 6) | (uint32_t) i);
            }
        }
        __syncthreads();
        /* reads1 / reads2 in order of first sighting (genomeFragment.c:40-48, hmm.c:221-248): per column the first sightings ... */
        for (int k = tid; k < K; k += FRAG_T) {
            const DevCol c = d.cols[h.col0 + k];
            const uint64_t p = part[k];
            int a = 0, b = 0;
            for (int i = 0; i < c.depth; i++) {
                const int rid = fa.col_read[c.read_off + i];
                const uint32_t key = ((uint32_t) k << 6) | (uint32_t) i;
                if ((p >> i) & 1ull) a += key1[rid] == key; else b += key2[rid] == key;
            }
            cnt1[k] = a; cnt2[k] = b;
        }
        __syncthreads();
        int n1 = 0, n2 = 0;
        {   /* ... prefix sums over the columns (a thread owns a run of consecutive columns), then the lists */
            const int per = (K + FRAG_T - 1) / FRAG_T, k0 = tid * per, k1 = k0 + per < K ? k0 + per : K;
            int a = 0, b = 0;
            for (int k = k0; k < k1; k++) { a += cnt1[k]; b += cnt2[k]; }
            int base1 = frag_block_excl_scan(a, &n1, scan_lds), base2 = frag_block_excl_scan(b, &n2, scan_lds);
            for (int k = k0; k < k1; k++) {
                const DevCol c = d.cols[h.col0 + k];
                const uint64_t p = part[k];
                for (int i = 0; i < c.depth; i++) {
                    const int rid = fa.col_read[c.read_off + i];
                    const uint32_t key = ((uint32_t) k << 6) | (uint32_t) i;
                    if ((p >> i) & 1ull) { if (key1[rid] == key) l1[base1++] = rid; }
                    else if (key2[rid] == key) l2[base2++] = rid;
                }
            }
        }
        /* stGenomeFragment_construct: the predicted genome of every column */
        for (int k = tid; k < K; k += FRAG_T) {
            const DevCol c = d.cols[h.col0 + k];
            frag_fill_column(ch, c, d.read_byte_off + c.read_off, part[k], sites, f.ref_start);
        }
        __syncthreads();
        /* stGenomeFragment_refineGenomeFragment genomeFragment.c:165-232: key1 / key2 now hold the move flags of a round */
        uint32_t *m12 = key1, *m21 = key2;
        for (int it = 0; it < f.max_iterations; it++) {
            for (int r = tid; r < nr; r += FRAG_T) { m12[r] = 0u; m21[r] = 0u; }
            if (tid < 2) moved[tid] = 0;
            __syncthreads();
            for (int i = tid; i < n1 + n2; i += FRAG_T) { /* :126-151: reads more probably generated by the other haplotype */
                const bool first_list = i < n1;
                const int rid = first_list ? l1[i] : l2[i - n1];
                uint32_t s1, s2;
                frag_read_sums(ch, reads[rid], sites, f.ref_start, f.length, &s1, &s2);
                if (first_list ? s1 > s2 : s2 > s1) { (first_list ? m12 : m21)[rid] = 1u; atomicAdd(&moved[first_list ? 0 : 1], 1); }
            }
            __syncthreads();
            if (moved[0] + moved[1] == 0) break;
            {   /* the stayers in order, then the arrivals in the other list's order (:199-203): four stable compactions */
                const int tot = n1 + n2, per = (tot + FRAG_T - 1) / FRAG_T, i0 = tid * per, i1 = i0 + per < tot ? i0 + per : tot;
                int stay1 = 0, stay2 = 0, go12 = 0, go21 = 0;
                for (int i = i0; i < i1; i++) {
                    if (i < n1) { if (m12[l1[i]]) go12++; else stay1++; }
                    else { if (m21[l2[i - n1]]) go21++; else stay2++; }
                }
                int t_s1, t_s2, t_12, t_21;
                int b_s1 = frag_block_excl_scan(stay1, &t_s1, scan_lds), b_s2 = frag_block_excl_scan(stay2, &t_s2, scan_lds);
                int b_12 = frag_block_excl_scan(go12, &t_12, scan_lds), b_21 = frag_block_excl_scan(go21, &t_21, scan_lds);
                for (int i = i0; i < i1; i++) {
                    if (i < n1) { const int rid = l1[i]; if (m12[rid]) w2[t_s2 + b_12++] = rid; else w1[b_s1++] = rid; }
                    else { const int rid = l2[i - n1]; if (m21[rid]) w1[t_s1 + b_21++] = rid; else w2[b_s2++] = rid; }
                }
                n1 = t_s1 + t_21; n2 = t_s2 + t_12;
                int32_t *t_ = l1; l1 = w1; w1 = t_; t_ = l2; l2 = w2; w2 = t_;
            }
            __syncthreads();
            for (int k = tid; k < K; k += FRAG_T) { /* :211-226: the moved reads change sides in every column that holds them */
                const DevCol c = d.cols[h.col0 + k];
                uint64_t flip = 0ull;
                for (int i = 0; i < c.depth; i++) { const int rid = fa.col_read[c.read_off + i]; flip |= (uint64_t) (m12[rid] ^ m21[rid]) << i; }
                if (!flip) continue;
                part[k] ^= flip;
                frag_fill_column(ch, c, d.read_byte_off + c.read_off, part[k], sites, f.ref_start);
            }
            __syncthreads();
        }
        __syncthreads();
        /* bubbleGraph.c:2772-2779: the reads the coverage filter took out go to the haplotype that explains them better, in order */
        if (tid == 0) {
            const int32_t *disc = fa.discarded + f.disc0;
            for (int i = 0; i < f.n_discarded; i++) {
                uint32_t s1, s2;
                frag_read_sums(ch, reads[disc[i]], sites, f.ref_start, f.length, &s1, &s2);
                if (s1 > s2) l2[n2++] = disc[i]; else l1[n1++] = disc[i];
            }
            fa.counts[2 * hi_] = n1; fa.counts[2 * hi_ + 1] = n2;
            /* results at fixed places: reads1 in the first half of the chunk's block of fa.lists, reads2 in the second */
            int32_t *o1 = fa.lists + 2 * f.list0, *o2 = o1 + cap;
            if (l1 != o1) { /* (an odd number of rounds moved something: the lists sit in the work halves) */
                for (int i = 0; i < n1; i++) o1[i] = l1[i];
                for (int i = 0; i < n2; i++) o2[i] = l2[i];
            }
        }
        if (__syncthreads_or(bad)) { if (tid == 0) { atomicOr(err, MRP_ENGINE_ERR_RANGE); atomicOr(err_hmm + hi_, MRP_ENGINE_ERR_RANGE); } }
    }
}

hipError_t mrp_launch_fragments(const MrpBatchDev &d, const PruneHmm *hmms_dev, int64_t n_hmms, FragArrays fa, int32_t *err, int32_t *err_hmm,
                                hipStream_t stream) {
    if (n_hmms <= 0) return hipSuccess;
    hipLaunchKernelGGL(mrp_fragment_kernel, dim3((unsigned) (n_hmms < 65536 ? n_hmms : 65536)), dim3(FRAG_T), 0, stream, d, hmms_dev, n_hmms, fa, err, err_hmm);
    return hipGetLastError();
}

hipError_t mrp_launch_traceback(const MrpBatchDev &d, const PruneHmm *hmms_dev, int64_t n_hmms, int32_t *err, int32_t *err_hmm,
                                hipStream_t stream) {
    if (n_hmms <= 0) return hipSuccess;
    hipLaunchKernelGGL(mrp_traceback_kernel, dim3((unsigned) (n_hmms < 65536 ? n_hmms : 65536)), dim3(64), 0, stream, d, hmms_dev,
                       n_hmms, err, err_hmm);
    return hipGetLastError();
}

hipError_t mrp_launch_compact(const MrpBatchDev &d, const CrossCol *ccols_dev, const PruneHmm *hmms_dev, const int32_t *col_hmm_dev, int64_t n_cols,
                              int64_t n_hmms_here, PruneParams p, PruneScratch s, hipStream_t stream) {
    if (n_cols <= 0 || n_hmms_here <= 0) return hipSuccess;
    int64_t wgs = (n_cols + 3) / 4;
    static const long cg = getenv("MRP_COMPACT_GRID") ? atol(getenv("MRP_COMPACT_GRID")) : 0; /* (development) */
    if (cg > 0 && wgs > cg) wgs = cg;
    hipLaunchKernelGGL(mrp_compact_kernel, dim3((unsigned) (wgs < 65536 ? wgs : 65536)), dim3(256), 0, stream, d, ccols_dev, hmms_dev,
                       col_hmm_dev, n_cols, (int32_t) n_hmms_here, p, s);
    return hipGetLastError();
}
