/*
 * mrp_kernels.hip -- gfx950 (CDNA4, wave64) kernels of the stRPHmm forward/backward sweep.
 *
 *  mrp_planes_kernel      calculateCountBitVectors (emissions.c:91-123): one wave per column,
 *                         lane = read, the 64-bit plane word is the wave ballot itself.
 *  mrp_emission_kernel    emissionLogProbability (emissions.c:221-240) for every cell of the batch:
 *                         independent of the recursion, so it runs chip-wide; one wave per tile of
 *                         a column, popcount bit-plane sums with the planes in scalar registers.
 *  mrp_sweep_i32_kernel   the recursion of stRPHmm_forwardBackward (hmm.c:931-942) in max-plus mode
 *                         (maxNotSumTransitions, every shipped config): one persistent workgroup
 *                         walks the columns of one HMM; merge-cell arrays live in LDS as int32 and
 *                         are combined with ds_max; the next column's cells are fetched while the
 *                         current one is processed.
 *  mrp_sweep_f64_kernel   the same recursion in fp64 for log-sum-exp mode (hmm.c:19,
 *                         stMath_logAddExact) and for HMMs too large for the int32/LDS path.
 *
 * Arithmetic identities used (all exact in integers):
 *   hap2 cost of allele a  = sum_{i not in P} prob_i[a] = total[a] - sum_{i in P} prob_i[a]
 *       (emissions.c:200 evaluates the same sum with ~partition against zero-padded planes)
 *   column total (max mode) = max_c (f_c + b_c) = max_m (mf_m + mb_m) over the following merge
 *       column, because b_c = mb[next(c)] and mf_m = max_{c -> m} f_c (hmm.c:887-907).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mrp_device.h"
#include "mrp_kernels.h"
#include "../../include/margin_rphmm.h"

#define WAVE 64

/* Read-only data that is uniform across a wave (column descriptors, bit planes, site tables) is
 * loaded through the constant address space so that hipcc emits s_load (scalar cache, SGPR
 * operands) instead of 64 identical vector loads.  None of it is written by the kernel reading it. */
#define K_AS(T) const __attribute__((address_space(4))) T *
#define K_PTR(T, p) ((K_AS(T)) (p))
/* whole-struct load through the scalar path (dword copies; SROA turns them into s_load_dwordxN) */
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

static __device__ __forceinline__ double i32_to_log(int32_t v) {
    return v == MRP_NEG_I32 ? -__builtin_inf() : (double) v;
}
static __device__ __forceinline__ int32_t log_to_i32(double v) {
    return v == -__builtin_inf() ? MRP_NEG_I32 : (int32_t) v;
}
static __device__ __forceinline__ int32_t add_i32(int32_t a, int32_t b) {
    return a == MRP_NEG_I32 ? MRP_NEG_I32 : a + b;
}
static __device__ __forceinline__ int32_t wave_max_i32(int32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        int32_t t = __shfl_xor(v, o, WAVE);
        v = t > v ? t : v;
    }
    return v;
}

/* ------------------------------------------------------------------------------------------ */
/* bit planes                                                                                  */
/* ------------------------------------------------------------------------------------------ */
__global__ void __launch_bounds__(256) mrp_planes_kernel(const DevCol *__restrict__ cols, int64_t n_cols,
                                                         const DevChunk *__restrict__ chunks,
                                                         const int64_t *__restrict__ read_byte_off,
                                                         uint64_t *__restrict__ planes,
                                                         uint32_t *__restrict__ slot_total) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int64_t col = (int64_t) blockIdx.x * (blockDim.x / WAVE) + (threadIdx.x / WAVE);
    if (col >= n_cols) return;
    const DevCol c = cols[col];
    const uint8_t *__restrict__ pool = chunks[c.chunk].pool;
    const bool active = lane < c.depth;
    const int64_t off = active ? read_byte_off[c.read_off + lane] : 0;
    for (int s = 0; s < c.n_slots; s++) {
        const uint32_t byte = active ? pool[off + s] : 0u;
        uint64_t mine = 0;
        uint32_t total = 0;
#pragma unroll
        for (int b = 0; b < MRP_ALLELE_LOG_PROB_BITS; b++) {
            const uint64_t v = __ballot((byte >> b) & 1u);
            if (lane == b) mine = v;
            total += (uint32_t) __popcll(v) << b;
        }
        if (lane < MRP_ALLELE_LOG_PROB_BITS) planes[(c.slot_off + s) * MRP_ALLELE_LOG_PROB_BITS + lane] = mine;
        if (lane == 0) slot_total[c.slot_off + s] = total;
    }
}

hipError_t mrp_launch_planes(const MrpBatchDev &d, hipStream_t stream) {
    if (d.n_cols == 0) return hipSuccess;
    const int waves = 4;
    const int64_t grid = (d.n_cols + waves - 1) / waves;
    hipLaunchKernelGGL(mrp_planes_kernel, dim3((unsigned) grid), dim3(waves * WAVE), 0, stream, d.cols, d.n_cols,
                       d.chunks, d.read_byte_off, d.planes, d.slot_total);
    return hipGetLastError();
}

/* ------------------------------------------------------------------------------------------ */
/* emissions (emissions.c:125-240)                                                             */
/* ------------------------------------------------------------------------------------------ */
/* getLogProbOfAllele: sum_b popcount(plane_b & P) << b.  NARROW: depth <= 32, high dword is 0. */
template <bool NARROW>
static __device__ __forceinline__ uint32_t allele_cost(const uint64_t (&w)[8], uint64_t P) {
    uint32_t r = 0;
    const uint32_t plo = (uint32_t) P, phi = (uint32_t) (P >> 32);
#pragma unroll
    for (int b = 0; b < 8; b++) {
        uint32_t c = (uint32_t) __popc((uint32_t) w[b] & plo);
        if (!NARROW) c += (uint32_t) __popc((uint32_t) (w[b] >> 32) & phi);
        r += c << b;
    }
    return r;
}

/* cost of CPT partitions over one column, no ancestor substitution model (emissions.c:205-207) */
template <int CPT, bool NARROW>
static __device__ __forceinline__ void column_cost_plain(const DevCol &c, K_AS(uint32_t) allele_number,
                                                         K_AS(uint64_t) planes, K_AS(uint32_t) slot_total,
                                                         const uint64_t (&P)[CPT], uint32_t (&cost)[CPT]) {
    int64_t slot = c.slot_off;
#pragma unroll
    for (int j = 0; j < CPT; j++) cost[j] = 0;
    for (int s = 0; s < c.n_sites; s++) {
        const uint32_t A = allele_number[c.site_start + s];
        uint32_t m1[CPT], m2[CPT];
#pragma unroll
        for (int j = 0; j < CPT; j++) { m1[j] = 0xFFFFFFFFu; m2[j] = 0xFFFFFFFFu; }
        for (uint32_t a = 0; a < A; a++) {
            K_AS(uint64_t) pl = planes + (slot + a) * 8;
            uint64_t w[8];
#pragma unroll
            for (int b = 0; b < 8; b++) w[b] = pl[b];
            const uint32_t t = slot_total[slot + a];
#pragma unroll
            for (int j = 0; j < CPT; j++) {
                const uint32_t lp = allele_cost<NARROW>(w, P[j]);
                m1[j] = min(m1[j], lp);
                m2[j] = min(m2[j], t - lp);
            }
        }
#pragma unroll
        for (int j = 0; j < CPT; j++) cost[j] += m1[j] + m2[j];
        slot += A;
    }
}

/* one partition, ancestor substitution model on (emissions.c:209-218); final sweeps only */
static __device__ uint32_t column_cost_ancestor(const DevCol &c, const DevChunk &ch, K_AS(uint64_t) planes,
                                                K_AS(uint32_t) slot_total, uint64_t P) {
    int64_t slot = c.slot_off;
    uint32_t cost = 0;
    for (int s = 0; s < c.n_sites; s++) {
        const int site = c.site_start + s;
        const uint32_t A = K_PTR(uint32_t, ch.allele_number)[site];
        K_AS(uint16_t) sub = K_PTR(uint16_t, ch.sub) + K_PTR(uint32_t, ch.sub_offset)[site];
        K_AS(uint16_t) prior = K_PTR(uint16_t, ch.prior) + K_PTR(uint32_t, ch.allele_offset)[site];
        uint32_t h1[MRP_MAX_ALLELES], h2[MRP_MAX_ALLELES];
        for (uint32_t a = 0; a < A; a++) {
            K_AS(uint64_t) pl = planes + (slot + a) * 8;
            uint64_t w[8];
#pragma unroll
            for (int b = 0; b < 8; b++) w[b] = pl[b];
            const uint32_t lp = allele_cost<false>(w, P);
            h1[a] = lp;
            h2[a] = slot_total[slot + a] - lp;
        }
        uint32_t g = 0xFFFFFFFFu;
        for (uint32_t i = 0; i < A; i++) {
            uint32_t a1 = 0xFFFFFFFFu, a2 = 0xFFFFFFFFu;
            for (uint32_t k = 0; k < A; k++) {
                const uint32_t sc = sub[i * A + k];
                a1 = min(a1, h1[k] + sc);
                a2 = min(a2, h2[k] + sc);
            }
            g = min(g, a1 + a2 + (uint32_t) prior[i]);
        }
        cost += g;
        slot += A;
    }
    return cost;
}

/* ------------------------------------------------------------------------------------------ */
/* emission kernel: every cell of every column of every HMM of the batch, fully parallel       */
/* ------------------------------------------------------------------------------------------ */
/* One wave per tile of up to 64*EMIT_CPT consecutive cells of ONE column, so the column's bit
 * planes are wave-uniform (scalar loads) and the partition loads are coalesced. */
#define EMIT_CPT 4
#define EMIT_TILE (WAVE * EMIT_CPT)
__global__ void __launch_bounds__(256) mrp_emission_kernel(MrpBatchDev d, const int2 *__restrict__ tiles,
                                                           int64_t n_tiles) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int64_t tile = (int64_t) blockIdx.x * (blockDim.x / WAVE) + (threadIdx.x / WAVE);
    if (tile >= n_tiles) return;
    const int col_index = K_PTR(int32_t, tiles)[2 * tile];
    const int start = K_PTR(int32_t, tiles)[2 * tile + 1];
    const DevCol c = k_load(d.cols + col_index);
    const DevChunk ch = k_load(d.chunks + c.chunk);
    K_AS(uint64_t) planes = K_PTR(uint64_t, d.planes);
    K_AS(uint32_t) slot_total = K_PTR(uint32_t, d.slot_total);
    uint64_t P[EMIT_CPT];
    uint32_t cost[EMIT_CPT];
#pragma unroll
    for (int j = 0; j < EMIT_CPT; j++) {
        const int idx = start + j * WAVE + lane;
        P[j] = idx < c.n_cells ? d.partition[c.cell_off + idx] : 0ull;
    }
    if (c.flags & MRP_FLAG_INCLUDE_ANCESTOR_SUB_PROB) {
#pragma unroll
        for (int j = 0; j < EMIT_CPT; j++) cost[j] = column_cost_ancestor(c, ch, planes, slot_total, P[j]);
    } else if (c.depth <= 32) {
        column_cost_plain<EMIT_CPT, true>(c, K_PTR(uint32_t, ch.allele_number), planes, slot_total, P, cost);
    } else {
        column_cost_plain<EMIT_CPT, false>(c, K_PTR(uint32_t, ch.allele_number), planes, slot_total, P, cost);
    }
#pragma unroll
    for (int j = 0; j < EMIT_CPT; j++) {
        const int idx = start + j * WAVE + lane;
        if (idx < c.n_cells) d.cell_cost[c.cell_off + idx] = cost[j];
    }
}

hipError_t mrp_launch_emission(const MrpBatchDev &d, const int2 *tiles_dev, int64_t n_tiles, hipStream_t stream) {
    if (n_tiles == 0) return hipSuccess;
    const int waves = 4;
    hipLaunchKernelGGL(mrp_emission_kernel, dim3((unsigned) ((n_tiles + waves - 1) / waves)), dim3(waves * WAVE), 0,
                       stream, d, tiles_dev, n_tiles);
    return hipGetLastError();
}

/* ------------------------------------------------------------------------------------------ */
/* max-plus recursion, int32 merge arrays in LDS                                               */
/* ------------------------------------------------------------------------------------------ */
/* per-thread registers for one column's share of cells: emission cost, next / previous merge index */
template <int CPT>
struct CellRegs {
    uint32_t cost[CPT], nxt[CPT], prv[CPT];
};
template <int CPT>
static __device__ __forceinline__ void load_cells(CellRegs<CPT> &r, const MrpBatchDev &d, int64_t cell_off,
                                                  int n_cells, int tid, int T) {
#pragma unroll
    for (int j = 0; j < CPT; j++) {
        const int idx = j * T + tid;
        if (idx < n_cells) {
            const int64_t g = cell_off + idx;
            r.cost[j] = d.cell_cost[g];
            r.nxt[j] = d.cell_next[g];
            r.prv[j] = d.cell_prev[g];
        }
    }
}

template <int CPT>
__global__ void __launch_bounds__(1024)
mrp_sweep_i32_kernel(MrpBatchDev d, const int32_t *__restrict__ order, int max_merge) {
    extern __shared__ __attribute__((aligned(16))) int32_t lds[];
    int32_t *cur = lds;                 /* merge column read by the column being processed */
    int32_t *nxt = lds + max_merge;     /* merge column being accumulated */
    int32_t *red = lds + 2 * max_merge; /* [0] hmm forward, [1] hmm backward, [2] column total */

    const int tid = threadIdx.x, T = blockDim.x;
    const int64_t hmm_index = K_PTR(int32_t, order)[blockIdx.x];
    const DevHmm h = k_load(d.hmms + hmm_index);
    const DevCol *cols = d.cols + h.col0;
    const int K = h.n_cols;

    for (int i = tid; i < 2 * max_merge; i += T) lds[i] = MRP_NEG_I32;
    if (tid < 4) red[tid] = MRP_NEG_I32;

    /* ---------------- forward (hmm.c:827-879) ---------------- */
    DevCol c = k_load(cols);
    CellRegs<CPT> ra, rb;
    load_cells<CPT>(ra, d, c.cell_off, c.n_cells, tid, T);
    __syncthreads();
    for (int k = 0; k < K; k++) {
        const bool first = (k == 0), last = (k == K - 1);
        /* the next column's descriptor and cells do not depend on the recursion: fetch them now */
        DevCol cn = c;
        if (!last) {
            cn = k_load(cols + k + 1);
            load_cells<CPT>(rb, d, cn.cell_off, cn.n_cells, tid, T);
        }
        int32_t local_max = MRP_NEG_I32;
#pragma unroll
        for (int j = 0; j < CPT; j++) {
            const int idx = j * T + tid;
            if (idx < c.n_cells) {
                const int32_t fp = first ? 0 : cur[ra.prv[j]];
                const int32_t fv = add_i32(fp, -(int32_t) ra.cost[j]);    /* forwardCellCalc1, hmm.c:791-812 */
                d.cell_f[c.cell_off + idx] = i32_to_log(fv);
                if (!last) atomicMax(&nxt[ra.nxt[j]], fv);                /* forwardCellCalc2, hmm.c:814-825 */
                else local_max = max(local_max, fv);
            }
        }
        for (int idx = CPT * T + tid; idx < c.n_cells; idx += T) {        /* columns wider than CPT*T */
            const int64_t g = c.cell_off + idx;
            const int32_t fp = first ? 0 : cur[d.cell_prev[g]];
            const int32_t fv = add_i32(fp, -(int32_t) d.cell_cost[g]);
            d.cell_f[g] = i32_to_log(fv);
            if (!last) atomicMax(&nxt[d.cell_next[g]], fv);
            else local_max = max(local_max, fv);
        }
        if (last) {
            local_max = wave_max_i32(local_max);
            if ((tid & (WAVE - 1)) == 0) atomicMax(&red[0], local_max);
        }
        __syncthreads();
        if (!last) {
            for (int m = tid; m < c.n_merge; m += T) d.merge_f[c.mcell_off + m] = i32_to_log(nxt[m]);
            const int n_clear = (k + 2 < K) ? cn.n_merge : 0;
            for (int m = tid; m < n_clear; m += T) cur[m] = MRP_NEG_I32;
        }
        __syncthreads();
        int32_t *t = cur; cur = nxt; nxt = t;
        c = cn;
        ra = rb;
    }
    const int32_t hmm_forward = red[0];

    /* ---------------- backward (hmm.c:910-929) ---------------- */
    /* c is the last column.  cur = mb of the merge column after column k (read), nxt = mb of the
     * merge column before it (accumulated). */
    DevCol pc = c;
    if (K >= 2) {
        pc = k_load(cols + K - 2);
        for (int m = tid; m < pc.n_merge; m += T) nxt[m] = MRP_NEG_I32;
    }
    load_cells<CPT>(ra, d, c.cell_off, c.n_cells, tid, T);
    __syncthreads();
    for (int k = K - 1; k >= 0; k--) {
        const bool first = (k == 0), last = (k == K - 1);
        /* pc = column k-1 (already loaded); fetch its cells and the descriptor of column k-2 */
        DevCol ppc = pc;
        if (!first) {
            load_cells<CPT>(rb, d, pc.cell_off, pc.n_cells, tid, T);
            if (k >= 2) ppc = k_load(cols + k - 2);
        }
        int32_t local_max = MRP_NEG_I32;
#pragma unroll
        for (int j = 0; j < CPT; j++) {
            const int idx = j * T + tid;
            if (idx < c.n_cells) {
                const int32_t bv = last ? 0 : cur[ra.nxt[j]];             /* backwardCellCalc, hmm.c:881-908 */
                d.cell_b[c.cell_off + idx] = i32_to_log(bv);
                const int32_t p = add_i32(bv, -(int32_t) ra.cost[j]);
                if (!first) atomicMax(&nxt[ra.prv[j]], p);
                else local_max = max(local_max, p);
            }
        }
        for (int idx = CPT * T + tid; idx < c.n_cells; idx += T) {
            const int64_t g = c.cell_off + idx;
            const int32_t bv = last ? 0 : cur[d.cell_next[g]];
            d.cell_b[g] = i32_to_log(bv);
            const int32_t p = add_i32(bv, -(int32_t) d.cell_cost[g]);
            if (!first) atomicMax(&nxt[d.cell_prev[g]], p);
            else local_max = max(local_max, p);
        }
        if (first) {
            local_max = wave_max_i32(local_max);
            if ((tid & (WAVE - 1)) == 0) atomicMax(&red[1], local_max);
        }
        __syncthreads();
        if (!first) {
            int32_t tot = MRP_NEG_I32;
            for (int m = tid; m < pc.n_merge; m += T) {
                const int32_t mbv = nxt[m];
                d.merge_b[pc.mcell_off + m] = i32_to_log(mbv);
                const int32_t mfv = log_to_i32(d.merge_f[pc.mcell_off + m]);
                if (mbv != MRP_NEG_I32 && mfv != MRP_NEG_I32) tot = max(tot, mbv + mfv);
            }
            tot = wave_max_i32(tot);
            if ((tid & (WAVE - 1)) == 0) atomicMax(&red[2], tot);
            const int n_clear = (k >= 2) ? ppc.n_merge : 0;
            for (int m = tid; m < n_clear; m += T) cur[m] = MRP_NEG_I32;
        }
        __syncthreads();
        if (!first && tid == 0) {
            d.col_total[h.col0 + k - 1] = i32_to_log(red[2]);
            red[2] = MRP_NEG_I32;
        }
        int32_t *t = cur; cur = nxt; nxt = t;
        c = pc;
        pc = ppc;
        ra = rb;
    }
    if (tid == 0) {
        d.col_total[h.col0 + K - 1] = i32_to_log(hmm_forward);
        d.hmm_fb[2 * hmm_index] = i32_to_log(hmm_forward);
        d.hmm_fb[2 * hmm_index + 1] = i32_to_log(red[1]);
    }
}

hipError_t mrp_launch_sweep_i32(const MrpBatchDev &d, const int32_t *order_dev, int64_t n, int block_threads,
                                int max_merge, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    if (max_merge < 1) max_merge = 1;
    const size_t lds = (size_t) (2 * max_merge + 4) * sizeof(int32_t);
    auto k = mrp_sweep_i32_kernel<4>;
    hipError_t e = hipFuncSetAttribute((const void *) k, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3((unsigned) n), dim3(block_threads), lds, stream, d, order_dev, max_merge);
    return hipGetLastError();
}

/* ------------------------------------------------------------------------------------------ */
/* generic fp64 sweep: log-sum-exp mode and oversize HMMs                                      */
/* ------------------------------------------------------------------------------------------ */
static __device__ __forceinline__ double log_add_exact(double x, double y) {
    /* stMath_logAddExact (sonLib; call site hmm.c:19) */
    if (x == -__builtin_inf()) return y;
    if (y == -__builtin_inf()) return x;
    return x > y ? x + log(1.0 + exp(y - x)) : y + log(1.0 + exp(x - y));
}
static __device__ __forceinline__ double log_add_p(double a, double b, bool max_not_sum) { /* hmm.c:15-20 */
    return max_not_sum ? (a > b ? a : b) : log_add_exact(a, b);
}
static __device__ __forceinline__ double load_agent(double *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
static __device__ void atomic_log_add_p(double *addr, double v, bool max_not_sum) {
    if (v == -__builtin_inf()) return;
    unsigned long long *a = (unsigned long long *) addr;
    unsigned long long old = __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (true) {
        const double cur = __longlong_as_double((long long) old);
        const double nv = log_add_p(cur, v, max_not_sum);
        const unsigned long long nvb = (unsigned long long) __double_as_longlong(nv);
        if (nvb == old) return;
        const unsigned long long seen = atomicCAS(a, old, nvb);
        if (seen == old) return;
        old = seen;
    }
}
static __device__ __forceinline__ double wave_log_add_p(double v, bool max_not_sum) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double t = __shfl_xor(v, o, WAVE);
        v = log_add_p(v, t, max_not_sum);
    }
    return v;
}

__global__ void __launch_bounds__(1024) mrp_sweep_f64_kernel(MrpBatchDev d, const int32_t *__restrict__ order) {
    const int tid = threadIdx.x, T = blockDim.x;
    const int64_t hmm_index = K_PTR(int32_t, order)[blockIdx.x];
    const DevHmm h = k_load(d.hmms + hmm_index);
    const DevCol *cols = d.cols + h.col0;
    const int K = h.n_cols;
    const bool max_not_sum = (h.flags & MRP_FLAG_MAX_NOT_SUM) != 0;
    const double NEG = -__builtin_inf();
    double *hmm_f = d.hmm_fb + 2 * hmm_index, *hmm_b = hmm_f + 1;
    /* merge_f / merge_b / col_total / hmm_fb were filled with -inf before the launch (hmm.c:752-789) */

    for (int k = 0; k < K; k++) {
        const DevCol c = k_load(cols + k);
        const bool first = (k == 0), last = (k == K - 1);
        const DevCol pc = k_load(cols + (first ? k : k - 1));
        double local = NEG;
        for (int idx = tid; idx < c.n_cells; idx += T) {
            const int64_t g = c.cell_off + idx;
            const double e = -((double) d.cell_cost[g]);                         /* emissions.c:239 */
            double fv = first ? 0.0 : load_agent(&d.merge_f[pc.mcell_off + d.cell_prev[g]]);
            fv += e;
            d.cell_f[g] = fv;
            if (!last) atomic_log_add_p(&d.merge_f[c.mcell_off + d.cell_next[g]], fv, max_not_sum);
            else local = log_add_p(local, fv, max_not_sum);
        }
        if (last) {
            local = wave_log_add_p(local, max_not_sum);
            if ((tid & (WAVE - 1)) == 0) atomic_log_add_p(hmm_f, local, max_not_sum);
        }
        __syncthreads();
    }
    for (int k = K - 1; k >= 0; k--) {
        const DevCol c = k_load(cols + k);
        const bool first = (k == 0), last = (k == K - 1);
        const DevCol pc = k_load(cols + (first ? k : k - 1));
        double local_b = NEG, local_t = NEG;
        for (int idx = tid; idx < c.n_cells; idx += T) {
            const int64_t g = c.cell_off + idx;
            double p = -((double) d.cell_cost[g]);
            double bv = 0.0;
            if (!last) { bv = load_agent(&d.merge_b[c.mcell_off + d.cell_next[g]]); p += bv; }
            d.cell_b[g] = bv;
            if (!first) atomic_log_add_p(&d.merge_b[pc.mcell_off + d.cell_prev[g]], p, max_not_sum);
            else local_b = log_add_p(local_b, p, max_not_sum);
            local_t = log_add_p(local_t, d.cell_f[g] + bv, max_not_sum);  /* hmm.c:906-907 */
        }
        if (first) {
            local_b = wave_log_add_p(local_b, max_not_sum);
            if ((tid & (WAVE - 1)) == 0) atomic_log_add_p(hmm_b, local_b, max_not_sum);
        }
        local_t = wave_log_add_p(local_t, max_not_sum);
        if ((tid & (WAVE - 1)) == 0) atomic_log_add_p(&d.col_total[h.col0 + k], local_t, max_not_sum);
        __syncthreads();
    }
}

hipError_t mrp_launch_sweep_f64(const MrpBatchDev &d, const int32_t *order_dev, int64_t n, int block_threads,
                                hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(mrp_sweep_f64_kernel, dim3((unsigned) n), dim3(block_threads), 0, stream, d, order_dev);
    return hipGetLastError();
}

__global__ void mrp_fill_f64_kernel(double *p, int64_t n, double v) {
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t) gridDim.x * blockDim.x)
        p[i] = v;
}
hipError_t mrp_launch_fill_f64(double *p, int64_t n, double v, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    int64_t grid = (n + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(mrp_fill_f64_kernel, dim3((unsigned) grid), dim3(256), 0, stream, p, n, v);
    return hipGetLastError();
}

/* emissionLogProbability for a list of partitions of one column (unit-test seam) */
__global__ void mrp_emissions_kernel(const DevCol *__restrict__ col, const DevChunk *__restrict__ chunks,
                                     const uint64_t *__restrict__ planes, const uint32_t *__restrict__ slot_total,
                                     uint32_t flags, int64_t n_cells, const uint64_t *__restrict__ partitions,
                                     double *__restrict__ out) {
    const DevCol c = k_load(col);
    const DevChunk ch = k_load(chunks + c.chunk);
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_cells) return;
    const uint64_t P[1] = {partitions[i]};
    uint32_t cost[1];
    if (flags & MRP_FLAG_INCLUDE_ANCESTOR_SUB_PROB)
        cost[0] = column_cost_ancestor(c, ch, K_PTR(uint64_t, planes), K_PTR(uint32_t, slot_total), P[0]);
    else
        column_cost_plain<1, false>(c, K_PTR(uint32_t, ch.allele_number), K_PTR(uint64_t, planes),
                                    K_PTR(uint32_t, slot_total), P, cost);
    out[i] = -((double) cost[0]);
}
hipError_t mrp_launch_emissions(const DevCol *col_dev, const DevChunk *chunks, const uint64_t *planes,
                                const uint32_t *slot_total, uint32_t flags, int64_t n_cells,
                                const uint64_t *partitions, double *out, hipStream_t stream) {
    if (n_cells <= 0) return hipSuccess;
    hipLaunchKernelGGL(mrp_emissions_kernel, dim3((unsigned) ((n_cells + 255) / 256)), dim3(256), 0, stream, col_dev,
                       chunks, planes, slot_total, flags, n_cells, partitions, out);
    return hipGetLastError();
}
