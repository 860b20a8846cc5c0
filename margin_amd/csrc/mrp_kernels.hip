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
#include "mrp_internal.h"
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
static __device__ __forceinline__ int32_t add_i32(int32_t a, int32_t b) {
    return a == MRP_NEG_I32 ? MRP_NEG_I32 : a + b;
}
/* Workgroup barrier that orders LDS traffic only.  __syncthreads() also emits s_waitcnt vmcnt(0),
 * which would drain the prefetched global loads (and wait for every store) at each column. */
static __device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
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
__global__ void __launch_bounds__(256) mrp_planes_kernel(const PlaneCol *__restrict__ pcols_all,
                                                         const int32_t *__restrict__ list, int64_t n_cols,
                                                         const int64_t *__restrict__ read_byte_off,
                                                         uint64_t *__restrict__ planes,
                                                         uint32_t *__restrict__ slot_total,
                                                         uint32_t *__restrict__ slot_bytes, int filter) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x / WAVE));
    /* Persistent waves striding over the columns.  Per column there are three dependent memory levels
     * (descriptor -> per-read byte offset -> profile bytes); the first two of the NEXT column are
     * requested while the current column is processed, and the bytes of up to PLANE_U allele slots
     * are requested together. */
    const int64_t wave_stride = (int64_t) gridDim.x * (blockDim.x / WAVE);
    int64_t col = (int64_t) blockIdx.x * (blockDim.x / WAVE) + wave;
    if (col >= n_cols) return;
    /* list == NULL: every column; otherwise the columns named by the list */
#define PLANE_COL(i) (pcols_all + (list ? (int64_t) __builtin_amdgcn_readfirstlane(list[i]) : (i)))
    PlaneCol c = k_load(PLANE_COL(col));
    int64_t my_off = lane < c.depth ? read_byte_off[c.read_off + lane] : 0;
    for (; col < n_cols; col += wave_stride) {
    const int64_t ncol = col + wave_stride < n_cols ? col + wave_stride : col;
    const PlaneCol cn = k_load(PLANE_COL(ncol));
    const int64_t next_off = lane < cn.depth ? read_byte_off[cn.read_off + lane] : 0;
    const bool active = lane < c.depth;
    const uint8_t *__restrict__ src = c.pool + my_off;
#define PLANE_U 8
    const int n_slots_here = (filter && !c.need_planes) ? 0 : c.n_slots; /* filter: the pack kernel handles the other columns */
    for (int s0 = 0; s0 < n_slots_here; s0 += PLANE_U) {
        uint32_t bytes[PLANE_U];
#pragma unroll
        for (int u = 0; u < PLANE_U; u++) bytes[u] = (active && s0 + u < c.n_slots) ? src[s0 + u] : 0u;
#pragma unroll
        for (int u = 0; u < PLANE_U; u++) {
            const int s = s0 + u;
            if (s < c.n_slots) { /* wave-uniform */
                const uint32_t byte = bytes[u];
                /* read-major copy for the dot-product emission kernel: word w = bytes of reads 4w..4w+3 */
                uint32_t packed = byte << (8 * (lane & 3));
                packed |= __shfl_xor(packed, 1, WAVE);
                packed |= __shfl_xor(packed, 2, WAVE);
                if ((lane & 3) == 0) slot_bytes[(c.slot_off + s) * 16 + (lane >> 2)] = packed;
                /* column-wide byte sum (the hap2 cost is total - hap1 cost) */
                uint32_t total = __builtin_amdgcn_udot4(packed, 0x01010101u, 0u, false);
                total += __shfl_xor(total, 4, WAVE);
                total += __shfl_xor(total, 8, WAVE);
                total += __shfl_xor(total, 16, WAVE);
                total += __shfl_xor(total, 32, WAVE);
                if (lane == 0) slot_total[c.slot_off + s] = total;
                if (c.need_planes) { /* bit planes: only the general / ancestor emission path reads them */
                    uint64_t mine = 0;
#pragma unroll
                    for (int b = 0; b < MRP_ALLELE_LOG_PROB_BITS; b++) {
                        const uint64_t v = __ballot((byte >> b) & 1u);
                        if (lane == b) mine = v;
                    }
                    if (lane < MRP_ALLELE_LOG_PROB_BITS) planes[(c.slot_off + s) * MRP_ALLELE_LOG_PROB_BITS + lane] = mine;
                }
            }
        }
    }
    c = cn;
    my_off = next_off;
    }
#undef PLANE_U
#undef PLANE_COL
}

/* The packed bytes and byte sums alone (columns of the fast emission path): 16 lanes per column, lane = word
 * (4 reads); every load of a lane is independent of the other lanes, nothing is exchanged but the 16-lane sum.
 * Round 5: FOUR slots per step -- a read's profile bytes of consecutive slots are consecutive bytes of the pool, so one (unaligned)
 * dword per read brings four slots, a 4 x 4 byte transpose (eight v_perm_b32) turns the four reads' dwords into the four slots'
 * words, and the four byte sums travel through the 16-lane reduction two to a register.  A byte load per read and slot (round 1) made
 * a column of the first merge levels -- two to four reads over sixty slots -- sixty dependent trips to memory per lane: the kernel
 * waited 80 % of its cycles and a step paid 12 ms for it (marginal cost, MRP_DUP=p).  The last dword of a read's run may reach up to
 * three bytes past it: into the next read's bytes or the padding behind the pool (mrp_chunk_create / mrp_chunk_block_create). */
struct __attribute__((packed, aligned(1))) pack_u32 { uint32_t v; };
__global__ void __launch_bounds__(256) mrp_pack_kernel(const PlaneCol *__restrict__ pcols, const int32_t *__restrict__ list,
                                                       int64_t n_list, const int64_t *__restrict__ read_byte_off,
                                                       uint32_t *__restrict__ slot_total, uint32_t *__restrict__ slot_bytes, int filter) {
    const int64_t q = ((int64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int w = threadIdx.x & 15;
    if (q >= n_list) return;
    const PlaneCol c = pcols[list ? list[q] : q];
    if (filter && c.need_planes) return; /* (all 16 lanes of a column leave together) */
    int64_t off[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int read = 4 * w + r;
        off[r] = read < c.depth ? read_byte_off[c.read_off + read] : -1;
    }
    for (int s = 0; s < c.n_slots; s += 4) {
        uint32_t v[4];
#pragma unroll
        for (int r = 0; r < 4; r++) /* (the pool is device memory: said to the compiler, which would otherwise use flat loads, counted twice) */
            v[r] = off[r] >= 0 ? ((const __attribute__((address_space(1))) pack_u32 *) (c.pool + off[r] + s))->v : 0u;
        /* v[r] = slots s .. s + 3 of read r  ->  o[j] = reads 0 .. 3 of slot s + j */
        const uint32_t ta = __builtin_amdgcn_perm(v[1], v[0], 0x05010400u), tb = __builtin_amdgcn_perm(v[1], v[0], 0x07030602u);
        const uint32_t tc = __builtin_amdgcn_perm(v[3], v[2], 0x05010400u), td = __builtin_amdgcn_perm(v[3], v[2], 0x07030602u);
        const uint32_t o[4] = {__builtin_amdgcn_perm(tc, ta, 0x05040100u), __builtin_amdgcn_perm(tc, ta, 0x07060302u),
                               __builtin_amdgcn_perm(td, tb, 0x05040100u), __builtin_amdgcn_perm(td, tb, 0x07060302u)};
        const int left = c.n_slots - s; /* (the same for the column's 16 lanes) */
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (j < left) slot_bytes[(c.slot_off + s + j) * 16 + w] = o[j];
        /* byte sums of the four slots, two per register (a slot's sum over 64 reads is below 2^16) */
        uint32_t t01 = __builtin_amdgcn_udot4(o[0], 0x01010101u, 0u, false) | (__builtin_amdgcn_udot4(o[1], 0x01010101u, 0u, false) << 16);
        uint32_t t23 = __builtin_amdgcn_udot4(o[2], 0x01010101u, 0u, false) | (__builtin_amdgcn_udot4(o[3], 0x01010101u, 0u, false) << 16);
        /* all-reduce over the column's 16 lanes (= one DPP row) by rotations: row_ror 8, 4, 2, 1 */
        t01 += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) t01, 0x128, 0xf, 0xf, false); t23 += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) t23, 0x128, 0xf, 0xf, false);
        t01 += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) t01, 0x124, 0xf, 0xf, false); t23 += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) t23, 0x124, 0xf, 0xf, false);
        t01 += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) t01, 0x122, 0xf, 0xf, false); t23 += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) t23, 0x122, 0xf, 0xf, false);
        t01 += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) t01, 0x121, 0xf, 0xf, false); t23 += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) t23, 0x121, 0xf, 0xf, false);
        if (w < 4 && w < left) slot_total[c.slot_off + s + w] = w == 0 ? (t01 & 0xFFFFu) : (w == 1 ? (t01 >> 16) : (w == 2 ? (t23 & 0xFFFFu) : (t23 >> 16)));
    }
}

hipError_t mrp_launch_planes(const MrpBatchDev &d, hipStream_t stream) {
    if (d.n_cols == 0) return hipSuccess;
    const int waves = 4;
    const bool split = d.pack_list != nullptr || d.plane_list != nullptr || d.list_filter != 0;
    const int64_t n_planes = split ? d.n_plane_list : d.n_cols;
    if (n_planes > 0) {
        int64_t grid = (n_planes + waves - 1) / waves;
        if (grid > MRP_PERSISTENT_GRID) grid = MRP_PERSISTENT_GRID;
        hipLaunchKernelGGL(mrp_planes_kernel, dim3((unsigned) grid), dim3(waves * WAVE), 0, stream, d.pcols,
                           split ? d.plane_list : (const int32_t *) nullptr, n_planes, d.read_byte_off, d.planes, d.slot_total,
                           d.slot_bytes, d.list_filter);
    }
    if (split && d.n_pack_list > 0) {
        const int64_t grid = (d.n_pack_list * 16 + 255) / 256;
        hipLaunchKernelGGL(mrp_pack_kernel, dim3((unsigned) grid), dim3(256), 0, stream, d.pcols, d.pack_list, d.n_pack_list,
                           d.read_byte_off, d.slot_total, d.slot_bytes, d.list_filter);
    }
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
/* getLogProbOfAllele (emissions.c:125-138) as a dot product: sum_{i in P} prob_i[a] =
 * sum_w dot4(bytes of reads 4w..4w+3, the four partition bits expanded to 0/1 bytes).  The
 * expansion is done once per cell and reused for every site and allele of the column. */
template <int NW>
static __device__ __forceinline__ void expand_partition(uint64_t P, uint32_t (&sel)[NW]) {
    const uint32_t lo = (uint32_t) P, hi = (uint32_t) (P >> 32);
#pragma unroll
    for (int w = 0; w < NW; w++) {
        const uint32_t nib = w < 8 ? ((lo >> (4 * w)) & 0xFu) : ((hi >> (4 * (w - 8))) & 0xFu);
        sel[w] = (nib * 0x00204081u) & 0x01010101u; /* bit k of the nibble -> byte k = 0/1 */
    }
}

/* cost of CP partitions over one column whose sites all have A alleles, no ancestor model
 * (emissions.c:187-207, :221-240).  NW = ceil(depth / 4) rounded up to 4, 8, 12 or 16. */
template <int CP, int NW>
static __device__ __forceinline__ void column_cost_dot(int64_t slot0, int n_slots, int A, K_AS(uint32_t) slot_bytes,
                                                       K_AS(uint32_t) slot_total, const uint64_t (&P)[CP],
                                                       uint32_t (&cost)[CP]) {
    uint32_t sel[CP][NW];
    uint32_t m1[CP], m2[CP];
#pragma unroll
    for (int j = 0; j < CP; j++) {
        expand_partition<NW>(P[j], sel[j]);
        cost[j] = 0;
        m1[j] = 0xFFFFFFFFu;
        m2[j] = 0xFFFFFFFFu;
    }
    /* scalar ping-pong: the bytes of the next allele slot are requested at the top of the step that
     * consumes the current one, into the other register set (no copies in between) */
    uint32_t B0[NW], B1[NW];
    uint32_t tot0, tot1 = 0;
    {
        K_AS(uint32_t) p = slot_bytes + slot0 * 16;
#pragma unroll
        for (int w = 0; w < NW; w++) B0[w] = p[w];
        tot0 = slot_total[slot0];
    }
    int a = 0;
#define DOT_STEP(BC, TC, BN, TN, g)                                                            \
    {                                                                                            \
        const int gn_ = (g) + 1 < n_slots ? (g) + 1 : (g);                                       \
        K_AS(uint32_t) p_ = slot_bytes + (slot0 + gn_) * 16;                                     \
        _Pragma("unroll") for (int w = 0; w < NW; w++) BN[w] = p_[w];                            \
        TN = slot_total[slot0 + gn_];                                                            \
        uint32_t lp_[CP];                                                                        \
        _Pragma("unroll") for (int j = 0; j < CP; j++) lp_[j] = 0;                               \
        /* the CP accumulation chains are interleaved word by word */                           \
        _Pragma("unroll") for (int w = 0; w < NW; w++) {                                         \
            _Pragma("unroll") for (int j = 0; j < CP; j++)                                       \
                lp_[j] = __builtin_amdgcn_udot4(BC[w], sel[j][w], lp_[j], false);                \
        }                                                                                        \
        _Pragma("unroll") for (int j = 0; j < CP; j++) {                                         \
            m1[j] = min(m1[j], lp_[j]);                                                          \
            m2[j] = min(m2[j], TC - lp_[j]);                                                     \
        }                                                                                        \
        if (++a == A) { /* site boundary */                                                      \
            a = 0;                                                                               \
            _Pragma("unroll") for (int j = 0; j < CP; j++) {                                     \
                cost[j] += m1[j] + m2[j];                                                        \
                m1[j] = 0xFFFFFFFFu;                                                             \
                m2[j] = 0xFFFFFFFFu;                                                             \
            }                                                                                    \
        }                                                                                        \
    }
    for (int g = 0; g < n_slots; g += 2) {
        DOT_STEP(B0, tot0, B1, tot1, g)
        if (g + 1 < n_slots) DOT_STEP(B1, tot1, B0, tot0, g + 1)
    }
#undef DOT_STEP
}

/* One wave per tile of up to MRP_EMIT_TILE consecutive cells of ONE column (uniform allele count,
 * no ancestor model): the column's packed bytes are wave-uniform (scalar loads, SGPR operands of
 * v_dot4_u32_u8).  Each lane owns EMIT_CP PAIRS of adjacent cells.  The emission cost is symmetric
 * under complement (hap1 <-> hap2, emissions.c:197-218) and the cross product stores every
 * partition next to its complement (hmm.c:627-655), so when cell 2i+1 is the complement of cell 2i
 * -- checked per lane -- the cost is computed once; otherwise a second pass handles the odd cells. */
template <int NW, int EMIT_CP>
static __device__ __forceinline__ void emit_tile(const MrpBatchDev &d, const EmitTile &t, int lane, int pair0) {
    K_AS(uint32_t) slot_bytes = K_PTR(uint32_t, d.slot_bytes);
    K_AS(uint32_t) slot_total = K_PTR(uint32_t, d.slot_total);
    const uint64_t mask = t.depth < 64 ? ~(0xFFFFFFFFFFFFFFFFull << t.depth) : 0xFFFFFFFFFFFFFFFFull;
    const int n_slots = t.n_sites * t.uniform_alleles;
    uint64_t Pe[EMIT_CP], Po[EMIT_CP];
    bool has_odd[EMIT_CP], paired[EMIT_CP];
    bool any_unpaired = false;
#pragma unroll
    for (int u = 0; u < EMIT_CP; u++) {
        const int e = 2 * (pair0 + u * WAVE + lane);
        Pe[u] = e < t.n ? d.partition[t.cell_off + e] : 0ull;
        has_odd[u] = e + 1 < t.n;
        Po[u] = has_odd[u] ? d.partition[t.cell_off + e + 1] : 0ull;
        paired[u] = has_odd[u] && Po[u] == (~Pe[u] & mask);
        any_unpaired |= has_odd[u] && !paired[u];
    }
    uint32_t cost[EMIT_CP];
    column_cost_dot<EMIT_CP, NW>(t.slot_off, n_slots, t.uniform_alleles, slot_bytes, slot_total, Pe, cost);
#pragma unroll
    for (int u = 0; u < EMIT_CP; u++) {
        const int e = 2 * (pair0 + u * WAVE + lane);
        if (e < t.n) d.cell_cost[t.cell_off + e] = cost[u];
        if (paired[u]) d.cell_cost[t.cell_off + e + 1] = cost[u];
    }
    if (__any(any_unpaired)) { /* wave-uniform: rare (pruned hmms, non-inverted mode) */
        column_cost_dot<EMIT_CP, NW>(t.slot_off, n_slots, t.uniform_alleles, slot_bytes, slot_total, Po, cost);
#pragma unroll
        for (int u = 0; u < EMIT_CP; u++) {
            const int e = 2 * (pair0 + u * WAVE + lane);
            if (has_odd[u] && !paired[u]) d.cell_cost[t.cell_off + e + 1] = cost[u];
        }
    }
}

__global__ void __launch_bounds__(256) mrp_emission_kernel(MrpBatchDev d, const EmitTile *__restrict__ tiles,
                                                           int64_t n_tiles) {
    const int lane = threadIdx.x & (WAVE - 1);
    /* threadIdx.x / 64 is wave-uniform but hipcc cannot prove it: without readfirstlane every
     * "scalar" load below degrades to 64 identical vector loads */
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x / WAVE));
    const int64_t tile = (int64_t) blockIdx.x * (blockDim.x / WAVE) + wave;
    if (tile >= n_tiles) return;
    const EmitTile t = k_load(tiles + tile);
    /* MRP_EMIT_TILE = 512 cells = 256 pairs: four pairs per lane, or two passes of two pairs per
     * lane for deep columns (their expanded partitions need 12-16 registers per pair) */
    if (t.depth <= 16) emit_tile<4, 4>(d, t, lane, 0);
    else if (t.depth <= 32) emit_tile<8, 4>(d, t, lane, 0);
    else if (t.depth <= 48) {
        emit_tile<12, 2>(d, t, lane, 0);
        if (t.n > 256) emit_tile<12, 2>(d, t, lane, 128);
    } else {
        emit_tile<16, 2>(d, t, lane, 0);
        if (t.n > 256) emit_tile<16, 2>(d, t, lane, 128);
    }
}

/* tiles whose column mixes allele counts or uses the ancestor substitution model (final sweeps) */
__global__ void __launch_bounds__(256) mrp_emission_general_kernel(MrpBatchDev d, const EmitTile *__restrict__ tiles,
                                                                   int64_t n_tiles) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x / WAVE));
    const int64_t tile = (int64_t) blockIdx.x * (blockDim.x / WAVE) + wave;
    if (tile >= n_tiles) return;
    const EmitTile t = k_load(tiles + tile);
    const DevCol c = k_load(d.cols + t.col);
    const DevChunk ch = k_load(d.chunks + c.chunk);
    K_AS(uint64_t) planes = K_PTR(uint64_t, d.planes);
    K_AS(uint32_t) slot_total = K_PTR(uint32_t, d.slot_total);
    const bool ancestor = (t.flags & MRP_FLAG_INCLUDE_ANCESTOR_SUB_PROB) != 0;
    for (int idx = lane; idx < t.n; idx += WAVE) {
        const uint64_t P[1] = {d.partition[t.cell_off + idx]};
        uint32_t cost[1];
        if (ancestor) cost[0] = column_cost_ancestor(c, ch, planes, slot_total, P[0]);
        else column_cost_plain<1, false>(c, K_PTR(uint32_t, ch.allele_number), planes, slot_total, P, cost);
        d.cell_cost[t.cell_off + idx] = cost[0];
    }
}

/* Resident batches: the host describes columns only; their tiles (48 B per 512 cells, tens of MB at the large levels) are
 * written here instead of being filled by host threads, partitioned and uploaded. */
__global__ void __launch_bounds__(256) mrp_tiles_kernel(const DevCol *__restrict__ cols, const TileCol *__restrict__ tc, int64_t n_cols,
                                                        EmitTile *__restrict__ tiles) {
    const int64_t col = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= n_cols) return;
    const DevCol c = cols[col];
    const TileCol t0 = tc[col];
    int64_t at = t0.first;
    for (int32_t off = 0; off < c.n_cells; off += MRP_EMIT_TILE, at++) {
        EmitTile t;
        t.cell_off = c.cell_off + off;
        t.slot_off = c.slot_off;
        t.n = c.n_cells - off < MRP_EMIT_TILE ? c.n_cells - off : MRP_EMIT_TILE;
        t.col = (int32_t) col;
        t.n_sites = c.n_sites;
        t.uniform_alleles = t0.uniform_alleles;
        t.depth = c.depth;
        t.flags = c.flags;
        t.pad[0] = 0;
        t.pad[1] = 0;
        tiles[at] = t;
    }
}

hipError_t mrp_launch_tiles(const DevCol *cols_dev, const TileCol *tilecols_dev, int64_t n_cols, EmitTile *tiles_dev, hipStream_t stream) {
    if (n_cols <= 0) return hipSuccess;
    hipLaunchKernelGGL(mrp_tiles_kernel, dim3((unsigned) ((n_cols + 255) / 256)), dim3(256), 0, stream, cols_dev, tilecols_dev, n_cols, tiles_dev);
    return hipGetLastError();
}

hipError_t mrp_launch_emission(const MrpBatchDev &d, const EmitTile *tiles_dev, int64_t n_fast, int64_t n_general,
                               hipStream_t stream) {
    const int waves = 4;
    if (n_fast > 0)
        hipLaunchKernelGGL(mrp_emission_kernel, dim3((unsigned) ((n_fast + waves - 1) / waves)), dim3(waves * WAVE), 0,
                           stream, d, tiles_dev, n_fast);
    if (n_general > 0)
        hipLaunchKernelGGL(mrp_emission_general_kernel, dim3((unsigned) ((n_general + waves - 1) / waves)),
                           dim3(waves * WAVE), 0, stream, d, tiles_dev + n_fast, n_general);
    return hipGetLastError();
}

/* ------------------------------------------------------------------------------------------ */
/* max-plus recursion, int32 merge arrays in LDS                                               */
/* ------------------------------------------------------------------------------------------ */
/*
 * One persistent workgroup per hmm.  The cells of an hmm are ONE contiguous stream in HBM (column
 * after column), so the kernel prefetches by stream ROUNDS of 4*T cells -- full-width, unconditional,
 * 16-byte-per-lane loads of (cost, next|prev) held in a ring of SWEEP_R register sets -- and the
 * sequential column walk merely consumes them: a wide column spans several rounds (no barrier in
 * between), a round may hold many narrow columns.  Nothing on the recursion's dependency chain
 * waits for HBM.  Column descriptors are staged through LDS in windows of SWEEP_WIN columns.
 *
 * Column totals: in max-plus arithmetic max_c(f_c + b_c) is the score of the best complete path
 * for EVERY column, i.e. stRPColumn.totalLogProb == stRPHmm.forwardLogProb exactly (the reference
 * computes the same integers, hmm.c:906-907), so col_total is a broadcast of the forward score.
 */
#define SWEEP_R 8
#ifndef SWEEP_WIN
#define SWEEP_WIN 128
#endif

struct SweepShared {
    int32_t *cur;  /* merge column read by the column being processed */
    int32_t *nxt;  /* merge column being accumulated */
    int32_t *red;  /* [0] hmm forward, [1] hmm backward */
    int2 *dsc;     /* staged {n_cells, n_merge} of columns [win0, win0 + SWEEP_WIN) */
    int win0;
};

static __device__ __forceinline__ void stage_window(SweepShared &S, const SweepCol *cols, int K, int win0, int tid, int T) {
    __syncthreads(); /* everybody is done with the old window (also a full memory fence; rare) */
    for (int i = tid; i < SWEEP_WIN; i += T) {
        const int k = win0 + i;
        if (k >= 0 && k < K) S.dsc[i] = make_int2(cols[k].n_cells, cols[k].n_merge);
    }
    S.win0 = win0;
    __syncthreads();
}
static __device__ __forceinline__ int2 col_desc(const SweepShared &S, int k) {
    const int2 v = S.dsc[k - S.win0];
    return make_int2(__builtin_amdgcn_readfirstlane(v.x), __builtin_amdgcn_readfirstlane(v.y));
}

__global__ void __launch_bounds__(1024)
mrp_sweep_i32_kernel(MrpBatchDev d, const int32_t *__restrict__ order, int max_merge) {
    extern __shared__ __attribute__((aligned(16))) int32_t lds[];
    SweepShared S;
    S.cur = lds;
    S.nxt = lds + max_merge;
    S.red = lds + 2 * max_merge;
    S.dsc = reinterpret_cast<int2 *>(lds + 2 * max_merge + 4);
    S.win0 = 0;

    const int tid = threadIdx.x, T = blockDim.x, lane = threadIdx.x & (WAVE - 1);
    /* the forward and the backward recursion do not depend on each other (only the posteriors combine
     * them): they run as two workgroups, halving the sequential chain per workgroup */
    const bool backward = (blockIdx.x & 1) != 0;
    const int64_t hmm_index = K_PTR(int32_t, order)[blockIdx.x >> 1];
    const DevHmm h = k_load(d.hmms + hmm_index);
    const SweepCol *cols = d.scols + h.col0;
    const int K = h.n_cols;
    const SweepCol first_col = k_load(cols);
    const int N = (int) h.n_cells;                 /* cells of this hmm (stream length) */
    const int G = (N + 3) / 4;                     /* groups of 4 consecutive cells: one per thread per round */
    const int Q = (G + T - 1) / T;                 /* rounds */
    /* every hmm starts at a multiple of 4 cells in the batch arrays: 16-byte aligned vector access */
    const uint32_t *__restrict__ cost = d.cell_cost + first_col.cell_off;
    const uint32_t *__restrict__ np = d.cell_np + first_col.cell_off;
    /* max-plus results are exact integers: they stay in HBM as int32 (log(0) = MRP_NEG_I32) and are
     * widened to the reference's doubles when they cross the host boundary (mrp_batch_download) */
    int32_t *__restrict__ out_f = d.cell_f32 + first_col.cell_off;
    int32_t *__restrict__ out_b = d.cell_b32 + first_col.cell_off;
    int32_t *__restrict__ out_mf = d.merge_f32 + first_col.mcell_off;
    int32_t *__restrict__ out_mb = d.merge_b32 + first_col.mcell_off;

    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
    u32x4 rc0, rc1, rc2, rc3, rc4, rc5, rc6, rc7, rn0, rn1, rn2, rn3, rn4, rn5, rn6, rn7;
    /* The ring loads are issued by inline asm so that hipcc's waitcnt pass does not see them (through
     * this control flow it would wait with vmcnt(0), i.e. drain the whole ring, before every use).
     * We wait ourselves.  Memory operations retire in issue order and stores share the counter, so
     * the count of operations issued AFTER the wanted ring entry must be known exactly: every round
     * issues exactly 1 vector store (unconditional; lanes without a valid group write to a scratch
     * slot) followed by exactly 2 ring loads, plus a variable number of merge-column stores that
     * can only make the wait stricter.  Entry i is the oldest ring entry when its round starts:
     * in steady state 7 rounds * 3 operations are younger -> vmcnt(21); in the first lap round i has
     * only i rounds of stores behind it -> vmcnt(14 + i).  RING_WAIT ties the registers so no use
     * can be scheduled above the wait. */
#define RING_LOAD(i, q)                                                                              \
    {                                                                                                \
        int g_ = (q) * T + tid;                                                                      \
        g_ = g_ < 0 ? 0 : (g_ > G - 1 ? G - 1 : g_);                                                 \
        asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %3, off"            \
                     : "=&v"(rc##i), "=&v"(rn##i)                                                    \
                     : "v"(cost + 4 * g_), "v"(np + 4 * g_)                                          \
                     : "memory");                                                                    \
    }
#ifdef MRP_SAFE_WAIT /* debugging aid: drain everything */
#define RING_WAIT(i, n) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rc##i), "+v"(rn##i)::"memory");
#else
#define RING_WAIT(i, n) asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(rc##i), "+v"(rn##i)::"memory");
#endif
    if (!backward) {
        RING_LOAD(0, 0) RING_LOAD(1, 1) RING_LOAD(2, 2) RING_LOAD(3, 3)
        RING_LOAD(4, 4) RING_LOAD(5, 5) RING_LOAD(6, 6) RING_LOAD(7, 7)
    } else {
        RING_LOAD(0, Q - 1) RING_LOAD(1, Q - 2) RING_LOAD(2, Q - 3) RING_LOAD(3, Q - 4)
        RING_LOAD(4, Q - 5) RING_LOAD(5, Q - 6) RING_LOAD(6, Q - 7) RING_LOAD(7, Q - 8)
    }

    for (int i = tid; i < 2 * max_merge; i += T) lds[i] = MRP_NEG_I32;
    if (tid < 4) S.red[tid] = MRP_NEG_I32;
    stage_window(S, cols, K, 0, tid, T);

    /* exactly ONE 16-byte store per lane per round; the hmm's cell range is padded to a multiple of 4
     * cells, so a partial last group spills into padding; lanes past the stream use the scratch slot */
#define STORE4(out, p0, ok, v)                                                                               \
    {                                                                                                        \
        int32_t *dst_ = (ok) ? (out) + (p0) : scratch;                                                       \
        const i32x4 q_ = {v[0], v[1], v[2], v[3]};                                                           \
        *reinterpret_cast<i32x4 *>(dst_) = q_;                                                               \
    }
    int32_t *scratch = d.cell_f32 + d.n_cells; /* 64 bytes of slack behind the batch array, never read */

    /* Retire the ring.  The last loads of each entry are never consumed, so for the compiler their
     * registers are dead: the drain must NAME them, otherwise register-only instructions hoisted above
     * a plain "s_waitcnt" asm could be allocated into registers a load in flight still overwrites. */
#define RING_DRAIN() asm volatile("s_waitcnt vmcnt(0)"                                                                          \
                 : "+v"(rc0), "+v"(rc1), "+v"(rc2), "+v"(rc3), "+v"(rc4), "+v"(rc5), "+v"(rc6), "+v"(rc7),       \
                   "+v"(rn0), "+v"(rn1), "+v"(rn2), "+v"(rn3), "+v"(rn4), "+v"(rn5), "+v"(rn6), "+v"(rn7)        \
                 :                                                                                             \
                 : "memory");

    /* ---------------- forward (hmm.c:827-879) ---------------- */
    if (!backward) {
        int k = 0;
        int2 dk = col_desc(S, 0);
        int cs = 0, ce = dk.x;   /* stream interval of column k */
        int mo = 0;              /* merge cells before merge column k */
        int32_t local_max = MRP_NEG_I32;
#define FWD_ROUND(i, qq, nw)                                                                               \
    RING_WAIT(i, nw)                                                                                         \
    {                                                                                                        \
    const int pos0 = 4 * ((qq) * T + tid);                                                                   \
    int32_t fv[4] = {MRP_NEG_I32, MRP_NEG_I32, MRP_NEG_I32, MRP_NEG_I32};                                    \
    if ((qq) < Q && k < K) {                                                                                 \
        const int rend = 4 * ((qq) + 1) * T;                                                                 \
        for (;;) {                                                                                           \
            const bool first = (k == 0), last = (k == K - 1);                                                \
            /* batched and branch-free inside the wave: four independent LDS gathers, one wait, four     \
             * ds_max.  Cells outside the column use a per-lane dummy slot and the value log(0), for   \
             * which max is a no-op; waves without any cell of the column skip the block. */            \
            bool act[4];                                                                                     \
            bool any_act = false;                                                                            \
            _Pragma("unroll") for (int j = 0; j < 4; j++) {                                                  \
                const int p = pos0 + j;                                                                      \
                act[j] = (p >= cs) & (p < ce);                                                               \
                any_act |= act[j];                                                                           \
            }                                                                                                \
            if (__any(any_act)) {                                                                            \
                int32_t fp[4] = {0, 0, 0, 0};                                                                \
                if (!first) {                                                                                \
                    _Pragma("unroll") for (int j = 0; j < 4; j++)                                            \
                        fp[j] = S.cur[act[j] ? (rn##i[j] >> 16) : lane];     /* forwardCellCalc1 :791 */      \
                }                                                                                            \
                _Pragma("unroll") for (int j = 0; j < 4; j++) {                                              \
                    const int32_t v = add_i32(fp[j], -(int32_t) rc##i[j]);                                   \
                    fv[j] = act[j] ? v : fv[j];                                                              \
                    const int32_t val = act[j] ? v : MRP_NEG_I32;                                            \
                    if (!last) atomicMax(&S.nxt[act[j] ? (rn##i[j] & 0xFFFFu) : lane], val); /* :814 */      \
                    else local_max = max(local_max, val);                                                    \
                }                                                                                            \
            }                                                                                                \
            if (ce > rend) break; /* the column continues in the next round */                               \
            /* column k is complete */                                                                       \
            if (last) {                                                                                      \
                local_max = wave_max_i32(local_max);                                                         \
                if ((tid & (WAVE - 1)) == 0) atomicMax(&S.red[0], local_max);                                \
                k = K;                                                                                       \
                break;                                                                                       \
            }                                                                                                \
            if (k + 1 >= S.win0 + SWEEP_WIN) stage_window(S, cols, K, k, tid, T);                            \
            const int2 dn = col_desc(S, k + 1);                                                              \
            lds_barrier();                                                                                   \
            for (int m = tid; m < dk.y; m += T) out_mf[mo + m] = S.nxt[m];                       \
            if (k + 2 < K) for (int m = tid; m < dn.y; m += T) S.cur[m] = MRP_NEG_I32;                       \
            lds_barrier();                                                                                   \
            { int32_t *t_ = S.cur; S.cur = S.nxt; S.nxt = t_; }                                              \
            mo += dk.y;                                                                                      \
            k++;                                                                                             \
            dk = dn;                                                                                         \
            cs = ce;                                                                                         \
            ce = cs + dk.x;                                                                                  \
            if (cs >= rend) break;                                                                           \
        }                                                                                                    \
    }                                                                                                        \
    STORE4(out_f, pos0, (qq) < Q && pos0 < N, fv)                                                            \
    }                                                                                                        \
    RING_LOAD(i, (qq) + SWEEP_R)
        /* first lap: round i has only i rounds of stores behind its ring entry */
        FWD_ROUND(0, 0, 14) FWD_ROUND(1, 1, 15) FWD_ROUND(2, 2, 16) FWD_ROUND(3, 3, 17)
        FWD_ROUND(4, 4, 18) FWD_ROUND(5, 5, 19) FWD_ROUND(6, 6, 20) FWD_ROUND(7, 7, 21)
        for (int q0 = SWEEP_R; q0 < Q; q0 += SWEEP_R) {
            FWD_ROUND(0, q0, 21) FWD_ROUND(1, q0 + 1, 21) FWD_ROUND(2, q0 + 2, 21) FWD_ROUND(3, q0 + 3, 21)
            FWD_ROUND(4, q0 + 4, 21) FWD_ROUND(5, q0 + 5, 21) FWD_ROUND(6, q0 + 6, 21) FWD_ROUND(7, q0 + 7, 21)
        }
#undef FWD_ROUND
        RING_DRAIN()
        __syncthreads();
        /* stRPColumn.totalLogProb of every column == forward score (see the header comment) */
        const double total = i32_to_log(S.red[0]);
        for (int kk = tid; kk < K; kk += T) d.col_total[h.col0 + kk] = total;
        if (tid == 0) d.hmm_fb[2 * hmm_index] = total;
    }
    /* ---------------- backward (hmm.c:910-929) ---------------- */
    /* cur = mb of the merge column after column k (read), nxt = mb of the one before (accumulated) */
    else {
        int k = K - 1;
        if (k < S.win0 || k >= S.win0 + SWEEP_WIN) stage_window(S, cols, K, max(0, K - SWEEP_WIN), tid, T);
        int2 dk = col_desc(S, k);
        int ce = N, cs = N - dk.x;
        int mo = (int) h.n_merge;  /* merge cells up to and including merge column k-1 end here */
        if (K >= 2) {
            if (k - 1 < S.win0) stage_window(S, cols, K, max(0, k - SWEEP_WIN + 1), tid, T);
            const int2 dp = col_desc(S, k - 1);
            for (int m = tid; m < dp.y; m += T) S.nxt[m] = MRP_NEG_I32;
        }
        lds_barrier();
        int32_t local_max = MRP_NEG_I32;
#define BWD_ROUND(i, qq, nw)                                                                                 \
    RING_WAIT(i, nw)                                                                                         \
    {                                                                                                        \
    const int pos0 = 4 * ((qq) * T + tid);                                                                   \
    int32_t bv[4] = {MRP_NEG_I32, MRP_NEG_I32, MRP_NEG_I32, MRP_NEG_I32};                                    \
    if ((qq) >= 0 && k >= 0) {                                                                               \
        const int rbeg = 4 * (qq) * T;                                                                       \
        for (;;) {                                                                                           \
            const bool first = (k == 0), last = (k == K - 1);                                                \
            bool act[4];                                                                                     \
            bool any_act = false;                                                                            \
            _Pragma("unroll") for (int j = 0; j < 4; j++) {                                                  \
                const int p = pos0 + j;                                                                      \
                act[j] = (p >= cs) & (p < ce);                                                               \
                any_act |= act[j];                                                                           \
            }                                                                                                \
            if (__any(any_act)) {                                                                            \
                int32_t bn[4] = {0, 0, 0, 0};                                                                \
                if (!last) {                                                                                 \
                    _Pragma("unroll") for (int j = 0; j < 4; j++)                                            \
                        bn[j] = S.cur[act[j] ? (rn##i[j] & 0xFFFFu) : lane]; /* backwardCellCalc :881 */      \
                }                                                                                            \
                _Pragma("unroll") for (int j = 0; j < 4; j++) {                                              \
                    bv[j] = act[j] ? bn[j] : bv[j];                                                          \
                    const int32_t pv = act[j] ? add_i32(bn[j], -(int32_t) rc##i[j]) : MRP_NEG_I32;           \
                    if (!first) atomicMax(&S.nxt[act[j] ? (rn##i[j] >> 16) : lane], pv);                     \
                    else local_max = max(local_max, pv);                                                     \
                }                                                                                            \
            }                                                                                                \
            if (cs < rbeg) break; /* the column continues in the previous round */                           \
            if (first) {                                                                                     \
                local_max = wave_max_i32(local_max);                                                         \
                if ((tid & (WAVE - 1)) == 0) atomicMax(&S.red[1], local_max);                                \
                k = -1;                                                                                      \
                break;                                                                                       \
            }                                                                                                \
            if (k - 2 >= 0 && k - 2 < S.win0) stage_window(S, cols, K, max(0, k - SWEEP_WIN + 1), tid, T);   \
            else if (k - 1 < S.win0) stage_window(S, cols, K, max(0, k - SWEEP_WIN + 1), tid, T);            \
            const int2 dp = col_desc(S, k - 1);                                                              \
            const int n_clear = (k >= 2) ? col_desc(S, k - 2).y : 0;                                         \
            lds_barrier();                                                                                   \
            for (int m = tid; m < dp.y; m += T) out_mb[mo - dp.y + m] = S.nxt[m];                \
            for (int m = tid; m < n_clear; m += T) S.cur[m] = MRP_NEG_I32;                                   \
            lds_barrier();                                                                                   \
            { int32_t *t_ = S.cur; S.cur = S.nxt; S.nxt = t_; }                                              \
            mo -= dp.y;                                                                                      \
            k--;                                                                                             \
            dk = dp;                                                                                         \
            ce = cs;                                                                                         \
            cs = ce - dk.x;                                                                                  \
            if (ce <= rbeg) break;                                                                           \
        }                                                                                                    \
    }                                                                                                        \
    STORE4(out_b, pos0, (qq) >= 0 && pos0 < N, bv)                                                           \
    }                                                                                                        \
    RING_LOAD(i, (qq) - SWEEP_R)
        BWD_ROUND(0, Q - 1, 14) BWD_ROUND(1, Q - 2, 15) BWD_ROUND(2, Q - 3, 16) BWD_ROUND(3, Q - 4, 17)
        BWD_ROUND(4, Q - 5, 18) BWD_ROUND(5, Q - 6, 19) BWD_ROUND(6, Q - 7, 20) BWD_ROUND(7, Q - 8, 21)
        for (int q0 = Q - 1 - SWEEP_R; q0 >= 0; q0 -= SWEEP_R) {
            BWD_ROUND(0, q0, 21) BWD_ROUND(1, q0 - 1, 21) BWD_ROUND(2, q0 - 2, 21) BWD_ROUND(3, q0 - 3, 21)
            BWD_ROUND(4, q0 - 4, 21) BWD_ROUND(5, q0 - 5, 21) BWD_ROUND(6, q0 - 6, 21) BWD_ROUND(7, q0 - 7, 21)
        }
#undef BWD_ROUND
        RING_DRAIN()
        __syncthreads();
        if (tid == 0) d.hmm_fb[2 * hmm_index + 1] = i32_to_log(S.red[1]);
    }
#undef RING_DRAIN
#undef RING_LOAD
#undef RING_WAIT
#undef STORE4
}

hipError_t mrp_launch_sweep_i32(const MrpBatchDev &d, const int32_t *order_dev, int64_t n, int block_threads,
                                int max_merge, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    if (max_merge < WAVE) max_merge = WAVE; /* per-lane dummy slots of the branch-free cell step */
    const size_t lds = (size_t) (2 * max_merge + 4) * sizeof(int32_t) + SWEEP_WIN * sizeof(int2);
    auto k = mrp_sweep_i32_kernel;
    /* once per device, never per launch (concurrent host threads launch different size classes) */
    static PerDeviceOnce once;
    const hipError_t attr_status = once.run([] {
        return hipFuncSetAttribute((const void *) mrp_sweep_i32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    if (attr_status != hipSuccess) return attr_status;
    hipLaunchKernelGGL(k, dim3((unsigned) (2 * n)), dim3(block_threads), lds, stream, d, order_dev, max_merge);
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
    const SweepCol *cols = d.scols + h.col0;
    const int K = h.n_cols;
    const bool wide = h.wide_idx != 0;
    const bool max_not_sum = (h.flags & MRP_FLAG_MAX_NOT_SUM) != 0;
    const double NEG = -__builtin_inf();
    double *hmm_f = d.hmm_fb + 2 * hmm_index, *hmm_b = hmm_f + 1;
    /* merge_f / merge_b / col_total / hmm_fb were filled with -inf before the launch (hmm.c:752-789) */

    for (int k = 0; k < K; k++) {
        const SweepCol c = k_load(cols + k);
        const bool first = (k == 0), last = (k == K - 1);
        const SweepCol pc = k_load(cols + (first ? k : k - 1));
        double local = NEG;
        for (int idx = tid; idx < c.n_cells; idx += T) {
            const int64_t g = c.cell_off + idx;
            const double e = -((double) d.cell_cost[g]);                         /* emissions.c:239 */
            const uint32_t nx = wide ? d.cell_next[g] : (d.cell_np[g] & 0xFFFFu);
            const uint32_t pv = wide ? d.cell_prev[g] : (d.cell_np[g] >> 16);
            double fv = first ? 0.0 : load_agent(&d.merge_f[pc.mcell_off + pv]);
            fv += e;
            d.cell_f[g] = fv;
            if (!last) atomic_log_add_p(&d.merge_f[c.mcell_off + nx], fv, max_not_sum);
            else local = log_add_p(local, fv, max_not_sum);
        }
        if (last) {
            local = wave_log_add_p(local, max_not_sum);
            if ((tid & (WAVE - 1)) == 0) atomic_log_add_p(hmm_f, local, max_not_sum);
        }
        __syncthreads();
    }
    for (int k = K - 1; k >= 0; k--) {
        const SweepCol c = k_load(cols + k);
        const bool first = (k == 0), last = (k == K - 1);
        const SweepCol pc = k_load(cols + (first ? k : k - 1));
        double local_b = NEG, local_t = NEG;
        for (int idx = tid; idx < c.n_cells; idx += T) {
            const int64_t g = c.cell_off + idx;
            const uint32_t nx = wide ? d.cell_next[g] : (d.cell_np[g] & 0xFFFFu);
            const uint32_t pv = wide ? d.cell_prev[g] : (d.cell_np[g] >> 16);
            double p = -((double) d.cell_cost[g]);
            double bv = 0.0;
            if (!last) { bv = load_agent(&d.merge_b[c.mcell_off + nx]); p += bv; }
            d.cell_b[g] = bv;
            if (!first) atomic_log_add_p(&d.merge_b[pc.mcell_off + pv], p, max_not_sum);
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

/* ------------------------------------------------------------------------------------------ */
/* log-sum-exp sweep (sum mode, hmm.c:15-20 logAddP with maxNotSumTransitions == false)         */
/* ------------------------------------------------------------------------------------------ */
/*
 * One workgroup per hmm; the merge column being summed lives in LDS.  The reference adds a column's cells into their merge
 * cells one after the other (forwardCellCalc2 hmm.c:814-825, backwardCellCalc :881-908); floating point addition in a
 * different order gives different last bits, so a parallel version is only reproducible if its result does not depend on
 * the order at all.  Here a merge cell's value is  r + log(sum_i exp(v_i - r))  with
 *   r       the largest contribution, rounded UP to a float: a 32-bit integer atomic max in LDS (order-free), and
 *   the sum accumulated in 64-bit FIXED POINT (2^-50 units; every term is <= 1, a column has < 2^14 cells): integer
 *           atomic adds in LDS, associative, hence identical from run to run and for any wave schedule.
 * A term below 2^-50 of the largest one contributes nothing (35 nats down); the value differs from the reference's sequential
 * sum by about n 2^-50 relative, i.e. 1e-11 at most per column and well inside the 1e-9 the tests ask for (north star: 1e-5
 * on posteriors).  Column totals (hmm.c:906-907) and the hmm's forward / backward probability are reductions over ALL cells of
 * a column: per thread in index order, then across the wave by shuffles, then across the waves in wave order.
 * LDS: 20 B per merge cell (value f64, sum u64, reference point i32), or 12 B when the finished values are read back from
 * HBM (merge columns above 8000 cells); hmms with still wider merge columns, or with transitions that need 32 bits, take
 * mrp_sweep_f64_kernel.
 */
#define LSE_KEY_NEG_INF ((int32_t) 0x807FFFFF)
static __device__ __forceinline__ int32_t lse_key_up(double v) { /* order-preserving integer image of v rounded up to float */
    float f = (float) v;
    if ((double) f < v) f = __int_as_float(__float_as_int(f) + (f >= 0.0f ? 1 : -1)); /* next float towards +inf (v finite, f != 0 here or f = -0/+0 -> fine) */
    const int32_t b = __float_as_int(f);
    return b >= 0 ? b : (int32_t) (b ^ 0x7FFFFFFF);
}
static __device__ __forceinline__ double lse_key_value(int32_t k) {
    return (double) __int_as_float(k >= 0 ? k : (int32_t) (k ^ 0x7FFFFFFF));
}
#define LSE_FIX 1125899906842624.0 /* 2^50 */
struct LseAcc { double m, s; };     /* running log-sum-exp: value = m + log(s) */
static __device__ __forceinline__ void lse_push(LseAcc &a, double v) {
    if (v == -__builtin_inf()) return;
    if (v > a.m) { a.s = a.s * exp(a.m - v) + 1.0; a.m = v; }
    else a.s += exp(v - a.m);
}
static __device__ __forceinline__ void lse_merge(LseAcc &a, double m, double s) {
    if (m == -__builtin_inf()) return;
    if (m > a.m) { a.s = a.s * exp(a.m - m) + s; a.m = m; }
    else a.s += s * exp(m - a.m);
}
static __device__ __forceinline__ double lse_value(const LseAcc &a) { return a.m == -__builtin_inf() ? a.m : a.m + log(a.s); }

/* CUR_LDS: the finished merge column (what the next column's cells read) is kept in LDS too; otherwise it is read back from
 * the merge_f / merge_b arrays in HBM, which the workgroup has just written (12 B of LDS per merge cell: merge columns up to
 * the 13 456 cells of two 116-cell parents fit). */
template <bool CUR_LDS>
__global__ void __launch_bounds__(512) mrp_sweep_lse_kernel(MrpBatchDev d, const int32_t *__restrict__ order, int max_merge) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lse_lds[];
    unsigned long long *acc = reinterpret_cast<unsigned long long *>(lse_lds);
    double *cur = reinterpret_cast<double *>(lse_lds + 8 * (size_t) max_merge);
    double *part = cur + (CUR_LDS ? max_merge : 0); /* [2 * 8] per-wave partial reductions */
    int32_t *rkey = reinterpret_cast<int32_t *>(part + 16);
    const int tid = threadIdx.x, T = blockDim.x, lane = tid & (WAVE - 1), wave = tid / WAVE, n_waves = T / WAVE;
    const int64_t hmm_index = K_PTR(int32_t, order)[blockIdx.x];
    const DevHmm h = k_load(d.hmms + hmm_index);
    const SweepCol *cols = d.scols + h.col0;
    const int K = h.n_cols;
    const double NEG = -__builtin_inf();
    for (int m = tid; m < max_merge; m += T) { rkey[m] = LSE_KEY_NEG_INF; acc[m] = 0ull; if (CUR_LDS) cur[m] = NEG; }
    __syncthreads();

    /* what a finished merge column holds: value of merge cell m, then the cell is cleared for the next column */
    auto finish = [&](int m) -> double {
        const unsigned long long a = acc[m];
        const double v = a ? lse_key_value(rkey[m]) + log((double) a * (1.0 / LSE_FIX)) : NEG;
        rkey[m] = LSE_KEY_NEG_INF;
        acc[m] = 0ull;
        return v;
    };

    /* ---------------- forward (hmm.c:827-879) ---------------- */
    for (int k = 0; k < K; k++) {
        const SweepCol c = k_load(cols + k);
        const bool first = (k == 0), last = (k == K - 1);
        const double *prev_mf = d.merge_f + k_load(cols + (first ? k : k - 1)).mcell_off; /* __syncthreads() below: written, visible */
        for (int idx = tid; idx < c.n_cells; idx += T) {
            const int64_t g = c.cell_off + idx;
            const uint32_t np = d.cell_np[g];
            const double fv = (first ? 0.0 : (CUR_LDS ? cur[np >> 16] : load_agent(const_cast<double *>(prev_mf + (np >> 16))))) - (double) d.cell_cost[g]; /* forwardCellCalc1 :791, emissions.c:239 */
            d.cell_f[g] = fv;
            if (fv != NEG) atomicMax(&rkey[last ? 0u : (np & 0xFFFFu)], lse_key_up(fv));
        }
        __syncthreads();
        for (int idx = tid; idx < c.n_cells; idx += T) {
            const int64_t g = c.cell_off + idx;
            const uint32_t to = last ? 0u : (d.cell_np[g] & 0xFFFFu);
            const double fv = d.cell_f[g];
            if (fv != NEG) atomicAdd(&acc[to], (unsigned long long) (exp(fv - lse_key_value(rkey[to])) * LSE_FIX)); /* forwardCellCalc2 :814 */
        }
        __syncthreads();
        if (!last) {
            for (int m = tid; m < c.n_merge; m += T) {
                const double v = finish(m);
                if (CUR_LDS) cur[m] = v;
                d.merge_f[c.mcell_off + m] = v;
            }
        } else if (tid == 0) d.hmm_fb[2 * hmm_index] = finish(0); /* stRPHmm.forwardLogProb :872-878 */
        __syncthreads();
    }
    /* ---------------- backward (hmm.c:910-929), column totals (:906-907) ---------------- */
    for (int k = K - 1; k >= 0; k--) {
        const SweepCol c = k_load(cols + k);
        const bool first = (k == 0), last = (k == K - 1);
        const SweepCol pc = k_load(cols + (first ? k : k - 1));
        LseAcc tot = {NEG, 0.0};
        for (int idx = tid; idx < c.n_cells; idx += T) {
            const int64_t g = c.cell_off + idx;
            const uint32_t np = d.cell_np[g];
            const double bv = last ? 0.0 : (CUR_LDS ? cur[np & 0xFFFFu] : load_agent(d.merge_b + c.mcell_off + (np & 0xFFFFu)));
            d.cell_b[g] = bv;
            const double p = bv - (double) d.cell_cost[g]; /* backwardCellCalc :881 */
            if (p != NEG) atomicMax(&rkey[first ? 0u : (np >> 16)], lse_key_up(p));
            lse_push(tot, d.cell_f[g] + bv);
        }
        /* the column total: across the wave by shuffles (fixed tree), then across the waves in wave order */
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double om = __shfl_xor(tot.m, o, WAVE), os = __shfl_xor(tot.s, o, WAVE);
            /* both partners must combine the same two operands in the same order: lower lane's value first */
            LseAcc lo = (lane & o) ? LseAcc{om, os} : tot, hi = (lane & o) ? tot : LseAcc{om, os};
            lse_merge(lo, hi.m, hi.s);
            tot = lo;
        }
        if (lane == 0) { part[2 * wave] = tot.m; part[2 * wave + 1] = tot.s; }
        __syncthreads();
        for (int idx = tid; idx < c.n_cells; idx += T) {
            const int64_t g = c.cell_off + idx;
            const uint32_t to = first ? 0u : (d.cell_np[g] >> 16);
            const double p = d.cell_b[g] - (double) d.cell_cost[g];
            if (p != NEG) atomicAdd(&acc[to], (unsigned long long) (exp(p - lse_key_value(rkey[to])) * LSE_FIX));
        }
        if (tid == 0) {
            LseAcc t = {NEG, 0.0};
            for (int w = 0; w < n_waves; w++) lse_merge(t, part[2 * w], part[2 * w + 1]);
            d.col_total[h.col0 + k] = lse_value(t);
        }
        __syncthreads();
        if (!first) {
            for (int m = tid; m < pc.n_merge; m += T) {
                const double v = finish(m);
                if (CUR_LDS) cur[m] = v;
                d.merge_b[pc.mcell_off + m] = v;
            }
        } else if (tid == 0) d.hmm_fb[2 * hmm_index + 1] = finish(0);
        __syncthreads();
    }
}

hipError_t mrp_launch_sweep_lse(const MrpBatchDev &d, const int32_t *order_dev, int64_t n, int max_merge, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    if (max_merge < 1) max_merge = 1;
    max_merge = (max_merge + 1) & ~1; /* keeps the arrays behind it 16-byte aligned */
    const bool cur_lds = max_merge <= MRP_LSE_CUR_LDS_MAX_MERGE;
    const size_t lds = (size_t) max_merge * (cur_lds ? 20 : 12) + 16 * sizeof(double);
    static PerDeviceOnce once;
    const hipError_t attr_status = once.run([] {
        hipError_t e = hipFuncSetAttribute((const void *) mrp_sweep_lse_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *) mrp_sweep_lse_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return e;
    });
    if (attr_status != hipSuccess) return attr_status;
    if (cur_lds) hipLaunchKernelGGL(mrp_sweep_lse_kernel<true>, dim3((unsigned) n), dim3(512), lds, stream, d, order_dev, max_merge);
    else hipLaunchKernelGGL(mrp_sweep_lse_kernel<false>, dim3((unsigned) n), dim3(512), lds, stream, d, order_dev, max_merge);
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
