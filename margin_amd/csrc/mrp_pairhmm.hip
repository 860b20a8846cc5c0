/*
 * mrp_pairhmm.hip -- read x allele alignment likelihoods: the banded pair-HMM forward probability of the reference
 * (computeForwardProbability, impl/pairwiseAligner.c:849-903) for batches of string pairs, and the alleleReadSupports
 * loop around it (impl/bubbleGraph.c:1421-1464).  gfx950 only; compiled with -ffp-contract=off.
 *
 * The recursion (stateMachine3_cellCalculate, impl/stateMachine.c:562-586) gives every dp cell (x, y) three states from
 * its neighbours (x-1, y), (x-1, y-1), (x, y-1); a neighbour outside the band contributes nothing, which is what a
 * neighbour holding LOG_ZERO contributes, so the kernels keep -inf where the reference keeps NULL.  Each state is a
 * chain of three logAdd (pairwiseAligner.c:279-299: cubic interpolation in fp64, float literals, no exp / log) in the
 * reference's order.  Two mappings:
 *
 *   phm_lane_kernel   a pair per LANE, for pairs whose x string has at most 100 symbols and no anchors (their band is
 *                     the whole matrix): the lane walks its matrix row by row, the previous row lives in LDS as
 *                     row[x][state][lane] (conflict-free), the column x = 0 in registers.  All 64 lanes do useful
 *                     work in every step when the pairs of a wave have similar shapes (the host sorts them), which
 *                     is the case that matters: margin phase aligns ~25-symbol alleles to ~25-symbol read substrings,
 *                     10^5 pairs per 1 Mb chunk.  LDS per wave = 1 536 B x (longest x string of the launch class):
 *                     25 symbols -> 4 waves per CU, one per SIMD.
 *   phm_wave_kernel   a pair per WAVE for everything else (long strings, anchored bands): x+y diagonal by diagonal, a
 *                     lane per cell of the diagonal, the last two diagonals in LDS.
 *
 * Bound: fp64 VALU issue (six logAdd per cell); the kernels move ~0.1 B per flop.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "../../include/margin_rphmm.h"
#include "mrp_internal.h"
#include "rphmm_host.h"

#pragma clang fp contract(off)

namespace {

constexpr int PHM_WAVE = 64;
constexpr int PHM_LANE_MAX_X = 100;    /* 100 * 1 600 B = 156 KB of the 160 KB, the rest holds the tables */
constexpr int PHM_LDS_BYTES = 160 * 1024;
#ifndef PHM_ROWS
#define PHM_ROWS 2 /* rows a lane of the pair-per-lane kernel advances together (measured: 1 -> 2.49 ms, 2 -> 2.11, 3 -> 2.75, 4 -> 2.61, 6 -> 3.23 for 4.8e5 pairs) */
#endif
constexpr int PHM_LANE_BYTES_PER_X = 3 * 64 * 8 + 64; /* LDS per wave and x position: three states per lane + the lane's symbol */
constexpr int PHM_ETAB = 25 * 6;       /* doubles per model in the emission + transition table */
constexpr int PHM_WAVE_MAX_WIDTH = 2048; /* 3 diagonals * 2 048 cells * 3 states * 8 B = 144 KB */

struct PhmModelDev {
    double t[9];     /* order of mrp_pair_hmm */
    double em[25];   /* [cx * 5 + cy], N rows / columns hold log(0.25^2) as written in stateMachine.c:380 */
    double ex[5], ey[5];
    double start[3]; /* stateMachine3_startStateProb / raggedStartStateProb */
    double end[3];   /* stateMachine3_endStateProb / raggedEndStateProb */
};

struct PhmPair {
    int64_t x_off, y_off;
    int64_t band_off; /* first diagonal in the band array, -1: whole matrix */
    int32_t lx, ly, model, out;
};

struct PhmLanePair { /* pair-per-lane kernel: no band */
    int64_t x_off, y_off;
    int32_t lx, ly, model, out;
};

struct St {
    double m, x, y;
};

#define PHM_NEG (-__builtin_inf())

/* lookup(), pairwiseAligner.c:282-293: the float literals are rounded to float first, as the C compiler does */
static __device__ __forceinline__ double phm_lookup(double x) {
    const bool a = x <= 1.0, b = x <= 2.5, c = x <= 4.5;
    const double c3 = a ? (double) -0.009350833524763f : b ? (double) -0.014532321752540f : c ? (double) -0.004605031767994f : (double) -0.000458661602210f;
    const double c2 = a ? (double) 0.130659527668286f : b ? (double) 0.139942324101744f : c ? (double) 0.063427417320019f : (double) 0.009695946122598f;
    const double c1 = a ? (double) 0.498799810682272f : b ? (double) 0.495635523139337f : c ? (double) 0.695956496475118f : (double) 0.930734667215156f;
    const double c0 = a ? (double) 0.693203116424741f : b ? (double) 0.692140569840976f : c ? (double) 0.514272634594009f : (double) 0.168037164329057f;
    return ((c3 * x + c2) * x + c1) * x + c0;
}
/* logAdd(), pairwiseAligner.c:295-299 */
static __device__ __forceinline__ double phm_log_add(double x, double y) {
    const bool lt = x < y;
    const double hi = lt ? y : x, lo = lt ? x : y;
    const double d = hi - lo; /* NaN for two LOG_ZEROs: not used, lo == LOG_ZERO decides first */
    return (lo == PHM_NEG || d >= 7.5) ? hi : phm_lookup(d) + lo;
}
/* The same two functions with the coefficients of the interval read from LDS (coef[interval][c3, c2, c1, c0]) instead
 * of selected with 24 v_cndmask: the pair-per-lane kernel is bound by VALU issue, its LDS pipe is mostly idle */
__device__ const float PHM_COEF[16] = {-0.009350833524763f, 0.130659527668286f, 0.498799810682272f, 0.693203116424741f,
                                       -0.014532321752540f, 0.139942324101744f, 0.495635523139337f, 0.692140569840976f,
                                       -0.004605031767994f, 0.063427417320019f, 0.695956496475118f, 0.514272634594009f,
                                       -0.000458661602210f, 0.009695946122598f, 0.930734667215156f, 0.168037164329057f};
/* -DPHM_COEF_SELECT: the coefficients are float literals, select the float's bits (3 v_cndmask each) and widen */
static __device__ __forceinline__ double phm_lookup_t(double x, const double *coef) {
#ifdef PHM_COEF_SELECT
    const bool a = x <= 1.0, b = x <= 2.5, c = x <= 4.5;
    auto sel = [&](float f0, float f1, float f2, float f3) {
        uint32_t v = c ? __float_as_uint(f2) : __float_as_uint(f3);
        v = b ? __float_as_uint(f1) : v;
        v = a ? __float_as_uint(f0) : v;
        double d;
        asm("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(v));
        return d;
    };
    const double c3 = sel(-0.009350833524763f, -0.014532321752540f, -0.004605031767994f, -0.000458661602210f);
    const double c2 = sel(0.130659527668286f, 0.139942324101744f, 0.063427417320019f, 0.009695946122598f);
    const double c1 = sel(0.498799810682272f, 0.495635523139337f, 0.695956496475118f, 0.930734667215156f);
    const double c0 = sel(0.693203116424741f, 0.692140569840976f, 0.514272634594009f, 0.168037164329057f);
    return ((c3 * x + c2) * x + c1) * x + c0;
#else
    const int idx = (x > 1.0 ? 1 : 0) + (x > 2.5 ? 1 : 0) + (x > 4.5 ? 1 : 0);
    const double2 a = *reinterpret_cast<const double2 *>(coef + idx * 4);
    const double2 b = *reinterpret_cast<const double2 *>(coef + idx * 4 + 2);
    return ((a.x * x + a.y) * x + b.x) * x + b.y;
#endif
}
/* Branch-free on purpose: the interpolation is evaluated for every lane and thrown away where logAdd returns the larger
 * operand (d = inf or NaN then selects a valid table row and produces a value nobody reads).  With the conditional
 * written around the interpolation the compiler emits one basic block per logAdd, each waiting for its own LDS read, and
 * the single wave of a SIMD sits idle for the latency six times per cell; as straight-line code the three state chains
 * of a cell interleave. */
static __device__ __forceinline__ double phm_log_add_t(double x, double y, const double *coef) {
    /* the larger and the smaller operand (for x == y the reference takes hi = x, lo = y: the same two numbers) */
    const double hi = __builtin_fmax(x, y), lo = __builtin_fmin(x, y);
    const double d = hi - lo; /* inf if lo is LOG_ZERO, NaN if both are */
    const double v = phm_lookup_t(d, coef) + lo;
    return !(d < 7.5) ? hi : v; /* lo == LOG_ZERO || d >= 7.5 */
}
/* HAS_SWITCH = false: every model has TRANSITION_GAP_SWITCH_TO_X / _Y = log(0) (the shipped margin parameters), the
 * third term of the gap chains is LOG_ZERO and logAdd(a, LOG_ZERO) == a */
template <bool THIRD>
static __device__ __forceinline__ double phm_chain_t(double a, double b, double c, const double *coef) {
    const double ab = phm_log_add_t(a, b, coef);
    return THIRD ? phm_log_add_t(ab, c, coef) : ab;
}

/* toCells[to] = logAdd(toCells[to], from + (eP + tP)) three times, starting from LOG_ZERO (logAdd(LOG_ZERO, a) == a) */
static __device__ __forceinline__ double phm_chain(double a, double b, double c) { return phm_log_add(phm_log_add(a, b), c); }

struct PhmRowT { /* eP + tP of the gap-y transitions of a row (y fixed) */
    double open, extend, sw;
};
static __device__ __forceinline__ St phm_cell(const St &lower, const St &middle, const St &upper, const double *t, double eX, double eM,
                                              const PhmRowT &ty) {
    St cur;
    cur.x = phm_chain(lower.m + (eX + t[3]), lower.x + (eX + t[5]), lower.y + (eX + t[7]));
    cur.m = phm_chain(middle.m + (eM + t[0]), middle.x + (eM + t[1]), middle.y + (eM + t[2]));
    cur.y = phm_chain(upper.m + ty.open, upper.y + ty.extend, upper.x + ty.sw);
    return cur;
}
/* cell_dotProduct, pairwiseAligner.c:333-339 */
static __device__ __forceinline__ double phm_dot(const St &f, const double *e) {
    double tot = f.m + e[0];
    tot = phm_log_add(tot, f.x + e[1]);
    return phm_log_add(tot, f.y + e[2]);
}
static __device__ __forceinline__ int phm_wave_max(int v) {
    for (int o = 32; o >= 1; o >>= 1) {
        const int w = __shfl_xor(v, o, PHM_WAVE);
        v = w > v ? w : v;
    }
    return v;
}

/* LDS: coef[16] | etab[n_models][cx][cy][6] = eX + {open, extend, switch to x}, eM + {continue, from x, from y} |
 * rows[wave][x - 1][state][lane].  The waves of a workgroup share the tables and nothing else.
 *
 * A lane walks its matrix R rows at a time, row r one column behind row r - 1 (x_r = k - r in step k), so the R cells
 * of a step depend only on cells of earlier steps: left = the row's own cell of step k - 1, up / diagonal = the cells
 * of the row above from steps k - 1 / k - 2, all in registers.  Only the first row of a pass reads the LDS row (the last
 * row of the previous pass) and only the last row writes it.  With one wave per SIMD this is what hides the latency of
 * the dependent fp64 chains and of the table reads: R * 3 independent chains per step instead of 3.
 * Cells beyond a lane's own strings are computed and ignored: nothing flows from larger x or y to smaller. */
template <int R, bool HAS_SWITCH>
__global__ void __launch_bounds__(256) phm_lane_kernel(const PhmLanePair *__restrict__ pairs, int64_t n_pairs, const uint8_t *__restrict__ pool,
                                                       const PhmModelDev *__restrict__ models, int n_models, int cap, double *__restrict__ out) {
    extern __shared__ double phm_lds[];
    double *coef = phm_lds;
    double *etab = phm_lds + 16;
    const int lane = threadIdx.x & (PHM_WAVE - 1), wave = threadIdx.x / PHM_WAVE, n_waves = blockDim.x / PHM_WAVE;
    uint8_t *wave_base = reinterpret_cast<uint8_t *>(etab + (size_t) n_models * PHM_ETAB) + (size_t) wave * cap * PHM_LANE_BYTES_PER_X;
    double *row = reinterpret_cast<double *>(wave_base) + lane;          /* [x - 1][state][lane] */
    uint8_t *symx = wave_base + (size_t) cap * 3 * PHM_WAVE * sizeof(double) + lane; /* [x - 1][lane]: the lane's x string */
    if (threadIdx.x < 16) coef[threadIdx.x] = (double) PHM_COEF[threadIdx.x];
    for (int i = threadIdx.x; i < n_models * 25; i += blockDim.x) {
        const PhmModelDev *__restrict__ Mi = models + i / 25;
        const int c = i % 25;
        const double eX = Mi->ex[c / 5], eM = Mi->em[c];
        double *e = etab + (size_t) i * 6;
        e[0] = eX + Mi->t[3];
        e[1] = eX + Mi->t[5];
        e[2] = eX + Mi->t[7];
        e[3] = eM + Mi->t[0];
        e[4] = eM + Mi->t[1];
        e[5] = eM + Mi->t[2];
    }
    __syncthreads();
    const int64_t pi = ((int64_t) blockIdx.x * n_waves + wave) * PHM_WAVE + lane;
    const bool have = pi < n_pairs;
    PhmLanePair p;
    if (have) p = pairs[pi];
    else { p.x_off = 0; p.y_off = 0; p.lx = -1; p.ly = -1; p.model = 0; p.out = 0; }
    const int lx = p.lx, ly = p.ly;
    const int mx = phm_wave_max(lx), my = phm_wave_max(ly);
    const int mx1 = mx > 1 ? mx : 1;
    const PhmModelDev *__restrict__ M = models + p.model;
    const double t_open_y = M->t[4], t_extend_y = M->t[6], t_switch_y = M->t[8];
    const uint8_t *__restrict__ sx = pool + p.x_off;
    const uint8_t *__restrict__ sy = pool + p.y_off;
    /* no global loads inside the step loop (the compiler waits for each one on the spot): the x string goes to LDS */
    for (int x = 1; x <= mx1; x++) {
        int c = 4;
        if (x <= lx) { c = sx[x - 1]; c = c > 4 ? 4 : c; }
        symx[(x - 1) * PHM_WAVE] = (uint8_t) c;
#pragma unroll
        for (int s = 0; s < 3; s++) row[((x - 1) * 3 + s) * PHM_WAVE] = PHM_NEG;
    }
    const St neg{PHM_NEG, PHM_NEG, PHM_NEG};
    St b_prev = neg; /* cell (0, y0 - 1) */
    St fin = neg;    /* cell (lx, ly) */
    const double *__restrict__ etab_m = etab + (size_t) p.model * PHM_ETAB;
    int cy_next[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        cy_next[r] = 4;
        if (r >= 1 && r <= ly) { cy_next[r] = sy[r - 1]; cy_next[r] = cy_next[r] > 4 ? 4 : cy_next[r]; }
    }
    for (int y0 = 0; y0 <= my; y0 += R) {
        const double *etab_r[R];
        double ty_open[R], ty_extend[R], ty_switch[R];
        St b0[R]; /* cell (0, y0 + r): only the gap-y state can be reached */
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int y = y0 + r;
            const int cy = cy_next[r];
            /* the y symbols of the next pass are requested now and waited for after the step loop */
            cy_next[r] = 4;
            if (y + R <= ly) { cy_next[r] = sy[y + R - 1]; cy_next[r] = cy_next[r] > 4 ? 4 : cy_next[r]; }
            const double eY = M->ey[cy];
            ty_open[r] = eY + t_open_y;
            ty_extend[r] = eY + t_extend_y;
            ty_switch[r] = eY + t_switch_y;
            etab_r[r] = etab_m + cy * 6;
            const St above = r == 0 ? b_prev : b0[r > 0 ? r - 1 : 0];
            if (y == 0) b0[r] = St{M->start[0], M->start[1], M->start[2]};
            else b0[r] = St{PHM_NEG, PHM_NEG, phm_chain_t<HAS_SWITCH>(above.m + ty_open[r], above.y + ty_extend[r], above.x + ty_switch[r], coef)};
            if (lx == 0 && y == ly) fin = b0[r];
        }
        St c1[R], c2[R]; /* the row's cells of the last two steps */
        int cxs[R];
#pragma unroll
        for (int r = 0; r < R; r++) { c1[r] = neg; c2[r] = neg; cxs[r] = 4; }
        St up0_prev = b_prev;
        for (int k = 0; k <= mx + R - 1; k++) {
            /* one basic block: no branch between the R cells, so that their chains interleave */
            const int kc = (k < 1 ? 1 : (k > mx1 ? mx1 : k)) - 1; /* steps 0 and > mx read a valid slot and ignore it */
            const double *r0 = row + (size_t) kc * 3 * PHM_WAVE;
            const St up0{r0[0], r0[PHM_WAVE], r0[2 * PHM_WAVE]};
            const int c0 = symx[kc * PHM_WAVE];
#pragma unroll
            for (int r = R - 1; r >= 1; r--) cxs[r] = cxs[r - 1];
            cxs[0] = (k >= 1 && k <= lx) ? c0 : 4;
            St nw[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int x = k - r;
                const St left = c1[r];
                const St up = r == 0 ? up0 : c1[r > 0 ? r - 1 : 0];
                const St diag = r == 0 ? up0_prev : c2[r > 0 ? r - 1 : 0];
                const double2 *__restrict__ e = reinterpret_cast<const double2 *>(etab_r[r] + cxs[r] * 30);
                const double2 e01 = e[0], e23 = e[1], e45 = e[2];
                St cur;
                cur.x = phm_chain_t<HAS_SWITCH>(left.m + e01.x, left.x + e01.y, left.y + e23.x, coef);
                cur.m = phm_chain_t<true>(diag.m + e23.y, diag.x + e45.x, diag.y + e45.y, coef);
                cur.y = phm_chain_t<HAS_SWITCH>(up.m + ty_open[r], up.y + ty_extend[r], up.x + ty_switch[r], coef);
                const bool last = x == lx && y0 + r == ly && lx >= 1;
                fin.m = last ? cur.m : fin.m;
                fin.x = last ? cur.x : fin.x;
                fin.y = last ? cur.y : fin.y;
                nw[r].m = x == 0 ? b0[r].m : cur.m;
                nw[r].x = x == 0 ? b0[r].x : cur.x;
                nw[r].y = x == 0 ? b0[r].y : cur.y;
            }
            const int xl = k - (R - 1);
            if (xl >= 1 && xl <= mx) {
                double *rl = row + (size_t) (xl - 1) * 3 * PHM_WAVE;
                rl[0] = nw[R - 1].m;
                rl[PHM_WAVE] = nw[R - 1].x;
                rl[2 * PHM_WAVE] = nw[R - 1].y;
            }
#pragma unroll
            for (int r = 0; r < R; r++) { c2[r] = c1[r]; c1[r] = nw[r]; }
            up0_prev = k == 0 ? b_prev : up0;
        }
        b_prev = b0[R - 1];
    }
    if (have) {
        const double e_end[3] = {M->end[0], M->end[1], M->end[2]};
        out[p.out] = (lx == 0 && ly == 0) ? 0.0 : phm_dot(fin, e_end); /* :860-862; the last diagonal holds one cell */
    }
}

__global__ void __launch_bounds__(PHM_WAVE) phm_wave_kernel(const PhmPair *__restrict__ pairs, int64_t n_pairs, const uint8_t *__restrict__ pool,
                                                              const PhmModelDev *__restrict__ models, const int32_t *__restrict__ band, int W,
                                                              double *__restrict__ out) {
    extern __shared__ double phm_diag[]; /* 3 diagonals x W cells x 3 states */
    const int lane = threadIdx.x;
    for (int64_t pi = blockIdx.x; pi < n_pairs; pi += gridDim.x) {
        const PhmPair p = pairs[pi];
        const int lx = p.lx, ly = p.ly, n = lx + ly;
        if (n == 0) {
            if (lane == 0) out[p.out] = 0.0;
            continue;
        }
        const PhmModelDev *__restrict__ M = models + p.model;
        double t[9];
#pragma unroll
        for (int i = 0; i < 9; i++) t[i] = M->t[i];
        const uint8_t *__restrict__ sx = pool + p.x_off;
        const uint8_t *__restrict__ sy = pool + p.y_off;
        double *d0 = phm_diag, *d1 = phm_diag + 3 * W, *d2 = phm_diag + 6 * W; /* being written, xay - 1, xay - 2 */
        auto limits = [&](int d, int &l, int &r) {
            if (p.band_off >= 0) {
                l = band[2 * (p.band_off + d)];
                r = band[2 * (p.band_off + d) + 1];
            } else {
                const int xlo = d - ly > 0 ? d - ly : 0, xhi = d < lx ? d : lx;
                l = 2 * xlo - d;
                r = 2 * xhi - d;
            }
        };
        int l1, r1, l2 = 1, r2 = 0;
        limits(0, l1, r1);
        for (int i = lane; i <= (r1 - l1) / 2; i += PHM_WAVE) {
            d1[3 * i] = M->start[0];
            d1[3 * i + 1] = M->start[1];
            d1[3 * i + 2] = M->start[2];
        }
        __syncthreads();
        for (int d = 1; d <= n; d++) {
            int l, r;
            limits(d, l, r);
            const int width = (r - l) / 2 + 1;
            for (int i = lane; i < width; i += PHM_WAVE) {
                const int xmy = l + 2 * i;
                const int x = (d + xmy) >> 1, y = (d - xmy) >> 1;
                St lower{PHM_NEG, PHM_NEG, PHM_NEG}, middle = lower, upper = lower;
                if (xmy - 1 >= l1 && xmy - 1 <= r1) { const double *c = d1 + 3 * ((xmy - 1 - l1) >> 1); lower = St{c[0], c[1], c[2]}; }
                if (xmy + 1 >= l1 && xmy + 1 <= r1) { const double *c = d1 + 3 * ((xmy + 1 - l1) >> 1); upper = St{c[0], c[1], c[2]}; }
                if (xmy >= l2 && xmy <= r2) { const double *c = d2 + 3 * ((xmy - l2) >> 1); middle = St{c[0], c[1], c[2]}; }
                int cx = 4, cy = 4;
                if (x > 0) { cx = sx[x - 1]; cx = cx > 4 ? 4 : cx; }
                if (y > 0) { cy = sy[y - 1]; cy = cy > 4 ? 4 : cy; }
                const double eY = M->ey[cy];
                const PhmRowT ty{eY + t[4], eY + t[6], eY + t[8]};
                const St cur = phm_cell(lower, middle, upper, t, M->ex[cx], M->em[cx * 5 + cy], ty);
                d0[3 * i] = cur.m;
                d0[3 * i + 1] = cur.x;
                d0[3 * i + 2] = cur.y;
            }
            __syncthreads();
            double *tmp = d2;
            d2 = d1; d1 = d0; d0 = tmp;
            l2 = l1; r2 = r1; l1 = l; r1 = r;
        }
        /* diagonalCalculationTotalProbability on the last diagonal (:578-596, no diagonal beyond it): dpDiagonal_dotProduct */
        if (lane == 0) {
            const double e_end[3] = {M->end[0], M->end[1], M->end[2]};
            double tot = PHM_NEG;
            for (int i = 0; i <= (r1 - l1) / 2; i++) tot = phm_log_add(tot, phm_dot(St{d1[3 * i], d1[3 * i + 1], d1[3 * i + 2]}, e_end));
            out[p.out] = tot;
        }
        __syncthreads();
    }
}

/* ---------------- host ---------------- */

int fail(int code, const char *msg) { return mrp_set_error(code, "%s", msg); }

#define PHM_HIP(expr)                                                                                                  \
    do {                                                                                                               \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess) return mrp_set_error(MRP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));       \
    } while (0)

/* band_construct in closed form.  Between two consecutive anchor points P = (px, py) and N = (nx, ny) (matrix
 * coordinates; the first P is (0, 0), the last N is (lx, ly)) the reference bounds the diagonals xay in (px + py,
 * nx + ny] by xL = px - e/2, yL = ny + e/2, xU = nx + e/2, yU = py - e/2 (clamped to the matrix), :218-221; on such a
 * diagonal band_setCurrentDiagonal (:96-114) yields the smallest xmy of the right parity with xmy >= xL - yL, x >= xL,
 * y <= yL and the largest with xmy <= (xU - yU rounded UP to the parity), x <= xU, y >= yU. */
int band_closed_form(const int64_t *anchors, int64_t n_anchors, int64_t lx, int64_t ly, int64_t expansion, int32_t *L, int32_t *R,
                     int64_t *cells, int *max_width) {
    if (lx < 0 || ly < 0 || expansion < 0 || expansion % 2 != 0) return MRP_ERR_ARG;
    const int64_t e2 = expansion / 2;
    auto clamp = [](int64_t z, int64_t hi) { return z < 0 ? (int64_t) 0 : (z > hi ? hi : z); };
    L[0] = 0;
    R[0] = 0;
    int64_t total = 1;
    int mw = 1;
    int64_t px = 0, py = 0, ai = 0;
    while (px + py < lx + ly) {
        int64_t nx = lx, ny = ly;
        if (ai < n_anchors) {
            nx = anchors[2 * ai] + 1;
            ny = anchors[2 * ai + 1] + 1;
            ai++;
            if (!(nx > px && ny > py && nx <= lx && ny <= ly)) return MRP_ERR_ARG; /* asserts :206-211 */
        }
        const int64_t xL = clamp(px - e2, lx), yL = clamp(ny + e2, ly), xU = clamp(nx + e2, lx), yU = clamp(py - e2, ly);
        for (int64_t d = px + py + 1; d <= nx + ny; d++) {
            int64_t l = xL - yL, r = xU - yU;
            if ((d + l) % 2 != 0) l++;
            if ((d + r) % 2 != 0) r++;
            l = std::max(l, std::max(2 * xL - d, d - 2 * yL));
            r = std::min(r, std::min(2 * xU - d, d - 2 * yU));
            if (l > r) return MRP_ERR_ARG; /* diagonal_construct :22-27 throws */
            L[d] = (int32_t) l;
            R[d] = (int32_t) r;
            const int64_t w = (r - l) / 2 + 1;
            total += w;
            if (w > mw) mw = (int) w;
        }
        px = nx;
        py = ny;
    }
    /* the reference ignores anchors left over once (lx, ly) has been reached only if there are none: an anchor list that
     * runs past the end fails its asserts, an anchor exactly at (lx - 1, ly - 1) followed by nothing is fine */
    if (ai < n_anchors) return MRP_ERR_ARG;
    if (cells) *cells = total;
    if (max_width) *max_width = mw;
    return MRP_OK;
}

void model_to_device(const mrp_pair_hmm &m, int ragged_left, int ragged_right, PhmModelDev &d) {
    const double *t = &m.match_continue;
    for (int i = 0; i < 9; i++) d.t[i] = t[i];
    for (int x = 0; x < 5; x++)
        for (int y = 0; y < 5; y++) d.em[x * 5 + y] = (x >= 4 || y >= 4) ? -2.772588722 : m.e_match[x * 4 + y]; /* stateMachine.c:378-383 */
    for (int s = 0; s < 5; s++) {
        d.ex[s] = s >= 4 ? -1.386294361 : m.e_gap_x[s]; /* :363-368 */
        d.ey[s] = s >= 4 ? -1.386294361 : m.e_gap_y[s];
    }
    const double ninf = -INFINITY;
    /* stateMachine.c:521-560 */
    d.start[0] = ragged_left ? ninf : 0.0;
    d.start[1] = ragged_left ? 0.0 : ninf;
    d.start[2] = ragged_left ? 0.0 : ninf;
    if (ragged_right) {
        d.end[0] = (m.gap_open_x + m.gap_open_y) / 2.0;
        d.end[1] = m.gap_extend_x;
        d.end[2] = m.gap_extend_y;
    } else {
        d.end[0] = m.match_continue;
        d.end[1] = m.match_from_gap_x;
        d.end[2] = m.match_from_gap_y;
    }
}

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

const int WAVE_CLASS_CAP[4] = {64, 256, 1024, PHM_WAVE_MAX_WIDTH};

/* getKmerAlignmentAnchors (pairwiseAligner.c:1563-1627) with KMER_SIZE = 20 (:1519): first occurrence of every k-mer of x
 * (getKmers :1543-1555), then over the k-mers of y found among them, in order of y, the best chain with increasing x; the
 * walk back over earlier pairs stops at the first chainable one that was a running maximum (:1592).  Appends (x, y). */
constexpr int64_t PHM_KMER = 20;
int64_t kmer_anchors(const uint8_t *sx, int64_t lx, const uint8_t *sy, int64_t ly, std::vector<int64_t> &out) {
    if (PHM_KMER > lx || PHM_KMER > ly) return 0;
    std::unordered_map<std::string_view, int64_t> first;
    first.reserve((size_t) (lx - PHM_KMER + 1) * 2);
    for (int64_t i = 0; i + PHM_KMER <= lx; i++) first.emplace(std::string_view((const char *) sx + i, (size_t) PHM_KMER), i);
    struct ChainPair { int64_t x, y, score, back; bool high; };
    std::vector<ChainPair> cp;
    int64_t max_score = 0, max_pair = -1;
    for (int64_t y = 0; y + PHM_KMER <= ly; y++) {
        auto it = first.find(std::string_view((const char *) sy + y, (size_t) PHM_KMER));
        if (it == first.end()) continue;
        ChainPair c{it->second, y, 1, -1, false};
        for (int64_t j = (int64_t) cp.size() - 1; j >= 0; j--) {
            if (cp[(size_t) j].x < c.x) {
                if (cp[(size_t) j].score + 1 > c.score) { c.score = cp[(size_t) j].score + 1; c.back = j; }
                if (cp[(size_t) j].high) break;
            }
        }
        if (c.score >= max_score) { c.high = true; max_score = c.score; max_pair = (int64_t) cp.size(); }
        cp.push_back(c);
    }
    int64_t n = 0;
    for (int64_t q = max_pair; q != -1; q = cp[(size_t) q].back) n++;
    const size_t base = out.size();
    out.resize(base + 2 * (size_t) n);
    int64_t w = n;
    for (int64_t q = max_pair; q != -1; q = cp[(size_t) q].back) {
        w--;
        out[base + 2 * (size_t) w] = cp[(size_t) q].x + PHM_KMER / 2;
        out[base + 2 * (size_t) w + 1] = cp[(size_t) q].y + PHM_KMER / 2;
    }
    return n;
}

}  // namespace

extern "C" {

void mrp_symbols_from_chars(const char *s, int64_t n, uint8_t *out) {
    for (int64_t i = 0; i < n; i++) {
        switch (s[i]) {
            case 'A': case 'a': out[i] = 0; break;
            case 'C': case 'c': out[i] = 1; break;
            case 'G': case 'g': out[i] = 2; break;
            case 'T': case 't': out[i] = 3; break;
            default: out[i] = 4;
        }
    }
}

void mrp_pair_hmm_reverse_complement(mrp_pair_hmm *m) {
    for (int i = 0; i < 4; i++)
        for (int j = i + 1; j < 4; j++) std::swap(m->e_match[i * 4 + j], m->e_match[(3 - i) * 4 + (3 - j)]);
    std::swap(m->e_match[0], m->e_match[15]);
    std::swap(m->e_match[5], m->e_match[10]);
    for (int i = 0; i < 2; i++) {
        std::swap(m->e_gap_x[i], m->e_gap_x[3 - i]);
        std::swap(m->e_gap_y[i], m->e_gap_y[3 - i]);
    }
}

int64_t mrp_kmer_alignment_anchors(const uint8_t *x, int64_t lx, const uint8_t *y, int64_t ly, int64_t *out) {
    if (!out || lx < 0 || ly < 0 || (lx > 0 && !x) || (ly > 0 && !y)) return 0;
    std::vector<int64_t> v;
    const int64_t n = kmer_anchors(x, lx, y, ly, v);
    if (n > 0) memcpy(out, v.data(), sizeof(int64_t) * 2 * (size_t) n);
    return n;
}

int mrp_band_diagonals(const int64_t *anchors, int64_t n_anchors, int64_t lx, int64_t ly, int64_t expansion, int32_t *xmy_l, int32_t *xmy_r) {
    if (!xmy_l || !xmy_r || (n_anchors > 0 && !anchors) || n_anchors < 0) return fail(MRP_ERR_ARG, "mrp_band_diagonals: null argument");
    if (lx + ly >= (1ll << 30)) return fail(MRP_ERR_ARG, "mrp_band_diagonals: strings too long");
    const int rc = band_closed_form(anchors, n_anchors, lx, ly, expansion, xmy_l, xmy_r, nullptr, nullptr);
    return rc == MRP_OK ? rc : fail(rc, "mrp_band_diagonals: invalid anchors or expansion (pairwiseAligner.c:179,206-211)");
}

int mrp_forward_probabilities(mrp_context *ctx, const mrp_pair_hmm *models, int32_t n_models, int64_t n_pairs, const uint8_t *pool,
                              int64_t pool_bytes, const int64_t *x_off, const int32_t *x_len, const int64_t *y_off, const int32_t *y_len,
                              const uint8_t *model_index, const int64_t *anchor_off, const int64_t *anchors, int64_t expansion, int ragged_left,
                              int ragged_right, double *out, mrp_pairhmm_stats *stats) {
    const double t_begin = now_ms();
    if (stats) memset(stats, 0, sizeof(*stats));
    if (!ctx) return fail(MRP_ERR_NO_DEVICE, "mrp_forward_probabilities: no context (the pair-HMM path has no CPU fallback)");
    if (n_pairs < 0 || n_models <= 0 || !models || pool_bytes < 0) return fail(MRP_ERR_ARG, "mrp_forward_probabilities: bad sizes");
    if (n_pairs == 0) return MRP_OK;
    if (!x_off || !x_len || !y_off || !y_len || !out || (pool_bytes > 0 && !pool)) return fail(MRP_ERR_ARG, "mrp_forward_probabilities: null argument");
    if (n_pairs >= (1ll << 31)) return fail(MRP_ERR_ARG, "mrp_forward_probabilities: more than 2^31 pairs in one call");
    if (expansion < 0 || expansion % 2 != 0) return fail(MRP_ERR_ARG, "mrp_forward_probabilities: diagonalExpansion must be even (pairwiseAligner.c:855)");

    /* classify.  The pair-per-lane kernel takes the unanchored pairs whose x string fits its LDS row next to the tables;
     * launch classes by x length (4, 3, 2, 1 waves per workgroup = per CU). */
    const int table_bytes = (16 + n_models * PHM_ETAB) * (int) sizeof(double);
    /* (-1: the tables of this many models leave no room for a row, every pair goes to the pair-per-wave kernel) */
    const int lane_max_x = PHM_LDS_BYTES - table_bytes < PHM_LANE_BYTES_PER_X ? -1 : std::min(PHM_LANE_MAX_X, (PHM_LDS_BYTES - table_bytes) / PHM_LANE_BYTES_PER_X);
    int lane_cap[4];
    for (int c = 0; c < 4; c++) lane_cap[c] = std::min(lane_max_x, (PHM_LDS_BYTES - table_bytes) / ((4 - c) * PHM_LANE_BYTES_PER_X));
    constexpr uint32_t WAVE_KEY = 0xFFFFFFFFu;
    constexpr int LY_CLIP = 4095;
    HostVec<uint32_t> key((size_t) n_pairs);
    std::atomic<int64_t> bad{-1}, cells_atomic{0};
    mrp_parallel_for((n_pairs + 16383) / 16384, 1, [&](int64_t blk) {
        int64_t c_local = 0;
        for (int64_t i = blk * 16384; i < std::min(n_pairs, (blk + 1) * 16384); i++) {
            const int64_t lx = x_len[i], ly = y_len[i];
            const int mi = model_index ? model_index[i] : 0;
            const int64_t na = anchor_off ? anchor_off[i + 1] - anchor_off[i] : 0;
            if (lx < 0 || ly < 0 || x_off[i] < 0 || y_off[i] < 0 || x_off[i] + lx > pool_bytes || y_off[i] + ly > pool_bytes || mi >= n_models || na < 0 ||
                (na > 0 && !anchors)) {
                int64_t expect = -1;
                bad.compare_exchange_strong(expect, i);
                key[(size_t) i] = WAVE_KEY;
                continue;
            }
            if (na == 0 && lx <= lane_max_x) {
                key[(size_t) i] = (uint32_t) (std::min<int64_t>(ly, LY_CLIP) << 7 | lx);
                c_local += (lx + 1) * (ly + 1);
            } else {
                key[(size_t) i] = WAVE_KEY;
            }
        }
        cells_atomic += c_local;
    });
    if (bad.load() >= 0)
        return mrp_set_error(MRP_ERR_ARG, "mrp_forward_probabilities: pair %lld lies outside the symbol pool, names a model >= %d or has bad anchor offsets",
                             (long long) bad.load(), n_models);
    int64_t cells = cells_atomic.load();
    /* pairs of similar shape share a wave (counting sort on (y length, x length), longest first so that the tail of a
     * launch is made of the cheap ones) */
    HostVec<PhmLanePair> lane_pairs[4];
    int64_t lane_n[4] = {0, 0, 0, 0};
    {
        std::vector<int32_t> hist((size_t) (LY_CLIP + 1) << 7, 0);
        for (int64_t i = 0; i < n_pairs; i++)
            if (key[(size_t) i] != WAVE_KEY) hist[key[(size_t) i]]++;
        int cls_of[128];
        for (int lx = 0; lx < 128; lx++) {
            int c = 0;
            while (c < 3 && lx > lane_cap[c]) c++;
            cls_of[lx] = c;
        }
        for (int64_t b = (int64_t) hist.size() - 1; b >= 0; b--) {
            const int32_t h = hist[(size_t) b];
            if (!h) continue;
            const int c = cls_of[b & 127];
            hist[(size_t) b] = (int32_t) lane_n[c];
            lane_n[c] += h;
        }
        for (int c = 0; c < 4; c++) lane_pairs[c].resize((size_t) lane_n[c]);
        for (int64_t i = 0; i < n_pairs; i++) {
            const uint32_t k = key[(size_t) i];
            if (k == WAVE_KEY) continue;
            PhmLanePair &q = lane_pairs[cls_of[k & 127]][(size_t) hist[k]++];
            q.x_off = x_off[i];
            q.y_off = y_off[i];
            q.lx = x_len[i];
            q.ly = y_len[i];
            q.model = model_index ? model_index[i] : 0;
            q.out = (int32_t) i;
        }
    }
    HostVec<PhmPair> wave_pairs[4];
    HostVec<int32_t> band;
    std::vector<int32_t> L, R;
    for (int64_t i = 0; i < n_pairs; i++) {
        if (key[(size_t) i] != WAVE_KEY) continue;
        const int64_t lx = x_len[i], ly = y_len[i];
        const int64_t na = anchor_off ? anchor_off[i + 1] - anchor_off[i] : 0;
        PhmPair p;
        p.x_off = x_off[i];
        p.y_off = y_off[i];
        p.band_off = -1;
        p.lx = (int32_t) lx;
        p.ly = (int32_t) ly;
        p.model = model_index ? model_index[i] : 0;
        p.out = (int32_t) i;
        int width;
        if (na == 0) {
            width = (int) std::min(lx, ly) + 1;
            cells += (lx + 1) * (ly + 1);
        } else {
            if (lx + ly >= (1ll << 30)) return fail(MRP_ERR_ARG, "mrp_forward_probabilities: strings too long");
            L.resize((size_t) (lx + ly + 1));
            R.resize((size_t) (lx + ly + 1));
            int64_t c = 0;
            const int rc = band_closed_form(anchors + 2 * anchor_off[i], na, lx, ly, expansion, L.data(), R.data(), &c, &width);
            if (rc != MRP_OK) return mrp_set_error(rc, "mrp_forward_probabilities: pair %lld has invalid anchors (pairwiseAligner.c:206-211)", (long long) i);
            cells += c;
            p.band_off = (int64_t) band.size() / 2;
            for (int64_t d = 0; d <= lx + ly; d++) { band.push_back(L[(size_t) d]); band.push_back(R[(size_t) d]); }
        }
        if (width > PHM_WAVE_MAX_WIDTH)
            return mrp_set_error(MRP_ERR_UNSUPPORTED, "mrp_forward_probabilities: pair %lld has a diagonal of %d cells (limit %d)", (long long) i, width, PHM_WAVE_MAX_WIDTH);
        int c = 0;
        while (width > WAVE_CLASS_CAP[c]) c++;
        wave_pairs[c].push_back(p);
    }
    for (auto &v : wave_pairs)
        std::sort(v.begin(), v.end(), [](const PhmPair &a, const PhmPair &b) {
            const int64_t ca = (int64_t) a.lx * a.ly, cb = (int64_t) b.lx * b.ly;
            return ca != cb ? ca > cb : a.out < b.out;
        });

    PHM_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    std::vector<PhmModelDev> hm((size_t) n_models);
    bool has_switch = false;
    for (int i = 0; i < n_models; i++) {
        model_to_device(models[i], ragged_left, ragged_right, hm[(size_t) i]);
        if (!(models[i].gap_switch_to_x == -INFINITY && models[i].gap_switch_to_y == -INFINITY)) has_switch = true;
    }
    DevBuf<PhmModelDev> d_models;
    DevBuf<uint8_t> d_pool;
    DevBuf<int32_t> d_band;
    DevBuf<double> d_out;
    DevBuf<PhmLanePair> d_lane[4];
    DevBuf<PhmPair> d_wave[4];
    d_models.pool = d_pool.pool = d_band.pool = d_out.pool = &ctx->pool;
    /* declared after every host vector the queued copies read: an early return drains the stream before they are destroyed */
    struct Drain { hipStream_t s; ~Drain() { (void) hipStreamSynchronize(s); } } drain{s};
    PHM_HIP(d_models.upload(hm, s));
    PHM_HIP(d_pool.alloc((size_t) pool_bytes));
    if (pool_bytes) PHM_HIP(hipMemcpyAsync(d_pool.p, pool, (size_t) pool_bytes, hipMemcpyHostToDevice, s));
    PHM_HIP(d_band.upload(band, s));
    PHM_HIP(d_out.alloc((size_t) n_pairs));
    for (int c = 0; c < 4; c++) {
        d_lane[c].pool = d_wave[c].pool = &ctx->pool;
        PHM_HIP(d_lane[c].upload(lane_pairs[c], s));
        PHM_HIP(d_wave[c].upload(wave_pairs[c], s));
    }
    /* once per device (contexts of several host threads may call concurrently) */
    static PerDeviceOnce once;
    const hipError_t configured = once.run([] {
        hipError_t e = hipFuncSetAttribute((const void *) phm_lane_kernel<PHM_ROWS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PHM_LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *) phm_lane_kernel<PHM_ROWS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, PHM_LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *) phm_wave_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PHM_LDS_BYTES);
        return e;
    });
    PHM_HIP(configured);
    /* kernel_ms must not contain the tail of the uploads (the copy engine finishes them behind the event otherwise) */
    if (stats) PHM_HIP(hipStreamSynchronize(s));
    PHM_HIP(hipEventRecord(ctx->ev[0], s));
    for (int c = 0; c < 4; c++) {
        const int64_t n = (int64_t) lane_pairs[c].size();
        if (n == 0) continue;
        int cap = 1;
        for (const PhmLanePair &p : lane_pairs[c]) cap = std::max(cap, (int) p.lx);
        const int row_bytes = cap * PHM_LANE_BYTES_PER_X;
        const int nw = std::max(1, std::min(4, (PHM_LDS_BYTES - table_bytes) / row_bytes));
        const size_t lds = (size_t) table_bytes + (size_t) nw * row_bytes;
        const int64_t per_wg = (int64_t) nw * PHM_WAVE;
        if (has_switch)
            hipLaunchKernelGGL((phm_lane_kernel<PHM_ROWS, true>), dim3((unsigned) ((n + per_wg - 1) / per_wg)), dim3((unsigned) per_wg), lds, s, d_lane[c].p, n, d_pool.p,
                               d_models.p, (int) n_models, cap, d_out.p);
        else
            hipLaunchKernelGGL((phm_lane_kernel<PHM_ROWS, false>), dim3((unsigned) ((n + per_wg - 1) / per_wg)), dim3((unsigned) per_wg), lds, s, d_lane[c].p, n, d_pool.p,
                               d_models.p, (int) n_models, cap, d_out.p);
        PHM_HIP(hipGetLastError());
        if (stats) stats->pairs_lane += n;
    }
    for (int c = 0; c < 4; c++) {
        const int64_t n = (int64_t) wave_pairs[c].size();
        if (n == 0) continue;
        const int W = WAVE_CLASS_CAP[c];
        const size_t lds = (size_t) 9 * W * sizeof(double);
        hipLaunchKernelGGL(phm_wave_kernel, dim3((unsigned) std::min<int64_t>(n, 16384)), dim3(PHM_WAVE), lds, s, d_wave[c].p, n, d_pool.p,
                           d_models.p, d_band.p, W, d_out.p);
        PHM_HIP(hipGetLastError());
        if (stats) stats->pairs_wave += n;
    }
    PHM_HIP(hipEventRecord(ctx->ev[1], s));
    PHM_HIP(hipMemcpyAsync(out, d_out.p, (size_t) n_pairs * sizeof(double), hipMemcpyDeviceToHost, s));
    PHM_HIP(hipStreamSynchronize(s));
    if (stats) {
        float ms = 0.f;
        PHM_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
        stats->kernel_ms = ms;
        stats->cells = cells;
    }
    d_models.release(); d_pool.release(); d_band.release(); d_out.release();
    for (auto &b : d_lane) b.release();
    for (auto &b : d_wave) b.release();
    ctx->pool.reclaim();
    if (stats) stats->total_ms = now_ms() - t_begin;
    return MRP_OK;
}

int mrp_allele_read_supports(mrp_context *ctx, const mrp_pair_hmm *forward_model, const mrp_pair_hmm *reverse_model, int64_t n_bubbles,
                             const int64_t *allele_first, const int64_t *read_first, const uint8_t *pool, int64_t pool_bytes,
                             const int64_t *allele_off, const int32_t *allele_len, const int64_t *read_off, const int32_t *read_len,
                             const uint8_t *read_forward_strand, int64_t expansion, int64_t sv_threshold, float *support,
                             mrp_pairhmm_stats *stats) {
    if (stats) memset(stats, 0, sizeof(*stats));
    if (!ctx) return fail(MRP_ERR_NO_DEVICE, "mrp_allele_read_supports: no context (the pair-HMM path has no CPU fallback)");
    if (n_bubbles < 0) return fail(MRP_ERR_ARG, "mrp_allele_read_supports: bad sizes");
    if (n_bubbles == 0) return MRP_OK;
    if (!forward_model || !reverse_model || !allele_first || !read_first || !allele_off || !allele_len || !read_off || !read_len ||
        !read_forward_strand || !support || (pool_bytes > 0 && !pool))
        return fail(MRP_ERR_ARG, "mrp_allele_read_supports: null argument");
    const int64_t n_reads_total = read_first[n_bubbles];
    for (int64_t k = 0; k < n_reads_total; k++)
        if (read_len[k] < 0 || read_off[k] < 0 || read_off[k] + read_len[k] > pool_bytes) return fail(MRP_ERR_ARG, "mrp_allele_read_supports: read substring outside the pool");
    /* cachedScores (bubbleGraph.c:1418,1431-1441): the first read of the bubble with a given substring owns the scores */
    std::vector<int64_t> owner((size_t) n_reads_total);
    mrp_parallel_for(n_bubbles, 64, [&](int64_t b) {
        const int64_t r0 = read_first[b], r1 = read_first[b + 1];
        std::vector<int64_t> order((size_t) (r1 - r0));
        for (int64_t k = r0; k < r1; k++) order[(size_t) (k - r0)] = k;
        auto less = [&](int64_t a, int64_t c) {
            if (read_len[a] != read_len[c]) return read_len[a] < read_len[c];
            const int cmp = memcmp(pool + read_off[a], pool + read_off[c], (size_t) read_len[a]);
            return cmp != 0 ? cmp < 0 : a < c;
        };
        std::sort(order.begin(), order.end(), less);
        for (size_t i = 0; i < order.size(); i++) {
            const int64_t k = order[i];
            const bool same = i > 0 && read_len[order[i - 1]] == read_len[k] && memcmp(pool + read_off[order[i - 1]], pool + read_off[k], (size_t) read_len[k]) == 0;
            owner[(size_t) k] = same ? owner[(size_t) order[i - 1]] : k;
        }
    });
    const mrp_pair_hmm models[2] = {*forward_model, *reverse_model};
    std::vector<int64_t> xo, yo, where, anchor_off, anchors;
    std::vector<int32_t> xl, yl;
    std::vector<uint8_t> mi;
    anchor_off.push_back(0);
    std::vector<int64_t> support_first((size_t) n_bubbles + 1, 0);
    for (int64_t b = 0; b < n_bubbles; b++) {
        const int64_t na = allele_first[b + 1] - allele_first[b], nr = read_first[b + 1] - read_first[b];
        if (na < 0 || nr < 0) return fail(MRP_ERR_ARG, "mrp_allele_read_supports: offsets not ascending");
        support_first[(size_t) b + 1] = support_first[(size_t) b] + na * nr;
        for (int64_t k = read_first[b]; k < read_first[b + 1]; k++) {
            if (owner[(size_t) k] != k) continue;
            for (int64_t j = allele_first[b]; j < allele_first[b + 1]; j++) {
                xo.push_back(allele_off[j]);
                xl.push_back(allele_len[j]);
                yo.push_back(read_off[k]);
                yl.push_back(read_len[k]);
                mi.push_back(read_forward_strand[k] ? 0 : 1);
                where.push_back(support_first[(size_t) b] + (j - allele_first[b]) * nr + (k - read_first[b]));
                if (read_len[k] > sv_threshold || allele_len[j] > sv_threshold) { /* bubbleGraph.c:1448-1451 */
                    if (allele_len[j] < 0 || allele_off[j] < 0 || allele_off[j] + allele_len[j] > pool_bytes)
                        return fail(MRP_ERR_ARG, "mrp_allele_read_supports: allele outside the pool");
                    kmer_anchors(pool + allele_off[j], allele_len[j], pool + read_off[k], read_len[k], anchors);
                }
                anchor_off.push_back((int64_t) anchors.size() / 2);
            }
        }
    }
    std::vector<double> lp(xo.size());
    const int rc = mrp_forward_probabilities(ctx, models, 2, (int64_t) xo.size(), pool, pool_bytes, xo.data(), xl.data(), yo.data(), yl.data(), mi.data(),
                                             anchors.empty() ? nullptr : anchor_off.data(), anchors.empty() ? nullptr : anchors.data(), expansion, 0, 0, lp.data(), stats);
    if (rc != MRP_OK) return rc;
    for (size_t i = 0; i < lp.size(); i++) support[where[i]] = (float) lp[i];
    for (int64_t b = 0; b < n_bubbles; b++) {
        const int64_t na = allele_first[b + 1] - allele_first[b], nr = read_first[b + 1] - read_first[b];
        for (int64_t k = 0; k < nr; k++) {
            const int64_t o = owner[(size_t) (read_first[b] + k)] - read_first[b];
            if (o == k) continue;
            for (int64_t j = 0; j < na; j++) support[support_first[(size_t) b] + j * nr + k] = support[support_first[(size_t) b] + j * nr + o];
        }
    }
    return MRP_OK;
}

}  // extern "C"
