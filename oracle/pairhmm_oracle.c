/*
 * pairhmm_oracle.c -- CPU restatement of the reference's banded pair-HMM forward probability
 * (per-read x allele alignment likelihoods feeding Bubble.alleleReadSupports).
 *
 * TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 * Never linked or called by the product (margin_amd/).
 *
 * Follows the reference diagonal by diagonal, with its own memory layout (one array of cells per x+y
 * diagonal, NULL for a neighbour outside the band):
 *   band            impl/pairwiseAligner.c:86-118 (band_setCurrentDiagonal), :175-226 (band_construct)
 *   logAdd          impl/pairwiseAligner.c:279-299 (cubic interpolation with float literals, no exp/log)
 *   cell            impl/stateMachine.c:562-586 (stateMachine3_cellCalculate), :363-383 (emissions, N handling)
 *   start/end       impl/stateMachine.c:521-560
 *   forward         impl/pairwiseAligner.c:311-320,547-570 (diagonalCalculationForward)
 *   backward        impl/pairwiseAligner.c:322-331,572-576
 *   total           impl/pairwiseAligner.c:333-339,450-461,578-596
 *   driver          impl/pairwiseAligner.c:849-903 (computeForwardProbability)
 *   bubble loop     impl/bubbleGraph.c:1421-1464 (cachedScores keyed by the read substring alone)
 *   k-mer anchors   impl/pairwiseAligner.c:1519-1627 (getKmerAlignmentAnchors, for strings longer than
 *                   referenceExpansionForStructuralVariants)
 *
 * Parity pins available from the reference's own tests (tests/pairwiseAlignerTest.c): test_bands :64-127 (exact
 * diagonals), test_logAdd :129-139 (0.001), test_cell :168-197, test_diagonalDPCalculations :257-340 (forward ==
 * backward within 0.001, every diagonal total within 0.01, the four posterior match pairs of AGCG / AGTTCG),
 * test_computeForwardProbability :1153-1189 (LOG_ZERO < p <= LOG_ONE).  The forward VALUES themselves have no
 * golden vector in the reference: "parity unpinned" below those tolerances; the build's HIP path is compared
 * with this file bit for bit (the arithmetic is + and * in fp64 only, compiled without contraction).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PHO_LOG_ZERO (-INFINITY)

typedef struct {
    /* order of struct _StateMachine3, impl/stateMachine.c:507-519 */
    double match_continue, match_from_gap_x, match_from_gap_y, gap_open_x, gap_open_y, gap_extend_x, gap_extend_y,
            gap_switch_to_x, gap_switch_to_y;
    double e_match[16], e_gap_x[4], e_gap_y[4];
} pho_model;

enum { ST_MATCH = 0, ST_GAP_X = 1, ST_GAP_Y = 2 };

/* ---- logAdd, pairwiseAligner.c:279-299 ---- */
static double lookup(double x) {
    if (x <= 1.00f) return ((-0.009350833524763f * x + 0.130659527668286f) * x + 0.498799810682272f) * x + 0.693203116424741f;
    if (x <= 2.50f) return ((-0.014532321752540f * x + 0.139942324101744f) * x + 0.495635523139337f) * x + 0.692140569840976f;
    if (x <= 4.50f) return ((-0.004605031767994f * x + 0.063427417320019f) * x + 0.695956496475118f) * x + 0.514272634594009f;
    return ((-0.000458661602210f * x + 0.009695946122598f) * x + 0.930734667215156f) * x + 0.168037164329057f;
}

double pho_log_add(double x, double y) {
    if (x < y) return (x == PHO_LOG_ZERO || y - x >= 7.5) ? y : lookup(y - x) + x;
    return (y == PHO_LOG_ZERO || x - y >= 7.5) ? x : lookup(x - y) + y;
}

/* ---- band, pairwiseAligner.c:86-118,175-226; anchors are (x, y) sequence coordinates ---- */
static int64_t x_of(int64_t xay, int64_t xmy) { return (xay + xmy) / 2; }
static int64_t y_of(int64_t xay, int64_t xmy) { return (xay - xmy) / 2; }
static int64_t avoid_off_by_one(int64_t xay, int64_t xmy) { return (xay + xmy) % 2 == 0 ? xmy : xmy + 1; }
static void nudge(int64_t *xmy, int64_t i, int64_t j, int64_t k) { if (i < j) *xmy += 2 * (j - i) * k; }
static int64_t bound(int64_t z, int64_t lz) { return z < 0 ? 0 : (z > lz ? lz : z); }

int pho_band(const int64_t *anchors, int64_t n_anchors, int64_t lX, int64_t lY, int64_t expansion, int64_t *xmyL_out,
             int64_t *xmyR_out) {
    int64_t ai = 0, xay = 0, pxay = 0, pxmy = 0, nxay = 0, nxmy = 0, xL = 0, yL = 0, xU = 0, yU = 0;
    if (lX < 0 || lY < 0 || expansion % 2 != 0) return -1;
    while (xay <= lX + lY) {
        int64_t l = avoid_off_by_one(xay, xL - yL), r = avoid_off_by_one(xay, xU - yU);
        nudge(&l, x_of(xay, l), xL, 1);
        nudge(&l, yL, y_of(xay, l), 1);
        nudge(&r, xU, x_of(xay, r), -1);
        nudge(&r, y_of(xay, r), yU, -1);
        if ((xay + l) % 2 != 0 || (xay + r) % 2 != 0 || l > r) return -2; /* diagonal_construct would throw */
        xmyL_out[xay] = l;
        xmyR_out[xay] = r;
        if (nxay == xay++) {
            pxay = nxay;
            pxmy = nxmy;
            int64_t x = lX, y = lY;
            if (ai < n_anchors) {
                x = anchors[2 * ai] + 1;
                y = anchors[2 * ai + 1] + 1;
                ai++;
                if (!(x > x_of(pxay, pxmy) && y > y_of(pxay, pxmy) && x <= lX && y <= lY && x > 0 && y > 0)) return -3;
            }
            nxay = x + y;
            nxmy = x - y;
            xL = bound(x_of(pxay, pxmy - expansion), lX);
            yL = bound(y_of(nxay, nxmy - expansion), lY);
            xU = bound(x_of(nxay, nxmy + expansion), lX);
            yU = bound(y_of(pxay, pxmy + expansion), lY);
        }
    }
    return 0;
}

/* ---- emissions, stateMachine.c:363-383 ---- */
static double gap_prob(const double *t, int s) { return s >= 4 ? -1.386294361 : t[s]; }
static double match_prob(const pho_model *m, int x, int y) { return (x >= 4 || y >= 4) ? -2.772588722 : m->e_match[x * 4 + y]; }

typedef void (*transition_fn)(double *from, double *to, int f, int t, double eP, double tP);
static void fwd_t(double *from, double *to, int f, int t, double eP, double tP) { to[t] = pho_log_add(to[t], from[f] + (eP + tP)); }
static void bwd_t(double *from, double *to, int f, int t, double eP, double tP) { from[f] = pho_log_add(from[f], to[t] + (eP + tP)); }

/* stateMachine3_cellCalculate, stateMachine.c:562-586 */
static void cell(const pho_model *m, double *cur, double *lower, double *middle, double *upper, int cX, int cY, transition_fn go) {
    if (lower) {
        const double eP = gap_prob(m->e_gap_x, cX);
        go(lower, cur, ST_MATCH, ST_GAP_X, eP, m->gap_open_x);
        go(lower, cur, ST_GAP_X, ST_GAP_X, eP, m->gap_extend_x);
        go(lower, cur, ST_GAP_Y, ST_GAP_X, eP, m->gap_switch_to_x);
    }
    if (middle) {
        const double eP = match_prob(m, cX, cY);
        go(middle, cur, ST_MATCH, ST_MATCH, eP, m->match_continue);
        go(middle, cur, ST_GAP_X, ST_MATCH, eP, m->match_from_gap_x);
        go(middle, cur, ST_GAP_Y, ST_MATCH, eP, m->match_from_gap_y);
    }
    if (upper) {
        const double eP = gap_prob(m->e_gap_y, cY);
        go(upper, cur, ST_MATCH, ST_GAP_Y, eP, m->gap_open_y);
        go(upper, cur, ST_GAP_Y, ST_GAP_Y, eP, m->gap_extend_y);
        go(upper, cur, ST_GAP_X, ST_GAP_Y, eP, m->gap_switch_to_y);
    }
}

static double start_prob(int ragged, int s) { return ragged ? ((s == ST_GAP_X || s == ST_GAP_Y) ? 0.0 : PHO_LOG_ZERO) : (s == ST_MATCH ? 0.0 : PHO_LOG_ZERO); }
static double end_prob(const pho_model *m, int ragged, int s) {
    if (ragged) return s == ST_MATCH ? (m->gap_open_x + m->gap_open_y) / 2.0 : (s == ST_GAP_X ? m->gap_extend_x : m->gap_extend_y);
    return s == ST_MATCH ? m->match_continue : (s == ST_GAP_X ? m->match_from_gap_x : m->match_from_gap_y);
}

typedef struct { int64_t xay, l, r; double *cells; } diag_t;
static int64_t diag_width(const diag_t *d) { return (d->r - d->l) / 2 + 1; }
static double *diag_cell(diag_t *d, int64_t xmy) { /* dpDiagonal_getCell :425-431 */
    if (!d || !d->cells || xmy < d->l || xmy > d->r) return NULL;
    return d->cells + ((xmy - d->l) / 2) * 3;
}
static void diag_alloc(diag_t *d, int64_t xay, int64_t l, int64_t r) {
    d->xay = xay; d->l = l; d->r = r;
    const int64_t n = diag_width(d) * 3;
    d->cells = malloc(sizeof(double) * (size_t) n);
    for (int64_t i = 0; i < n; i++) d->cells[i] = PHO_LOG_ZERO;
}
static int sym_x(const uint8_t *s, int64_t xay, int64_t xmy) { const int64_t x = x_of(xay, xmy); return x > 0 ? s[x - 1] : 4; }
static int sym_y(const uint8_t *s, int64_t xay, int64_t xmy) { const int64_t y = y_of(xay, xmy); return y > 0 ? s[y - 1] : 4; }

/* diagonalCalculation :547-564 */
static void diag_calc(const pho_model *m, diag_t *d, diag_t *m1, diag_t *m2, const uint8_t *sx, const uint8_t *sy, transition_fn go) {
    for (int64_t xmy = d->l; xmy <= d->r; xmy += 2)
        cell(m, diag_cell(d, xmy), diag_cell(m1, xmy - 1), diag_cell(m2, xmy), diag_cell(m1, xmy + 1), sym_x(sx, d->xay, xmy),
             sym_y(sy, d->xay, xmy), go);
}
static double cell_dot(const double *a, const double *b) { /* :333-339 */
    double t = a[0] + b[0];
    for (int i = 1; i < 3; i++) t = pho_log_add(t, a[i] + b[i]);
    return t;
}
static double diag_dot(diag_t *a, diag_t *b) { /* :450-461 */
    double t = PHO_LOG_ZERO;
    for (int64_t xmy = a->l; xmy <= a->r; xmy += 2) t = pho_log_add(t, cell_dot(diag_cell(a, xmy), diag_cell(b, xmy)));
    return t;
}

/* computeForwardProbability :849-903.  Symbols: 0..3 = ACGT, >= 4 = N.  Returns NaN on an invalid band. */
double pho_forward_probability(const pho_model *m, const uint8_t *sx, int64_t lX, const uint8_t *sy, int64_t lY, const int64_t *anchors,
                               int64_t n_anchors, int64_t expansion, int ragged_left, int ragged_right) {
    const int64_t n = lX + lY;
    if (n == 0) return 0.0;
    int64_t *L = malloc(sizeof(int64_t) * (size_t) (n + 1)), *R = malloc(sizeof(int64_t) * (size_t) (n + 1));
    if (pho_band(anchors, n_anchors, lX, lY, expansion, L, R) != 0) { free(L); free(R); return NAN; }
    diag_t *f = calloc((size_t) n + 1, sizeof(diag_t));
    diag_alloc(&f[0], 0, L[0], R[0]);
    for (int64_t xmy = f[0].l; xmy <= f[0].r; xmy += 2)
        for (int s = 0; s < 3; s++) diag_cell(&f[0], xmy)[s] = start_prob(ragged_left, s);
    for (int64_t xay = 1; xay <= n; xay++) {
        diag_alloc(&f[xay], xay, L[xay], R[xay]);
        diag_calc(m, &f[xay], &f[xay - 1], xay >= 2 ? &f[xay - 2] : NULL, sx, sy, fwd_t);
        if (xay >= 2) { free(f[xay - 2].cells); f[xay - 2].cells = NULL; }
    }
    diag_t b;
    diag_alloc(&b, n, L[n], R[n]);
    for (int64_t xmy = b.l; xmy <= b.r; xmy += 2)
        for (int s = 0; s < 3; s++) diag_cell(&b, xmy)[s] = end_prob(m, ragged_right, s);
    const double total = diag_dot(&f[n], &b); /* diagonalCalculationTotalProbability with no diagonal n + 1 */
    free(b.cells);
    for (int64_t i = 0; i <= n; i++) free(f[i].cells);
    free(f); free(L); free(R);
    return total;
}

/* The complete matrices of tests/pairwiseAlignerTest.c:257-340 (no anchors): forward and backward totals, the total of
 * every diagonal (:578-596) and exp(f + b - total) of the match state for x, y >= 1 (posterior[(x-1) * lY + (y-1)]). */
int pho_full_matrices(const pho_model *m, const uint8_t *sx, int64_t lX, const uint8_t *sy, int64_t lY, int64_t expansion,
                      double *total_forward, double *total_backward, double *diag_totals, double *posterior) {
    const int64_t n = lX + lY;
    int64_t *L = malloc(sizeof(int64_t) * (size_t) (n + 1)), *R = malloc(sizeof(int64_t) * (size_t) (n + 1));
    if (pho_band(NULL, 0, lX, lY, expansion, L, R) != 0) { free(L); free(R); return -1; }
    diag_t *f = calloc((size_t) n + 1, sizeof(diag_t)), *b = calloc((size_t) n + 1, sizeof(diag_t));
    for (int64_t i = 0; i <= n; i++) { diag_alloc(&f[i], i, L[i], R[i]); diag_alloc(&b[i], i, L[i], R[i]); }
    for (int s = 0; s < 3; s++) { diag_cell(&f[0], 0)[s] = start_prob(0, s); diag_cell(&b[n], lX - lY)[s] = end_prob(m, 0, s); }
    for (int64_t i = 1; i <= n; i++) diag_calc(m, &f[i], &f[i - 1], i >= 2 ? &f[i - 2] : NULL, sx, sy, fwd_t);
    for (int64_t i = n; i > 0; i--) diag_calc(m, &b[i], &b[i - 1], i >= 2 ? &b[i - 2] : NULL, sx, sy, bwd_t);
    /* cell_dotProduct2 :341-347 */
    double tf = diag_cell(&f[n], lX - lY)[0] + end_prob(m, 0, 0), tb = diag_cell(&b[0], 0)[0] + start_prob(0, 0);
    for (int s = 1; s < 3; s++) { tf = pho_log_add(tf, diag_cell(&f[n], lX - lY)[s] + end_prob(m, 0, s)); tb = pho_log_add(tb, diag_cell(&b[0], 0)[s] + start_prob(0, s)); }
    *total_forward = tf;
    *total_backward = tb;
    for (int64_t i = 0; i <= n; i++) {
        double t = diag_dot(&f[i], &b[i]);
        if (i >= 1 && i + 1 <= n) {
            diag_t md;
            diag_alloc(&md, i + 1, L[i + 1], R[i + 1]);
            diag_calc(m, &md, NULL, &f[i - 1], sx, sy, fwd_t);
            t = pho_log_add(t, diag_dot(&md, &b[i + 1]));
            free(md.cells);
        }
        diag_totals[i] = t;
    }
    for (int64_t x = 1; x <= lX; x++)
        for (int64_t y = 1; y <= lY; y++)
            posterior[(x - 1) * lY + (y - 1)] = exp(diag_cell(&f[x + y], x - y)[0] + diag_cell(&b[x + y], x - y)[0] - tf);
    for (int64_t i = 0; i <= n; i++) { free(f[i].cells); free(b[i].cells); }
    free(f); free(b); free(L); free(R);
    return 0;
}

/* test_cell, tests/pairwiseAlignerTest.c:168-197: one cell with its three neighbours, forward and backward */
void pho_test_cell(const pho_model *m, int cX, int cY, double *total_forward, double *total_backward) {
    double lf[3], mf[3], uf[3], cf[3], lb[3], mb[3], ub[3], cb[3];
    for (int i = 0; i < 3; i++) {
        mf[i] = start_prob(0, i);
        mb[i] = lb[i] = ub[i] = lf[i] = uf[i] = cf[i] = PHO_LOG_ZERO;
        cb[i] = end_prob(m, 0, i);
    }
    cell(m, lf, NULL, NULL, mf, cX, cY, fwd_t);
    cell(m, uf, mf, NULL, NULL, cX, cY, fwd_t);
    cell(m, cf, lf, mf, uf, cX, cY, fwd_t);
    cell(m, cb, lb, mb, ub, cX, cY, bwd_t);
    cell(m, ub, mb, NULL, NULL, cX, cY, bwd_t);
    cell(m, lb, NULL, NULL, mb, cX, cY, bwd_t);
    double tf = cf[0] + end_prob(m, 0, 0), tb = mb[0] + start_prob(0, 0);
    for (int i = 1; i < 3; i++) { tf = pho_log_add(tf, cf[i] + end_prob(m, 0, i)); tb = pho_log_add(tb, mb[i] + start_prob(0, i)); }
    *total_forward = tf;
    *total_backward = tb;
}

/* getKmerAlignmentAnchors, impl/pairwiseAligner.c:1519-1627: the first occurrence of every k-mer of x in a hash (getKmers
 * :1543-1555, kmerKey :1524-1531), then for the k-mers of y found there, in order of y, the longest chain with increasing
 * x (the inner walk stops at the first chainable pair that was a running maximum, :1592).  Anchors are the k-mer centres.
 * out: at most ly - k + 1 pairs (x, y); returns their number. */
static uint64_t kmer_key(const uint8_t *c, int64_t k) {
    uint64_t h = 0;
    for (int64_t i = 0; i < k; i++) h = c[i] + (h << 6) + (h << 16) - h;
    return h;
}
int64_t pho_kmer_anchors(const uint8_t *sx, int64_t lX, const uint8_t *sy, int64_t lY, int64_t k, int64_t *out) {
    if (k > lX || k > lY) return 0;
    const int64_t nx = lX - k + 1, ny = lY - k + 1;
    int64_t cap = 16;
    while (cap < 2 * nx) cap *= 2;
    int64_t *tab = malloc(sizeof(int64_t) * (size_t) cap);
    for (int64_t i = 0; i < cap; i++) tab[i] = -1;
    for (int64_t i = 0; i < nx; i++) { /* first hit counts */
        uint64_t h = kmer_key(sx + i, k) & (uint64_t) (cap - 1);
        while (tab[h] >= 0 && memcmp(sx + tab[h], sx + i, (size_t) k) != 0) h = (h + 1) & (uint64_t) (cap - 1);
        if (tab[h] < 0) tab[h] = i;
    }
    typedef struct { uint64_t x, y, score; int64_t back; int high; } chain_pair;
    chain_pair *cp = malloc(sizeof(chain_pair) * (size_t) (ny > 0 ? ny : 1));
    int64_t n = 0, max_pair = -1;
    uint64_t max_score = 0;
    for (int64_t y = 0; y < ny; y++) {
        uint64_t h = kmer_key(sy + y, k) & (uint64_t) (cap - 1);
        while (tab[h] >= 0 && memcmp(sx + tab[h], sy + y, (size_t) k) != 0) h = (h + 1) & (uint64_t) (cap - 1);
        if (tab[h] < 0) continue;
        cp[n].x = (uint64_t) tab[h]; cp[n].y = (uint64_t) y; cp[n].score = 1; cp[n].back = -1;
        for (int64_t j = n - 1; j >= 0; j--) {
            if (cp[j].x < cp[n].x) {
                if (cp[j].score + 1 > cp[n].score) { cp[n].score = cp[j].score + 1; cp[n].back = j; }
                if (cp[j].high) break;
            }
        }
        if (cp[n].score >= max_score) { cp[n].high = 1; max_score = cp[n].score; max_pair = n; }
        else cp[n].high = 0;
        n++;
    }
    int64_t m = 0;
    for (int64_t q = max_pair; q != -1; q = cp[q].back) m++;
    int64_t w = m;
    for (int64_t q = max_pair; q != -1; q = cp[q].back) { /* stList_reverse: ascending */
        w--;
        out[2 * w] = (int64_t) cp[q].x + k / 2;
        out[2 * w + 1] = (int64_t) cp[q].y + k / 2;
    }
    free(cp); free(tab);
    return m;
}

/* The alleleReadSupports loop of bubbleGraph.c:1421-1464 for one bubble: support[j * n_reads + k] = (float) forward
 * probability of read substring k given allele j, with the state machine of the read's strand -- except that a read
 * whose substring equals that of an earlier read of the bubble copies that read's row (cachedScores is keyed by the
 * substring alone, so the earlier read's strand decides).  Strings longer than sv_threshold
 * (referenceExpansionForStructuralVariants) are anchored on shared 20-mers (:1448-1451). */
void pho_allele_read_supports(const pho_model *forward_model, const pho_model *reverse_model, int64_t n_alleles, const uint8_t *const *alleles,
                              const int64_t *allele_len, int64_t n_reads, const uint8_t *const *reads, const int64_t *read_len,
                              const uint8_t *read_forward_strand, int64_t expansion, int64_t sv_threshold, float *support) {
    for (int64_t k = 0; k < n_reads; k++) {
        int64_t first = -1;
        for (int64_t q = 0; q < k && first < 0; q++)
            if (read_len[q] == read_len[k] && memcmp(reads[q], reads[k], (size_t) read_len[k]) == 0) first = q;
        if (first >= 0) { /* q itself may be a copy: its row equals the row of the first read with this substring */
            for (int64_t j = 0; j < n_alleles; j++) support[j * n_reads + k] = support[j * n_reads + first];
            continue;
        }
        const pho_model *m = read_forward_strand[k] ? forward_model : reverse_model;
        for (int64_t j = 0; j < n_alleles; j++) {
            int64_t *anchors = NULL, na = 0;
            if (read_len[k] > sv_threshold || allele_len[j] > sv_threshold) {
                anchors = malloc(sizeof(int64_t) * 2 * (size_t) (read_len[k] + 1));
                na = pho_kmer_anchors(alleles[j], allele_len[j], reads[k], read_len[k], 20, anchors);
            }
            support[j * n_reads + k] = (float) pho_forward_probability(m, alleles[j], allele_len[j], reads[k], read_len[k], anchors, na, expansion, 0, 0);
            free(anchors);
        }
    }
}

/* batch entry used by the tests and by bench.py's cpu_baseline: strings in one pool */
void pho_forward_batch(const pho_model *models, int64_t n_pairs, const uint8_t *pool, const int64_t *x_off, const int32_t *x_len,
                       const int64_t *y_off, const int32_t *y_len, const uint8_t *model_index, const int64_t *anchor_off,
                       const int64_t *anchors, int64_t expansion, int ragged_left, int ragged_right, double *out) {
    for (int64_t i = 0; i < n_pairs; i++) {
        const int64_t a0 = anchor_off ? anchor_off[i] : 0, a1 = anchor_off ? anchor_off[i + 1] : 0;
        out[i] = pho_forward_probability(&models[model_index ? model_index[i] : 0], pool + x_off[i], x_len[i], pool + y_off[i], y_len[i],
                                         anchors ? anchors + 2 * a0 : NULL, a1 - a0, expansion, ragged_left, ragged_right);
    }
}
