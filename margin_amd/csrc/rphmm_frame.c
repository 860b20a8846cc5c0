/*
 * rphmm_frame.c -- what frames the read-partitioning path on either side, in C on
 * the host as in the reference: it is bookkeeping on a few thousand values per chunk, not a kernel.
 *
 *   f-2  input builder: bubble graph -> packed uint8 profile pool + site tables in the layout
 *        mrp_chunk_create / mrp_read take directly (bubbleGraph_getProfileSeqs bubbleGraph.c:2356-2441,
 *        bubbleGraph_getReference :2443-2474).
 *   f-4  after the path: read -> haplotype assignment with its phred score (genomeFragment.c:71-100, :234-276),
 *        the cis / trans decision when a chunk is stitched to its predecessors (stitching.c:244-403) and the
 *        rules that start a new phase set in the output VCF (vcf.c:869-945, bubbleGraph.c:2860-2883).
 *
 * Arithmetic that the reference leaves to undefined float -> integer conversions is pinned to what its x86-64
 * build does (cvttss2si): documented at each place.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "rphmm_host.h"

static void *fmalloc(size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p) abort();
    return p;
}
static void *fcalloc(size_t n, size_t s) {
    void *p = calloc(n ? n : 1, s ? s : 1);
    if (!p) abort();
    return p;
}

/* sonLib stMath_logAddExact (call sites bubbleGraph.c:2425, genomeFragment.c:98) */
static double log_add_exact(double x, double y) {
    if (x == -INFINITY) return y;
    if (y == -INFINITY) return x;
    return x > y ? x + log(1.0 + exp(y - x)) : y + log(1.0 + exp(x - y));
}

/* (int64_t) of a float as x86-64 converts it: out-of-range and NaN give INT64_MIN */
static int64_t f32_to_i64_x86(float v) {
    if (!(v >= -9223372036854775808.0f && v < 9223372036854775808.0f)) return INT64_MIN;
    return (int64_t) v;
}
/* (uint16_t) of a float as gcc/x86-64 converts it: cvttss2si to 32 bits, then the low 16 bits */
static uint16_t f32_to_u16_x86(float v) {
    if (!(v >= -2147483648.0f && v < 2147483648.0f)) return 0; /* 0x80000000 & 0xFFFF */
    return (uint16_t) (uint32_t) (int32_t) v;
}

/* ------------------------------------------------------------------------------------------ */
/* f-2: input builder                                                                          */
/* ------------------------------------------------------------------------------------------ */
static int check_bubbles(const mrp_bubbles *bg) {
    if (!bg || bg->n_bubbles < 0) return mrp_set_error(MRP_ERR_ARG, "bubble graph is NULL");
    if (bg->n_bubbles > 0 && (!bg->allele_number || !bg->read_off || !bg->support_off)) return mrp_set_error(MRP_ERR_ARG, "bubble graph arrays missing");
    for (int64_t i = 0; i < bg->n_bubbles; i++) {
        const int64_t nr = bg->read_off[i + 1] - bg->read_off[i];
        if (bg->allele_number[i] == 0 || nr < 0 || bg->support_off[i + 1] - bg->support_off[i] != (int64_t) bg->allele_number[i] * nr)
            return mrp_set_error(MRP_ERR_ARG, "bubble %lld: inconsistent allele / read counts", (long long) i);
    }
    return MRP_OK;
}

int mrp_reference_from_bubbles(const mrp_bubbles *bg, double het_substitution_probability, uint32_t **allele_number_out,
                               uint16_t **substitution_out, uint16_t **prior_out) {
    int rc = check_bubbles(bg);
    if (rc != MRP_OK) return rc;
    if (!allele_number_out || !substitution_out || !prior_out) return mrp_set_error(MRP_ERR_ARG, "mrp_reference_from_bubbles: NULL output");
    int64_t n_alleles = 0, n_sub = 0;
    for (int64_t i = 0; i < bg->n_bubbles; i++) { n_alleles += bg->allele_number[i]; n_sub += (int64_t) bg->allele_number[i] * bg->allele_number[i]; }
    uint32_t *an = fmalloc(sizeof(uint32_t) * (size_t) bg->n_bubbles);
    uint16_t *sub = fcalloc((size_t) n_sub, sizeof(uint16_t));
    uint16_t *prior = fcalloc((size_t) n_alleles, sizeof(uint16_t)); /* "these are all set equal": zero (bubbleGraph.c:2458) */
    /* bubbleGraph.c:2461-2467: off-diagonal = roundf(-log(hetSubstitutionProbability) * PROFILE_PROB_SCALAR) stored in a
     * uint16_t.  With the shipped hetSubstitutionProbability = 0 that is +inf, whose conversion yields 0 on x86-64. */
    const uint16_t off = f32_to_u16_x86(roundf((float) (-log(het_substitution_probability) * 30.0)));
    int64_t o = 0;
    for (int64_t i = 0; i < bg->n_bubbles; i++) {
        const uint32_t A = bg->allele_number[i];
        an[i] = A;
        for (uint32_t j = 0; j < A; j++)
            for (uint32_t k = 0; k < A; k++) sub[o + (int64_t) j * A + k] = j == k ? 0 : off;
        o += (int64_t) A * A;
    }
    *allele_number_out = an; *substitution_out = sub; *prior_out = prior;
    return MRP_OK;
}

int mrp_profile_seqs_from_bubbles(const mrp_bubbles *bg, int64_t n_reads, const char *const *read_names,
                                  const int32_t *forward_strand, mrp_read **seqs_out, int32_t **read_of_seq_out,
                                  int64_t *n_seqs_out, uint8_t **pool_out, int64_t *pool_bytes_out) {
    int rc = check_bubbles(bg);
    if (rc != MRP_OK) return rc;
    if (n_reads < 0 || !seqs_out || !read_of_seq_out || !n_seqs_out || !pool_out || !pool_bytes_out)
        return mrp_set_error(MRP_ERR_ARG, "mrp_profile_seqs_from_bubbles: bad arguments");
    const int64_t nb = bg->n_bubbles;
    /* first and last bubble of every read (bubbleGraph.c:2359-2381), sequences in order of first appearance */
    int64_t *first = fmalloc(sizeof(int64_t) * (size_t) n_reads), *last = fmalloc(sizeof(int64_t) * (size_t) n_reads);
    int32_t *order = fmalloc(sizeof(int32_t) * (size_t) n_reads), *seq_of = fmalloc(sizeof(int32_t) * (size_t) n_reads);
    for (int64_t r = 0; r < n_reads; r++) { first[r] = -1; last[r] = -1; }
    int64_t n_seqs = 0;
    for (int64_t i = 0; i < nb; i++)
        for (int64_t j = bg->read_off[i]; j < bg->read_off[i + 1]; j++) {
            const int32_t r = bg->reads[j];
            if (r < 0 || r >= n_reads) { free(first); free(last); free(order); free(seq_of); return mrp_set_error(MRP_ERR_ARG, "bubble %lld: read index %d out of range", (long long) i, r); }
            if (first[r] < 0) { first[r] = i; seq_of[r] = (int32_t) n_seqs; order[n_seqs++] = r; }
            last[r] = i;
        }
    /* allele offsets of the bubbles = stSite.alleleOffset (bubbleGraph.c:2455) */
    int64_t *allele_off = fmalloc(sizeof(int64_t) * (size_t) (nb + 1));
    allele_off[0] = 0;
    for (int64_t i = 0; i < nb; i++) allele_off[i + 1] = allele_off[i] + bg->allele_number[i];
    mrp_read *seqs = fcalloc((size_t) n_seqs, sizeof(*seqs));
    int64_t pool_bytes = 0;
    for (int64_t s = 0; s < n_seqs; s++) { /* stProfileSeq_constructEmptyProfile profileSeq.c:13-29 */
        const int32_t r = order[s];
        seqs[s].name = read_names ? read_names[r] : NULL;
        seqs[s].ref_start = (int32_t) first[r];
        seqs[s].length = (int32_t) (last[r] - first[r] + 1);
        seqs[s].forward_strand = forward_strand ? forward_strand[r] : 1;
        seqs[s].pool_offset = pool_bytes;
        pool_bytes += allele_off[last[r] + 1] - allele_off[first[r]];
    }
    uint8_t *pool = fcalloc((size_t) pool_bytes, 1); /* sites the read skips stay 0 */
    for (int64_t i = 0; i < nb; i++) {
        const int64_t nr = bg->read_off[i + 1] - bg->read_off[i];
        const uint32_t A = bg->allele_number[i];
        const float *sup = bg->allele_read_supports + bg->support_off[i];
        for (int64_t j = 0; j < nr; j++) {
            const int32_t r = bg->reads[bg->read_off[i] + j];
            const mrp_read *q = &seqs[seq_of[r]];
            /* normalising constant (:2421-2426) */
            double total = -INFINITY;
            for (uint32_t k = 0; k < A; k++) total = log_add_exact(total, (double) sup[(int64_t) nr * k + j]);
            uint8_t *dst = pool + q->pool_offset + (allele_off[i] - allele_off[q->ref_start]);
            for (uint32_t k = 0; k < A; k++) { /* :2429-2435 */
                const float lp = sup[(int64_t) nr * k + j];
                const int64_t l = f32_to_i64_x86(roundf((float) (30.0 * (total - (double) lp))));
                dst[k] = (uint8_t) (l > 255 ? 255 : l);
            }
        }
    }
    free(first); free(last); free(seq_of); free(allele_off);
    *seqs_out = seqs; *read_of_seq_out = order; *n_seqs_out = n_seqs; *pool_out = pool; *pool_bytes_out = pool_bytes;
    return MRP_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* f-4: read -> haplotype assignment                                                           */
/* ------------------------------------------------------------------------------------------ */
/* getLogProbOfReadGivenHaplotype genomeFragment.c:71-89 */
static double read_given_haplotype(const int64_t *allele_offset, const uint8_t *pool, const mrp_read *r, const uint64_t *hap,
                                   int64_t start, int64_t length) {
    double total = 0.0;
    const int64_t first = allele_offset[r->ref_start];
    for (int32_t i = 0; i < r->length; i++) {
        const int64_t j = (int64_t) i + r->ref_start - start;
        if (j >= 0 && j < length) total -= pool[r->pool_offset + (allele_offset[i + r->ref_start] - first) + (int64_t) hap[j]];
    }
    return total / 30.0;
}

int mrp_assign_reads_to_haplotypes(int64_t n_sites, const uint32_t *allele_number, const uint8_t *profile_pool,
                                   const mrp_read *reads, int64_t n_reads, const mrp_phase_result *gf, int64_t min_phred,
                                   int8_t *hap_out, double *phred_out) {
    if (n_sites < 0 || (n_sites > 0 && !allele_number) || (n_reads > 0 && (!reads || !profile_pool)) || !gf || !hap_out)
        return mrp_set_error(MRP_ERR_ARG, "mrp_assign_reads_to_haplotypes: NULL argument");
    int64_t *allele_offset = fmalloc(sizeof(int64_t) * (size_t) (n_sites + 1));
    allele_offset[0] = 0;
    for (int64_t i = 0; i < n_sites; i++) allele_offset[i + 1] = allele_offset[i] + allele_number[i];
    for (int64_t i = 0; i < n_reads; i++)
        if (reads[i].ref_start < 0 || reads[i].length < 0 || (int64_t) reads[i].ref_start + reads[i].length > n_sites) {
            free(allele_offset);
            return mrp_set_error(MRP_ERR_ARG, "read %lld lies outside the sites", (long long) i);
        }
    for (int64_t i = 0; i < n_reads; i++) { hap_out[i] = -1; if (phred_out) phred_out[i] = 0.0; }
    for (int side = 2; side >= 1; side--) { /* a read found in both sets counts as hap1 (:253) */
        const int32_t *set = side == 1 ? gf->reads1 : gf->reads2;
        const int64_t n = side == 1 ? gf->n_reads1 : gf->n_reads2;
        const uint64_t *mine = side == 1 ? gf->haplotype_string1 : gf->haplotype_string2;
        const uint64_t *other = side == 1 ? gf->haplotype_string2 : gf->haplotype_string1;
        for (int64_t q = 0; q < n; q++) {
            const int32_t r = set[q];
            if (r < 0 || r >= n_reads) { free(allele_offset); return mrp_set_error(MRP_ERR_ARG, "genome fragment names read %d", r); }
            /* :255-259 -- as written in the reference the first haplotype handed to getLogProbabilityOfBeingInPartition
             * for a hap1 read is haplotypeString2: the score is that of the OTHER haplotype having generated the read */
            const double a = read_given_haplotype(allele_offset, profile_pool, &reads[r], other, gf->ref_start, gf->length);
            const double b = read_given_haplotype(allele_offset, profile_pool, &reads[r], mine, gf->ref_start, gf->length);
            const double lp = a - log_add_exact(a, b); /* :91-100 */
            const double phred = -10 * lp / 2.302585;  /* :260 */
            if (phred_out) phred_out[r] = phred;
            hap_out[r] = phred < (double) min_phred ? 0 : (int8_t) side;
        }
    }
    free(allele_offset);
    return MRP_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* f-4: stitching adjacent chunks (stitching.c:244-403)                                        */
/* ------------------------------------------------------------------------------------------ */
typedef struct { char **key; double *val; uint8_t *used; int64_t cap, n; } name_map; /* insertion-ordered slots, tombstones */
static uint64_t str_hash(const char *s) { uint64_t h = 1469598103934665603ull; while (*s) { h ^= (unsigned char) *s++; h *= 1099511628211ull; } return h; }
typedef struct { name_map m; int64_t *slot; int64_t slot_cap; } name_table;
static void nt_init(name_table *t) { memset(t, 0, sizeof(*t)); }
static void nt_free(name_table *t) {
    for (int64_t i = 0; i < t->m.n; i++) free(t->m.key[i]);
    free(t->m.key); free(t->m.val); free(t->m.used); free(t->slot);
    memset(t, 0, sizeof(*t));
}
static void nt_rehash(name_table *t, int64_t cap) {
    free(t->slot);
    t->slot_cap = cap;
    t->slot = fmalloc(sizeof(int64_t) * (size_t) cap);
    for (int64_t i = 0; i < cap; i++) t->slot[i] = -1;
    for (int64_t i = 0; i < t->m.n; i++)
        if (t->m.used[i]) {
            uint64_t s = str_hash(t->m.key[i]) & (uint64_t) (cap - 1);
            while (t->slot[s] >= 0) s = (s + 1) & (uint64_t) (cap - 1);
            t->slot[s] = i;
        }
}
static int64_t nt_find(const name_table *t, const char *k) {
    if (t->slot_cap == 0) return -1;
    uint64_t s = str_hash(k) & (uint64_t) (t->slot_cap - 1);
    while (t->slot[s] != -1) {
        const int64_t i = t->slot[s];
        if (i >= 0 && t->m.used[i] && strcmp(t->m.key[i], k) == 0) return i;
        s = (s + 1) & (uint64_t) (t->slot_cap - 1);
    }
    return -1;
}
static void nt_put(name_table *t, const char *k, double v) { /* key must be absent */
    if ((t->m.n + 1) * 2 > t->slot_cap) nt_rehash(t, t->slot_cap ? t->slot_cap * 2 : 64);
    if (t->m.n == t->m.cap) {
        t->m.cap = t->m.cap ? t->m.cap * 2 : 32;
        t->m.key = realloc(t->m.key, sizeof(char *) * (size_t) t->m.cap);
        t->m.val = realloc(t->m.val, sizeof(double) * (size_t) t->m.cap);
        t->m.used = realloc(t->m.used, (size_t) t->m.cap);
        if (!t->m.key || !t->m.val || !t->m.used) abort();
    }
    const int64_t i = t->m.n++;
    t->m.key[i] = strdup(k); t->m.val[i] = v; t->m.used[i] = 1;
    uint64_t s = str_hash(k) & (uint64_t) (t->slot_cap - 1);
    while (t->slot[s] >= 0) s = (s + 1) & (uint64_t) (t->slot_cap - 1);
    t->slot[s] = i;
}
static void nt_remove(name_table *t, int64_t i) { t->m.used[i] = 0; } /* the slot keeps pointing at a dead entry */
static int64_t nt_size(const name_table *t) { int64_t n = 0; for (int64_t i = 0; i < t->m.n; i++) n += t->m.used[i]; return n; }

struct mrp_stitch { name_table hap1, hap2; };

int mrp_stitch_create(mrp_stitch **out) {
    if (!out) return mrp_set_error(MRP_ERR_ARG, "mrp_stitch_create: NULL");
    mrp_stitch *s = fcalloc(1, sizeof(*s));
    nt_init(&s->hap1); nt_init(&s->hap2);
    *out = s;
    return MRP_OK;
}
void mrp_stitch_destroy(mrp_stitch *s) {
    if (!s) return;
    nt_free(&s->hap1); nt_free(&s->hap2);
    free(s);
}
int64_t mrp_stitch_size(const mrp_stitch *s, int hap) { return s ? nt_size(hap == 1 ? &s->hap1 : &s->hap2) : 0; }
int mrp_stitch_lookup(const mrp_stitch *s, int hap, const char *name, double *prob) {
    const name_table *t = hap == 1 ? &s->hap1 : &s->hap2;
    const int64_t i = nt_find(t, name);
    if (i < 0) return 0;
    if (prob) *prob = t->m.val[i];
    return 1;
}

/* sizeOfIntersection / sizeOfIntersectionWithNonNegativeValues stitching.c:306-343 */
static int64_t intersection(const name_table *seen, int64_t n, const char *const *names, const double *probs, int primary_only) {
    int64_t c = 0;
    for (int64_t i = 0; i < n; i++) {
        if (primary_only && probs[i] < 0) continue;
        const int64_t j = nt_find(seen, names[i]);
        if (j < 0) continue;
        if (primary_only && seen->m.val[j] < 0) continue;
        c++;
    }
    return c;
}
/* addToHapReadsSeen stitching.c:244-283 */
static void add_seen(name_table *hap, name_table *other, int64_t n, const char *const *names, const double *probs) {
    for (int64_t i = 0; i < n; i++) {
        int64_t j = nt_find(other, names[i]);
        if (j >= 0) {
            if (probs[i] > other->m.val[j]) nt_remove(other, j);
            else continue;
        }
        j = nt_find(hap, names[i]);
        if (j < 0) nt_put(hap, names[i], probs[i]);
        else if (probs[i] > hap->m.val[j]) hap->m.val[j] = probs[i];
    }
}

int mrp_stitch_chunk(mrp_stitch *s, int64_t n1, const char *const *names1, const double *probs1, int64_t n2,
                     const char *const *names2, const double *probs2, int primary_reads_only, int do_not_switch,
                     int *switched, int64_t counts[4]) {
    if (!s || n1 < 0 || n2 < 0 || (n1 > 0 && (!names1 || !probs1)) || (n2 > 0 && (!names2 || !probs2)) || !switched)
        return mrp_set_error(MRP_ERR_ARG, "mrp_stitch_chunk: bad arguments");
    /* stitching.c:354-362 */
    const int64_t cisH1 = intersection(&s->hap1, n1, names1, probs1, primary_reads_only);
    const int64_t cisH2 = intersection(&s->hap2, n2, names2, probs2, primary_reads_only);
    const int64_t transH1 = intersection(&s->hap2, n1, names1, probs1, primary_reads_only);
    const int64_t transH2 = intersection(&s->hap1, n2, names2, probs2, primary_reads_only);
    if (counts) { counts[0] = cisH1; counts[1] = cisH2; counts[2] = transH1; counts[3] = transH2; }
    *switched = 0;
    if (cisH1 + cisH2 < transH1 + transH2 && !do_not_switch) *switched = 1; /* :380-394 */
    if (*switched) { /* the chunk's read sets trade places (:388, :390) */
        add_seen(&s->hap1, &s->hap2, n2, names2, probs2);
        add_seen(&s->hap2, &s->hap1, n1, names1, probs1);
    } else {
        add_seen(&s->hap1, &s->hap2, n1, names1, probs1);
        add_seen(&s->hap2, &s->hap1, n2, names2, probs2);
    }
    return MRP_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* f-4: phase sets (vcf.c:869-945)                                                             */
/* ------------------------------------------------------------------------------------------ */
/* bionomialCoefficient / binomialPValue bubbleGraph.c:2860-2883 (unsigned 128-bit arithmetic) */
static unsigned __int128 binomial_coefficient(int64_t n, int64_t k) {
    unsigned __int128 ans = 1;
    k = k > n - k ? n - k : k;
    for (int64_t j = 1; j <= k; j++, n--) {
        if (n % j == 0) ans *= (unsigned __int128) (n / j);
        else if (ans % (unsigned __int128) j == 0) ans = ans / (unsigned __int128) j * (unsigned __int128) n;
        else ans = (ans * (unsigned __int128) n) / (unsigned __int128) j;
    }
    return ans;
}
double mrp_binomial_coefficient(int64_t n, int64_t k, uint64_t *hi, uint64_t *lo) {
    if (n < 0 || k < 0 || k > n) { if (hi) *hi = 0; if (lo) *lo = 0; return 0.0; }
    const unsigned __int128 c = binomial_coefficient(n, k);
    if (hi) *hi = (uint64_t) (c >> 64);
    if (lo) *lo = (uint64_t) c;
    return (double) c;
}
double mrp_binomial_p_value(int64_t n, int64_t k) {
    unsigned __int128 j = 0;
    k = k < n / 2 ? n - k : k;
    for (int64_t i = k; i <= n; i++) j += binomial_coefficient(n, i);
    return (double) j / pow(2.0, (double) n);
}

static int64_t set_intersection(const int32_t *a, int64_t na, const int32_t *b, int64_t nb) {
    int64_t c = 0;
    for (int64_t i = 0; i < na; i++)
        for (int64_t j = 0; j < nb; j++)
            if (a[i] == b[j]) { c++; break; }
    return c;
}

int mrp_phase_sets(int64_t n_variants, const mrp_variant *v, int64_t min_spanning_reads, double min_binomial_read_split_likelihood,
                   double max_discordant_ratio, int32_t *phase_set_out, int32_t *reason_out) {
    if (n_variants < 0 || (n_variants > 0 && (!v || !phase_set_out))) return mrp_set_error(MRP_ERR_ARG, "mrp_phase_sets: bad arguments");
    const mrp_variant *prev_het = NULL, *curr = NULL;
    int32_t phase_set = -1;
    for (int64_t i = 0; i < n_variants; i++) {
        if (curr != NULL && curr->gt1 != curr->gt2) prev_het = curr; /* :869-872 */
        curr = &v[i];
        const int gt1 = curr->gt1, gt2 = curr->gt2;
        if (gt1 >= curr->n_alleles || gt2 >= curr->n_alleles) return mrp_set_error(MRP_ERR_ARG, "variant %lld: genotype outside its alleles", (long long) i);
        int64_t c1 = -1, c2 = -1, d1 = -1, d2 = -1;
        int determined = 0;
        if (prev_het != NULL && gt1 != gt2 && prev_het->gt1 >= 0 && gt1 >= 0) { /* :901-911 */
            if (gt2 < 0 || prev_het->gt2 < 0) return mrp_set_error(MRP_ERR_ARG, "variant %lld: negative second genotype", (long long) i);
#define READS_OF(var, g) ((var)->allele_reads + (var)->allele_read_off[g]), ((var)->allele_read_off[(g) + 1] - (var)->allele_read_off[g])
            c1 = set_intersection(READS_OF(prev_het, prev_het->gt1), READS_OF(curr, gt1));
            c2 = set_intersection(READS_OF(prev_het, prev_het->gt2), READS_OF(curr, gt2));
            d1 = set_intersection(READS_OF(prev_het, prev_het->gt2), READS_OF(curr, gt1));
            d2 = set_intersection(READS_OF(prev_het, prev_het->gt1), READS_OF(curr, gt2));
#undef READS_OF
            determined = 1;
        }
        int reason = MRP_PS_SAME;
        if (gt1 != gt2 && prev_het == NULL) reason = MRP_PS_NO_HET; /* :916-919 */
        else if (determined) {
            if (c1 + c2 < min_spanning_reads) reason = MRP_PS_MISSING_CONCORDANCY;                                          /* :922 */
            else if (mrp_binomial_p_value(c1 + c2, c1) < min_binomial_read_split_likelihood) reason = MRP_PS_UNLIKELY_CONCORDANCY; /* :927 */
            else if (1.0 * (double) (d1 + d2) / (double) (c1 + c2 + d1 + d2) > max_discordant_ratio) reason = MRP_PS_DISCORDANCY;  /* :932 */
        }
        if (reason != MRP_PS_SAME) phase_set = curr->pos; /* :941-945 */
        phase_set_out[i] = gt1 != gt2 ? phase_set : -1;   /* writePhaseSet :946-953 */
        if (reason_out) reason_out[i] = reason;
    }
    return MRP_OK;
}
