/*
 * phase_from_strings.c -- the drop-in boundary used from plain C, end to end, no Python:
 *
 *   read substrings x alleles --mrp_allele_read_supports--> Bubble.alleleReadSupports   (bubbleGraph.c:1421-1464)
 *   --mrp_profile_seqs_from_bubbles / mrp_reference_from_bubbles--> profile bytes, site tables (bubbleGraph.c:2356-2474)
 *   --mrp_chunk_create + mrp_phase_reads_many--> haplotypes + read partition              (bubbleGraph.c:2673-2801)
 *   --mrp_assign_reads_to_haplotypes--> HP tag per read                                   (genomeFragment.c:234-276)
 *
 * on a synthetic chunk (two haplotypes that differ at every site, noisy reads).  Prints how many reads were tagged and how
 * many tags agree with the haplotype the read was drawn from (up to the global label).
 *
 *   gcc -O2 -Iinclude examples/phase_from_strings.c -Lmargin_amd/lib -lmargin_rphmm -Wl,-rpath,$PWD/margin_amd/lib -o phase_from_strings
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "margin_rphmm.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(void) { /* xorshift64* */
    rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
    return (uint32_t) ((rng_state * 0x2545F4914F6CDD1Dull) >> 32);
}
static double uni(void) { return rnd() / 4294967296.0; }

#define CHECK(call)                                                                       \
    do {                                                                                  \
        int rc_ = (call);                                                                 \
        if (rc_ != MRP_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, mrp_last_error()); return 1; } \
    } while (0)

int main(void) {
    enum { N_SITES = 200, N_READS = 150, FLANK = 12, LA = 2 * FLANK + 1 };
    mrp_context *ctx = NULL;
    CHECK(mrp_context_create(0, &ctx));

    /* the state machine of the shipped parameter file (params/base_params.json, hmmForwardStrandReadGivenReference) */
    mrp_pair_hmm fwd;
    const double tr[9] = {0.8, 0.1, 0.1, 0.5, 0.5, 0.0, 0.5, 0.0, 0.5};
    const double em[16] = {0.969, 0.005, 0.017, 0.009, 0.008, 0.973, 0.007, 0.012, 0.021, 0.007, 0.967, 0.006, 0.008, 0.008, 0.004, 0.98};
    fwd.match_continue = log(tr[0]);
    fwd.match_from_gap_x = fwd.match_from_gap_y = log((tr[3] + tr[6]) / 2.0);
    fwd.gap_open_x = fwd.gap_open_y = log((tr[1] + tr[2]) / 2.0);
    fwd.gap_extend_x = fwd.gap_extend_y = log((tr[4] + tr[8]) / 2.0);
    fwd.gap_switch_to_x = fwd.gap_switch_to_y = log((tr[7] + tr[5]) / 2.0); /* log 0 = -inf */
    for (int i = 0; i < 16; i++) fwd.e_match[i] = log(em[i]);
    for (int i = 0; i < 4; i++) { fwd.e_gap_x[i] = log(1.0); fwd.e_gap_y[i] = log(0.25); }
    mrp_pair_hmm rev = fwd;
    mrp_pair_hmm_reverse_complement(&rev);

    /* reads: a span of sites, a haplotype, a strand */
    int span0[N_READS], span1[N_READS], hap[N_READS], strand[N_READS];
    for (int r = 0; r < N_READS; r++) {
        span0[r] = (int) (rnd() % (N_SITES - 5));
        span1[r] = span0[r] + 4 + (int) (rnd() % 40);
        if (span1[r] >= N_SITES) span1[r] = N_SITES - 1;
        hap[r] = (int) (rnd() & 1);
        strand[r] = (int) (rnd() & 1);
    }
    /* bubbles: two alleles per site (reference window, one substituted base), one noisy substring per spanning read */
    size_t cap = (size_t) N_SITES * (2 * LA + (size_t) N_READS * (LA + 8));
    uint8_t *pool = malloc(cap);
    int64_t pool_n = 0, n_subs = 0;
    int64_t allele_first[N_SITES + 1], read_first[N_SITES + 1], allele_off[2 * N_SITES];
    int32_t allele_len[2 * N_SITES];
    int64_t *read_off = malloc(sizeof(int64_t) * N_SITES * N_READS);
    int32_t *read_len = malloc(sizeof(int32_t) * N_SITES * N_READS), *sub_read = malloc(sizeof(int32_t) * N_SITES * N_READS);
    uint8_t *sub_strand = malloc((size_t) N_SITES * N_READS);
    int truth[N_SITES];
    for (int s = 0; s < N_SITES; s++) {
        uint8_t ref[LA], alt[LA];
        for (int i = 0; i < LA; i++) ref[i] = alt[i] = (uint8_t) (rnd() & 3);
        alt[FLANK] = (uint8_t) ((alt[FLANK] + 1 + rnd() % 3) & 3);
        truth[s] = (int) (rnd() & 1); /* allele of haplotype 0 */
        allele_first[s] = 2 * s;
        read_first[s] = n_subs;
        for (int a = 0; a < 2; a++) {
            allele_off[2 * s + a] = pool_n;
            allele_len[2 * s + a] = LA;
            memcpy(pool + pool_n, a ? alt : ref, LA);
            pool_n += LA;
        }
        for (int r = 0; r < N_READS; r++) {
            if (s < span0[r] || s > span1[r]) continue;
            const uint8_t *src = (hap[r] == 0 ? truth[s] : 1 - truth[s]) ? alt : ref;
            read_off[n_subs] = pool_n;
            int n = 0;
            for (int i = 0; i < LA; i++) { /* 3 % substitutions, 1 % deletions, 1 % insertions */
                const double u = uni();
                if (u < 0.01) continue;
                pool[pool_n + n++] = u < 0.04 ? (uint8_t) (rnd() & 3) : src[i];
                if (uni() < 0.01) pool[pool_n + n++] = (uint8_t) (rnd() & 3);
            }
            read_len[n_subs] = n;
            pool_n += n;
            sub_read[n_subs] = r;
            sub_strand[n_subs] = (uint8_t) strand[r];
            n_subs++;
        }
    }
    allele_first[N_SITES] = 2 * N_SITES;
    read_first[N_SITES] = n_subs;

    /* 1. alignment likelihoods on the device */
    float *supports = malloc(sizeof(float) * 2 * (size_t) n_subs);
    mrp_pairhmm_stats pst;
    CHECK(mrp_allele_read_supports(ctx, &fwd, &rev, N_SITES, allele_first, read_first, pool, pool_n, allele_off, allele_len, read_off, read_len,
                                   sub_strand, 4, 512, supports, &pst));

    /* 2. bubble graph -> profile sequences and site tables (host) */
    uint32_t an[N_SITES];
    int64_t support_off[N_SITES + 1];
    for (int s = 0; s <= N_SITES; s++) { if (s < N_SITES) an[s] = 2; support_off[s] = 2 * read_first[s]; }
    mrp_bubbles bg = {N_SITES, an, read_first, sub_read, support_off, supports};
    mrp_read *seqs = NULL;
    int32_t *read_of_seq = NULL, fs[N_READS];
    int64_t n_seqs = 0, prof_bytes = 0;
    uint8_t *prof = NULL;
    static char name_buf[N_READS][16];
    const char *names[N_READS]; /* read ids: they order hmms that share start and length (hmm.c:82-87) */
    for (int r = 0; r < N_READS; r++) { fs[r] = strand[r]; snprintf(name_buf[r], sizeof(name_buf[r]), "read%04d", r); names[r] = name_buf[r]; }
    CHECK(mrp_profile_seqs_from_bubbles(&bg, N_READS, names, fs, &seqs, &read_of_seq, &n_seqs, &prof, &prof_bytes));
    uint32_t *a_num = NULL;
    uint16_t *sub = NULL, *prior = NULL;
    CHECK(mrp_reference_from_bubbles(&bg, 0.0, &a_num, &sub, &prior));

    /* 3. phasing, merge levels resident on the device */
    mrp_chunk *chunk = NULL;
    CHECK(mrp_chunk_create(ctx, N_SITES, a_num, sub, prior, prof, prof_bytes, &chunk));
    mrp_params params = {1, 1, 1, 0, 100, 100, 0.0, 64, 2, 10}; /* params/base_params.json "phase" */
    const mrp_chunk *chunks[1] = {chunk};
    const mrp_read *reads[1] = {seqs};
    mrp_phase_result *res[1] = {NULL};
    mrp_phase_many_stats mst;
    CHECK(mrp_phase_reads_many(ctx, 1, chunks, reads, &n_seqs, &params, res, &mst));

    /* 4. HP tags */
    int8_t *tag = malloc((size_t) n_seqs);
    CHECK(mrp_assign_reads_to_haplotypes(N_SITES, a_num, prof, seqs, n_seqs, res[0], 0, tag, NULL));
    int64_t tagged = 0, agree = 0;
    for (int64_t q = 0; q < n_seqs; q++)
        if (tag[q] == 1 || tag[q] == 2) { tagged++; agree += (tag[q] - 1) == hap[read_of_seq[q]]; }
    if (2 * agree < tagged) agree = tagged - agree;
    printf("%s: %lld pairs aligned (%lld dp cells, %.2f ms), %lld profile sequences, resident=%d, %lld sweeps; "
           "%lld of %lld reads tagged, %lld agree with their haplotype\n",
           mrp_version(), (long long) (pst.pairs_lane + pst.pairs_wave), (long long) pst.cells, pst.kernel_ms, (long long) n_seqs, mst.resident,
           (long long) res[0]->n_sweeps, (long long) tagged, (long long) n_seqs, (long long) agree);
    const int ok = tagged * 10 >= n_seqs * 9 && agree * 10 >= tagged * 9;

    mrp_phase_result_destroy(res[0]);
    mrp_chunk_destroy(chunk);
    mrp_free(seqs); mrp_free(read_of_seq); mrp_free(prof); mrp_free(a_num); mrp_free(sub); mrp_free(prior);
    free(tag); free(supports); free(pool); free(read_off); free(read_len); free(sub_read); free(sub_strand);
    mrp_context_destroy(ctx);
    return ok ? 0 : 2;
}
