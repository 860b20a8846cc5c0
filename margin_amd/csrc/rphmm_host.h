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
int mrp_set_error(int code, const char *fmt, ...);

#ifdef __cplusplus
}
#endif
#endif
