/* TEST INFRASTRUCTURE.  Binds integration/stRPHmm_forwardBackward_adaptor.c -- written against margin's inc/margin.h and
 * sonLib -- to the CPU oracle's linked-list hmm (oracle/rphmm_oracle.h mirrors the reference's struct fields name for
 * name, citing inc/margin.h), so that the adaptor's flatten / scatter logic can be compiled and run here, where margin
 * itself cannot be built (sonLib / htslib submodules are empty). */
#ifndef MARGIN_AS_ORACLE_H_
#define MARGIN_AS_ORACLE_H_
#include <stdio.h>
#include "../../oracle/rphmm_oracle.h"

typedef orc_hmm stRPHmm;
typedef orc_column stRPColumn;
typedef orc_cell stRPCell;
typedef orc_merge_column stRPMergeColumn;
typedef orc_merge_cell stRPMergeCell;
typedef orc_profile_seq stProfileSeq;
typedef orc_reference stReference;
typedef orc_site stSite;
typedef orc_params stRPHmmParameters;

/* a merge column of the oracle is opaque; its merge cells are listed through accessors (insertion order) */
typedef orc_merge_column *adp_mcells;
#define ADP_MCOL_MASK_FROM(m) orc_mcol_maskFrom(m)
#define ADP_MCOL_MASK_TO(m) orc_mcol_maskTo(m)
#define ADP_MCOL_NEXT(m) orc_mcol_next(m)
#define ADP_MCELLS_GET(m) (m)
#define ADP_MCELLS_LEN(l) orc_mcol_size(l)
#define ADP_MCELLS_AT(l, i) orc_mcol_cell((l), (i))
#define ADP_MCELLS_FREE(l) ((void) (l))
#define ADP_ABORT(...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); abort(); } while (0)
#define ADP_MALLOC(n) malloc(n)
#define ADP_DEVICE_FOR_THIS_THREAD() 0
#define ADP_CPU_FORWARD_BACKWARD(hmm) orc_hmm_forwardBackward(hmm) /* the oracle's own body stands in for impl/hmm.c:931-942 */
#endif
