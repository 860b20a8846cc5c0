/* TEST INFRASTRUCTURE: entry points the adaptor test calls through ctypes */
#include <math.h>
#include "margin_as_oracle.h"

void stRPHmm_forwardBackward(stRPHmm *hmm);
void stRPHmm_forwardBackwardMany(stRPHmm **hmms, int64_t n);
void mrpAdaptor_registerProfileSeqs(stReference *ref, stProfileSeq **seqs, int64_t n);
void mrpAdaptor_unregister(stReference *ref);
void mrpAdaptor_threadCleanup(void);
void mrpAdaptor_setMinCells(int64_t cells);

/* every field stRPHmm_forwardBackward has to set, overwritten with NaN */
void adp_test_poison(orc_hmm *hmm) {
    hmm->forwardLogProb = hmm->backwardLogProb = NAN;
    for (orc_column *c = hmm->firstColumn;; c = orc_mcol_next(c->nColumn)) {
        c->totalLogProb = NAN;
        for (orc_cell *cell = c->head; cell != NULL; cell = cell->nCell) cell->forwardLogProb = cell->backwardLogProb = NAN;
        if (c->nColumn == NULL) break;
        for (int64_t i = 0; i < orc_mcol_size(c->nColumn); i++) {
            orc_merge_cell *m = orc_mcol_cell(c->nColumn, i);
            m->forwardLogProb = m->backwardLogProb = NAN;
        }
    }
}
void adp_test_forward_backward(orc_hmm *hmm) { stRPHmm_forwardBackward(hmm); }
void adp_test_forward_backward_many(orc_hmm **hmms, int64_t n) { stRPHmm_forwardBackwardMany(hmms, n); }
void adp_test_register(orc_reference *ref, orc_profile_seq **seqs, int64_t n) { mrpAdaptor_registerProfileSeqs(ref, seqs, n); }
void adp_test_unregister(orc_reference *ref) { mrpAdaptor_unregister(ref); }
void adp_test_cleanup(void) { mrpAdaptor_threadCleanup(); }
void adp_test_set_min_cells(int64_t cells) { mrpAdaptor_setMinCells(cells); }
