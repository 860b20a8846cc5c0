"""An independent, deliberately naive evaluation of the forward/backward recursion on a flattened
job (dicts keyed by masked partitions, per-read byte sums, Python floats).  Shares no code with
oracle/rphmm_oracle.c or the HIP kernels; used to cross-check both."""
import math

import numpy as np

NEG = float("-inf")


def log_add(a, b, max_mode):
    if max_mode:
        return a if a > b else b
    if a == NEG:
        return b
    if b == NEG:
        return a
    return a + math.log1p(math.exp(b - a)) if a > b else b + math.log1p(math.exp(a - b))


def emission(chunk, flat, k, partition, ancestor):
    start, n = int(flat["col_ref_start"][k]), int(flat["col_length"][k])
    d = int(flat["col_depth"][k])
    offs = flat["read_byte_off"][flat["col_read_off"][k]:flat["col_read_off"][k + 1]]
    first = int(chunk.allele_offset[start])
    cost, so = 0, int((chunk.allele_number[:start].astype(np.int64) ** 2).sum())
    for s in range(start, start + n):
        A = int(chunk.allele_number[s])
        rel = int(chunk.allele_offset[s]) - first
        h1, h2 = [0] * A, [0] * A
        for i in range(d):
            for a in range(A):
                v = int(chunk.pool[int(offs[i]) + rel + a])
                if (partition >> i) & 1:
                    h1[a] += v
                else:
                    h2[a] += v
        if not ancestor:
            cost += min(h1) + min(h2)
        else:
            sub = chunk.sub[so:so + A * A].reshape(A, A).astype(np.int64)
            pr = chunk.prior[int(chunk.allele_offset[s]):int(chunk.allele_offset[s]) + A].astype(np.int64)
            cost += min(min(h1[q] + int(sub[i][q]) for q in range(A)) + min(h2[q] + int(sub[i][q]) for q in range(A)) + int(pr[i])
                        for i in range(A))
        so += A * A
    return -float(cost)


def forward_backward(chunk, flat, flags):
    max_mode, ancestor = bool(flags & 1), bool(flags & 2)
    K = int(flat["n_columns"])
    co, mo = flat["col_cell_off"], flat["mcol_cell_off"]
    P = [int(x) for x in flat["partition"]]
    nC = len(P)
    e = [0.0] * nC
    f, b = [NEG] * nC, [NEG] * nC
    mf = [dict() for _ in range(max(K - 1, 0))]
    mb = [dict() for _ in range(max(K - 1, 0))]
    for k in range(K - 1):
        for m in range(int(mo[k]), int(mo[k + 1])):
            mf[k][int(flat["merge_from"][m])] = NEG      # keyed by fromPartition for the forward pass
    to_of_from = [dict() for _ in range(max(K - 1, 0))]
    from_of_to = [dict() for _ in range(max(K - 1, 0))]
    for k in range(K - 1):
        for m in range(int(mo[k]), int(mo[k + 1])):
            to_of_from[k][int(flat["merge_from"][m])] = int(flat["merge_to"][m])
            from_of_to[k][int(flat["merge_to"][m])] = int(flat["merge_from"][m])
    hmm_f = hmm_b = NEG
    for k in range(K):
        for c in range(int(co[k]), int(co[k + 1])):
            e[c] = emission(chunk, flat, k, P[c], ancestor)
            fv = 0.0
            if k > 0:
                key_to = P[c] & int(flat["mask_to"][k - 1])
                fv = mf[k - 1][from_of_to[k - 1][key_to]]
            fv += e[c]
            f[c] = fv
            if k + 1 < K:
                key = P[c] & int(flat["mask_from"][k])
                mf[k][key] = log_add(mf[k][key], fv, max_mode)
            else:
                hmm_f = log_add(hmm_f, fv, max_mode)
    total = [NEG] * K
    for k in range(K - 1):
        mb[k] = {key: NEG for key in mf[k]}
    for k in range(K - 1, -1, -1):
        for c in range(int(co[k]), int(co[k + 1])):
            p = e[c]
            if k + 1 < K:
                bv = mb[k][P[c] & int(flat["mask_from"][k])]
                p += bv
            else:
                bv = 0.0
            b[c] = bv
            if k > 0:
                key = from_of_to[k - 1][P[c] & int(flat["mask_to"][k - 1])]
                mb[k - 1][key] = log_add(mb[k - 1][key], p, max_mode)
            else:
                hmm_b = log_add(hmm_b, p, max_mode)
            total[k] = log_add(total[k], f[c] + bv, max_mode)
    merge_f = np.array([mf[k][int(flat["merge_from"][m])] for k in range(K - 1) for m in range(int(mo[k]), int(mo[k + 1]))])
    merge_b = np.array([mb[k][int(flat["merge_from"][m])] for k in range(K - 1) for m in range(int(mo[k]), int(mo[k + 1]))])
    return dict(cell_forward=np.array(f), cell_backward=np.array(b), merge_forward=merge_f, merge_backward=merge_b,
                col_total=np.array(total), hmm_forward=np.array([hmm_f]), hmm_backward=np.array([hmm_b]))
