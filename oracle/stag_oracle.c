/*
 * CPU oracle (plain C) for STAG's post-alignment classification path.
 *
 * TEST INFRASTRUCTURE ONLY: built into oracle/liboracle.so by oracle/Makefile and loaded
 * only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, as the
 * checker.  The product (stag_amd/) never links or loads it.
 *
 * Every function restates one reference function of zellerlab/stag 0.8.2 (paths are
 * relative to /root/reference) on the *dense* data the reference itself holds: bool[L]
 * one-hot vectors and float64[1+L] coefficient rows.  It shares no code with the HIP
 * library (which works on symbol codes and sparse rows), so agreement between the two is
 * evidence, not tautology.
 *
 * Pinning: tests/test_oracle_golden.py checks this file against tests/golden/ (vectors
 * produced by the reference's own stag.align / stag.classify, see oracle/make_golden.py).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* status codes shared with include/stag_hip.h (STAG_SEQ_*) */
enum {
    SEQ_OK = 0,
    SEQ_MISSING_CLASSIFIER = 1, /* classifiers[sibling] -> KeyError      (classify.py:40) */
    SEQ_STRING_CLASSIFIER = 2,  /* "no_negative_examples" row evaluated  (classify.py:24-25) */
    SEQ_RANKCUT_SHAPE = 3,      /* coeff * perc does not broadcast       (classify.py:25) */
    SEQ_EMPTY_PATH = 4,         /* perc[0] on an empty list -> IndexError (classify.py:106) */
    SEQ_ZERO_COLUMNS = 5        /* n_char == 0 -> ZeroDivisionError      (align.py:58) */
};

/* stag/align.py:28-58.  bits must hold 5*len bytes.  Returns n_char. */
int64_t oracle_encode_a2m(const char *s, int64_t len, uint8_t *bits, int64_t *n_aligned_out)
{
    int64_t n_char = 0, n_aligned = 0;
    for (int64_t i = 0; i < len; i++) {
        unsigned char c = (unsigned char)s[i];
        if (c >= 'a' && c <= 'z')
            continue; /* insert state: dropped (align.py:47) */
        uint8_t *b = bits + 5 * n_char;
        b[0] = b[1] = b[2] = b[3] = b[4] = 0;
        switch (c) {
        case 'A': b[4] = 1; n_aligned++; break;
        case 'C': b[3] = 1; n_aligned++; break;
        case 'G': b[2] = 1; n_aligned++; break;
        case 'T':
        case 'U': b[1] = 1; n_aligned++; break;
        default:  b[0] = 1; break;
        }
        n_char++;
    }
    *n_aligned_out = n_aligned;
    return n_char;
}

/* stag/align.py:58,160: n_aligned / n_char * 100 >= min_perc_state (n_char > 0). */
int oracle_passes_filter(int64_t n_aligned, int64_t n_char, double min_perc_state)
{
    return ((double)n_aligned / (double)n_char * 100.0) >= min_perc_state;
}

/* stag/align.py:63-94.  out must hold 3*plen bytes.  Returns output length, -1 for None. */
int64_t oracle_protein2gene(const char *prot, int64_t plen, const char *gene, int64_t glen,
                            int check_length, char *out)
{
    int64_t n_res = 0;
    for (int64_t i = 0; i < plen; i++)
        if (prot[i] != '-') n_res++;
    if (check_length && glen != 3 * n_res && glen - 3 != 3 * n_res)
        return -1;
    int64_t pos = 0, o = 0;
    for (int64_t i = 0; i < plen; i++) {
        unsigned char c = (unsigned char)prot[i];
        if (c == '-') {
            out[o++] = '-'; out[o++] = '-'; out[o++] = '-';
        } else if (c >= 'A' && c <= 'Z') {
            for (int64_t k = pos; k < pos + 3 && k < glen; k++) out[o++] = gene[k];
            pos += 3;
        } else if (c >= 'a' && c <= 'z') {
            pos += 3;
        }
    }
    return o;
}

static double sigmoid(double z) { return 1.0 / (1.0 + exp(-z)); }

/* stag/classify.py:22-27 for a bool vector. */
double oracle_logistic_bits(const uint8_t *x, int64_t n, const double *coeff_raw)
{
    double s = 0.0;
    for (int64_t i = 0; i < n; i++)
        if (x[i]) s += coeff_raw[1 + i];
    return sigmoid(s + coeff_raw[0]);
}

/* stag/classify.py:22-27 for a float64 vector, with NumPy's 1-d broadcasting of
 * coeff[nc] * x[nx]: equal lengths, or either length 1.  Returns NaN via *ok=0 otherwise. */
double oracle_logistic_f64(const double *x, int64_t nx, const double *coeff_raw, int64_t nc, int *ok)
{
    double s = 0.0;
    *ok = 1;
    if (nx == nc) {
        for (int64_t i = 0; i < nc; i++) s += coeff_raw[1 + i] * x[i];
    } else if (nx == 1) {
        for (int64_t i = 0; i < nc; i++) s += coeff_raw[1 + i] * x[0];
    } else if (nc == 1) {
        for (int64_t i = 0; i < nx; i++) s += coeff_raw[1] * x[i];
    } else {
        *ok = 0;
        return 0.0;
    }
    return sigmoid(s + coeff_raw[0]);
}

/*
 * stag/classify.py:31-52 + 94-114 for n sequences.
 *   child_off[n_nodes+1], child_ids[]: children in stored order, node 0 = "tree_root".
 *   row_kind[node]: 0 = no classifier, 1 = float64 row, 2 = "no_negative_examples".
 *   rows: [n_nodes][1+L] (only rows with kind 1 are read).
 *   tf: [K][1+n_tf] rank-cut rows in sorted-key order.
 * Outputs (per sequence, max_depth wide): path_node, path_prob (after the placeholder
 * fix-up of classify.py:106-110), path_len, single_mask (bit i: level i had exactly one
 * child), level_prob[K], selected (index into the sorted keys or -1), n_aligned, status.
 */
void oracle_classify_bits(int32_t n_nodes, int32_t L, const int32_t *child_off,
                          const int32_t *child_ids, const uint8_t *row_kind, const double *rows,
                          int32_t K, int32_t n_tf, const double *tf, const uint8_t *bits,
                          int64_t n, int32_t max_depth, int32_t *path_node, double *path_prob,
                          int32_t *path_len, uint32_t *single_mask, double *level_prob,
                          int32_t *selected, int32_t *n_aligned, int32_t *status)
{
    (void)n_nodes;
    for (int64_t s = 0; s < n; s++) {
        const uint8_t *x = bits + s * (int64_t)L;
        int32_t *pn = path_node + s * max_depth;
        double *pp = path_prob + s * max_depth;
        double *lp = level_prob + s * (int64_t)K;
        int32_t st = SEQ_OK, len = 0, node = 0;
        uint32_t mask = 0;

        int32_t ncols = (L + 4) / 5, gaps = 0; /* classify.py:85-87: len(seq[0::5]) - sum */
        for (int32_t c = 0; c < ncols; c++) gaps += x[5 * c] ? 1 : 0;
        n_aligned[s] = ncols - gaps;

        for (int d = 0; d < max_depth; d++) { pn[d] = -1; pp[d] = 0.0; }
        for (int k = 0; k < K; k++) lp[k] = 0.0;
        selected[s] = -1;

        while (st == SEQ_OK) {
            int32_t b = child_off[node], e = child_off[node + 1];
            if (e == b) break;                       /* leaf: empty child list */
            if (len >= max_depth) break;             /* caller sized max_depth too small */
            int32_t best = -1; double best_p = -1.0;
            if (e - b == 1) {
                best = child_ids[b]; best_p = 2.0; mask |= 1u << len;
            } else {
                for (int32_t j = b; j < e && st == SEQ_OK; j++) {
                    int32_t c = child_ids[j];
                    if (row_kind[c] == 0) { st = SEQ_MISSING_CLASSIFIER; break; }
                    if (row_kind[c] == 2) { st = SEQ_STRING_CLASSIFIER; break; }
                    double p = oracle_logistic_bits(x, L, rows + (int64_t)c * (1 + L));
                    if (p > best_p) { best_p = p; best = c; }
                }
                if (st != SEQ_OK) break;
                /* all-NaN probabilities leave best == -1: the reference stops there
                   (best_taxa stays "", classify.py:49) */
                if (best < 0) break;
            }
            pn[len] = best; pp[len] = best_p; len++;
            node = best;
        }
        if (st == SEQ_OK && len == 0) st = SEQ_EMPTY_PATH;
        if (st == SEQ_OK) {
            if (pp[0] == 2.0) pp[0] = 1.0;
            for (int i = 1; i < len; i++)
                if (pp[i] == 2.0) pp[i] = pp[i - 1];
            double best_p = 0.0;
            for (int k = 0; k < K && st == SEQ_OK; k++) {
                int ok;
                double p = oracle_logistic_f64(pp, len, tf + (int64_t)k * (1 + n_tf), n_tf, &ok);
                if (!ok) { st = SEQ_RANKCUT_SHAPE; break; }
                lp[k] = p;
                if (p > best_p) { best_p = p; selected[s] = k; }
            }
        }
        path_len[s] = len;
        single_mask[s] = mask;
        status[s] = st;
    }
}
