/* Memory-safety driver for libstag_ingest (include/stag_ingest.h): random buffers in exactly sized heap blocks, the
 * streaming protocol as stag_amd/ingest.py drives it, every output array exactly as large as the header says it must
 * be.  Built with -fsanitize=address,undefined by tests/test_host_ingest.py; any finding aborts the process. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "stag_ingest.h"

static uint64_t rng_state;
static uint32_t rnd(void) {
    rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(rng_state >> 33);
}
static int64_t below(int64_t n) { return n > 0 ? (int64_t)(rnd() % (uint64_t)n) : 0; }

static const char ALPHA_FASTA[] = ">>\n\n\n\r\t  ACGT-acgt..\x80\xc3\xa9\x1c";
static const char ALPHA_ONEHOT[] = "0101010101\t\t\t\t\t\t\t\t\n\n\r x2\xa0";

static char *random_text(int64_t len, const char *alpha, size_t n_alpha) {
    char *p = (char *)malloc(len > 0 ? (size_t)len : 1);
    for (int64_t i = 0; i < len; ++i) p[i] = alpha[rnd() % n_alpha];
    return p;
}

/* well-formed -s lines with a few damaged ones, so that the fast paths run too */
static char *onehot_text(int64_t *len_out, int nf) {
    const int lines = 1 + (int)below(6);
    char *p = (char *)malloc((size_t)lines * (size_t)(2 * nf + 16) + 1);
    int64_t o = 0;
    for (int l = 0; l < lines; ++l) {
        o += sprintf(p + o, "id%d", l);
        int n = nf + (below(5) == 0 ? (int)below(3) - 1 : 0);
        for (int f = 0; f < n; ++f) { p[o++] = '\t'; p[o++] = (char)('0' + (rnd() & 1)); }
        if (below(6) == 0 && o > 4) p[o - 1 - below(4)] = ALPHA_ONEHOT[rnd() % (sizeof ALPHA_ONEHOT - 1)];
        if (below(4) == 0) p[o++] = ' ';
        if (below(3) == 0) p[o++] = '\r';
        if (l + 1 < lines || below(2)) p[o++] = '\n';
    }
    *len_out = o;
    char *q = (char *)malloc(o > 0 ? (size_t)o : 1);      /* exactly sized */
    memcpy(q, p, (size_t)o);
    free(p);
    return q;
}

static void run_fasta(const char *text, int64_t len) {
    int64_t start = 0, end = 0;
    int at_start = 1;
    for (int guard = 0; guard < 100000; ++guard) {
        if (end < len) end += 1 + below(len - end);
        const int final = end >= len;
        const int64_t avail = end - start, max_rec = 1 + below(4);
        char *buf = (char *)malloc(avail > 0 ? (size_t)avail : 1);
        memcpy(buf, text + start, (size_t)avail);
        const int64_t used_in = below(3), cap = used_in + avail;
        char *out = (char *)malloc(cap > 0 ? (size_t)cap : 1);
        int64_t *offs = (int64_t *)malloc(sizeof(int64_t) * (size_t)(max_rec + 1));
        stag_record_t *recs = (stag_record_t *)malloc(sizeof(stag_record_t) * (size_t)max_rec);
        int64_t consumed = -1, text_used = -1;
        /* every other call asks for the trimmed form: two counts per record, exactly sized */
        const int trim = (int)below(2);
        int32_t *lt = trim ? (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)max_rec) : NULL;
        const int64_t n = trim ? stag_fasta_scan_trim(buf, avail, final, at_start, max_rec, out, cap, used_in, offs, recs, &consumed, &text_used, lt)
                               : stag_fasta_scan(buf, avail, final, at_start, max_rec, out, cap, used_in, offs, recs, &consumed, &text_used);
        if (n >= 0 && trim)
            for (int64_t i = 0; i < n; ++i) {
                if (lt[2 * i] < 0 || lt[2 * i + 1] < 0) abort();
                if (offs[i + 1] > offs[i] && (out[offs[i]] == '-' || out[offs[i + 1] - 1] == '-') && recs[i].flag == STAG_INGEST_OK) abort();
            }
        free(lt);
        if (n >= 0) {
            if (n > max_rec || consumed < 0 || consumed > avail || text_used < used_in || text_used > cap) abort();
            for (int64_t i = 0; i < n; ++i)
                if (recs[i].raw_begin < 0 || recs[i].raw_end > avail || recs[i].raw_begin > recs[i].raw_end ||
                    offs[i] > offs[i + 1] || (recs[i].id_len >= 0 && recs[i].id_begin + recs[i].id_len > avail))
                    abort();
            start += consumed;
            if (consumed > 0) at_start = 0;
        }
        free(buf); free(out); free(offs); free(recs);
        if (n < 0) return;                      /* a chunk that does not begin at a header: the caller's error */
        if (final && (consumed == 0 || start >= len)) return;
    }
}

static void run_onehot(const char *text, int64_t len, int nf, int lines_only) {
    const int W = ((nf + 31) / 32 + 3) / 4 * 4 + (int)below(2) * 4;
    int64_t start = 0, end = 0;
    for (int guard = 0; guard < 100000; ++guard) {
        if (end < len) end += 1 + below(len - end);
        const int final = end >= len;
        const int64_t avail = end - start, max_rows = 1 + below(4);
        char *buf = (char *)malloc(avail > 0 ? (size_t)avail : 1);
        memcpy(buf, text + start, (size_t)avail);
        stag_record_t *recs = (stag_record_t *)malloc(sizeof(stag_record_t) * (size_t)max_rows);
        int64_t consumed = -1, n;
        if (lines_only) {
            int64_t *bb = (int64_t *)malloc(8 * (size_t)max_rows), *be = (int64_t *)malloc(8 * (size_t)max_rows);
            n = stag_onehot_lines(buf, avail, final, max_rows, recs, bb, be, &consumed);
            for (int64_t i = 0; i < n; ++i)
                if (bb[i] < 0 || bb[i] > be[i] || be[i] > avail) abort();
            free(bb); free(be);
        } else {
            uint32_t *bits = (uint32_t *)malloc(4 * (size_t)max_rows * (size_t)W);
            n = stag_onehot_parse(buf, avail, final, nf, W, max_rows, bits, recs, &consumed);
            free(bits);
        }
        if (n < 0 || n > max_rows || consumed < 0 || consumed > avail) abort();
        for (int64_t i = 0; i < n; ++i)
            if (recs[i].raw_begin < 0 || recs[i].raw_end > avail || recs[i].id_begin + recs[i].id_len > avail) abort();
        start += consumed;
        free(buf); free(recs);
        if (final && (consumed == 0 || start >= len)) return;
    }
}

static void run_format(void) {
    const int nf = (int)below(70), W = ((nf + 31) / 32 + 3) / 4 * 4, n = (int)below(4);
    uint32_t *bits = (uint32_t *)malloc(4 * (size_t)(n * W) + 4);
    for (int i = 0; i < n * W; ++i) bits[i] = rnd();
    int64_t *id_off = (int64_t *)malloc(8 * (size_t)(n + 1));
    id_off[0] = 0;
    for (int i = 0; i < n; ++i) id_off[i + 1] = id_off[i] + below(5);
    char *ids = (char *)malloc((size_t)id_off[n] + 1);
    memset(ids, 'x', (size_t)id_off[n]);
    int64_t need = -stag_onehot_format(bits, n, nf, W, ids, id_off, NULL, 0);
    if (n == 0 ? need != 0 : need <= 0) abort();
    char *out = (char *)malloc(need > 0 ? (size_t)need : 1);
    if (stag_onehot_format(bits, n, nf, W, ids, id_off, out, need) != need) abort();
    if (need > 1 && stag_onehot_format(bits, n, nf, W, ids, id_off, out, need - 1) != -need) abort();
    free(bits); free(id_off); free(ids); free(out);
}

static void run_rows(void) {
    const int64_t n = below(6), n_str = 1 + below(3);
    char ids[64];
    int64_t id_begin[8], id_len[8], str_off[8];
    int32_t key[8];
    char blob[64];
    for (int i = 0; i < 64; ++i) { ids[i] = (char)('a' + i % 26); blob[i] = (char)('A' + i % 26); }
    str_off[0] = 0;
    for (int64_t j = 0; j < n_str; ++j) str_off[j + 1] = str_off[j] + below(9);
    int64_t want = 0;
    for (int64_t i = 0; i < n; ++i) {
        id_len[i] = below(8);
        id_begin[i] = below(64 - id_len[i]);
        key[i] = (int32_t)below(n_str + 1) - 1;
        if (key[i] >= 0) want += id_len[i] + 2 + (str_off[key[i] + 1] - str_off[key[i]]);
    }
    char *out = (char *)malloc(want > 0 ? (size_t)want : 1);    /* exactly sized */
    if (stag_rows_format_short(ids, id_begin, id_len, key, n, blob, str_off, out, want) != want) abort();
    if (want > 0 && stag_rows_format_short(ids, id_begin, id_len, key, n, blob, str_off, out, want - 1) != -want - 1) abort();
    free(out);
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    for (int it = 0; it < iters; ++it) {
        rng_state = 0x9e3779b97f4a7c15ull * (uint64_t)(it + 1);
        int64_t len = below(below(3) ? 200 : 3000);
        char *t = random_text(len, ALPHA_FASTA, sizeof ALPHA_FASTA - 1);
        run_fasta(t, len);
        free(t);
        len = below(400);
        t = random_text(len, ALPHA_ONEHOT, sizeof ALPHA_ONEHOT - 1);
        run_onehot(t, len, (int)below(40), it & 1);
        free(t);
        const int nf = 1 + (int)below(70);
        t = onehot_text(&len, nf);
        run_onehot(t, len, nf, 0);
        run_onehot(t, len, nf, 1);
        free(t);
        run_format();
        run_rows();
    }
    printf("ok %d\n", iters);
    return 0;
}
