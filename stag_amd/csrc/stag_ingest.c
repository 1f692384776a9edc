/* libstag_ingest.so: include/stag_ingest.h.  Host-only C; no allocation, no global state. */
#include "stag_ingest.h"

#include <string.h>

/* what str.rstrip() removes among ASCII bytes: \t \n \v \f \r, FS GS RS US, space */
static inline int ascii_space(unsigned char c) { return (c >= 0x09 && c <= 0x0d) || (c >= 0x1c && c <= 0x20); }

/* end of the line's content after rstrip; *fallback set when the content ends in a non-ASCII byte
 * (U+0085, U+00A0 ... are whitespace for Python too: let the caller's slow path decide) */
static inline int64_t rstrip_end(const char *buf, int64_t b, int64_t e, int *fallback) {
    while (e > b && ascii_space((unsigned char)buf[e - 1])) --e;
    if (e > b && ((unsigned char)buf[e - 1] & 0x80)) *fallback = 1;
    return e;
}

/* bytes of buf[b, e) before the first one that is not '-' */
static inline int64_t gap_run(const char *buf, int64_t b, int64_t e) {
    int64_t k = b;
    while (k + 8 <= e) {
        uint64_t w;
        memcpy(&w, buf + k, 8);
        if (w != 0x2d2d2d2d2d2d2d2dull) break;
        k += 8;
    }
    while (k < e && buf[k] == '-') ++k;
    return k - b;
}

/* bytes at the end of text[b, e) that are '-' */
static inline int64_t gap_run_back(const char *text, int64_t b, int64_t e) {
    int64_t k = e;
    while (k - 8 >= b) {
        uint64_t w;
        memcpy(&w, text + k - 8, 8);
        if (w != 0x2d2d2d2d2d2d2d2dull) break;
        k -= 8;
    }
    while (k > b && text[k - 1] == '-') --k;
    return e - k;
}

#if defined(__x86_64__)
#include <immintrin.h>
/* The lines of an open record that need nothing but copying, as many as follow each other from *pos on: at most 64
 * bytes before the '\n', at least one, none of them '\r' or outside ASCII, the last one not white space, no '>' in
 * front, and all 64 bytes inside the buffer.  That is every full line `esl-reformat a2m` writes (60 columns); what it
 * is not -- headers, the last line of a record if it is short of the buffer's end, anything odd -- stays with the
 * byte-wise code of the caller, which this function returns to.  A line is two 32-byte loads, three compares and
 * two stores instead of memchr + rstrip + memcpy + a scan for the high bit.  With `trim`, lines of nothing but '-'
 * before the record's first other character are counted (*lead), not copied; the line on which that run ends is the
 * caller's. */
__attribute__((target("avx2,bmi"))) static void copy_plain_lines(const char *buf, int64_t len, int64_t *pos_io,
                                                                  char *text_out, int64_t *used_io, int64_t r_text,
                                                                  int64_t *lines_io, int64_t *lead_io, int trim) {
    int64_t pos = *pos_io, used = *used_io, lines = *lines_io, lead = *lead_io;
    const __m256i nl = _mm256_set1_epi8('\n'), cr = _mm256_set1_epi8('\r'), gap = _mm256_set1_epi8('-');
    int Lp = 60;                                                /* the width `esl-reformat a2m` writes */
    while (pos + 64 <= len && buf[pos] != '>') {
        const __m256i a = _mm256_loadu_si256((const __m256i *)(buf + pos));
        const __m256i b = _mm256_loadu_si256((const __m256i *)(buf + pos + 32));
        const uint64_t m_nl = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(a, nl)) |
                              ((uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(b, nl)) << 32);
        if (m_nl == 0) break;                                   /* a longer line */
        /* bytes before the '\n'.  Lines of a file mostly have one width: when this one is as wide as the last, the
         * next line's address comes from that width (a value the processor already has) and not from this line's
         * bytes, so that the loads of line after line are in flight together */
        int L;
        if (((m_nl >> Lp) & 1ull) && (m_nl & ((1ull << Lp) - 1ull)) == 0) L = Lp;
        else Lp = L = (int)_tzcnt_u64(m_nl);
        if (L == 0) break;                                      /* a blank line */
        const uint64_t content = (1ull << L) - 1ull;            /* (L <= 63: the '\n' is among the 64 bytes) */
        const uint64_t m_bad = ((uint32_t)_mm256_movemask_epi8(_mm256_or_si256(a, _mm256_cmpeq_epi8(a, cr))) |
                                ((uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_or_si256(b, _mm256_cmpeq_epi8(b, cr))) << 32));
        if (m_bad & content) break;                             /* '\r' or a byte outside ASCII */
        if (ascii_space((unsigned char)buf[pos + L - 1])) break; /* something to strip */
        if (trim && used == r_text) {                           /* nothing but '-' so far */
            const uint64_t m_gap = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(a, gap)) |
                                   ((uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(b, gap)) << 32);
            if ((m_gap & content) != content) break;            /* the run ends on this line */
            lead += L;
        } else {
            /* 64 bytes are written, L of them count: the caller's text buffer has room for every byte not yet consumed */
            _mm256_storeu_si256((__m256i *)(text_out + used), a);
            _mm256_storeu_si256((__m256i *)(text_out + used + 32), b);
            used += L;
        }
        ++lines;
        pos += L + 1;
    }
    *pos_io = pos; *used_io = used; *lines_io = lines; *lead_io = lead;
}
#define STAG_HAVE_PLAIN_LINES 1
#endif

int64_t stag_fasta_scan(const char *buf, int64_t len, int32_t final, int32_t at_start, int64_t max_records,
                        char *text_out, int64_t text_cap, int64_t text_used_in, int64_t *offs,
                        stag_record_t *recs, int64_t *consumed, int64_t *text_used) {
    return stag_fasta_scan_trim(buf, len, final, at_start, max_records, text_out, text_cap, text_used_in, offs, recs,
                                consumed, text_used, NULL);
}

int64_t stag_fasta_scan_trim(const char *buf, int64_t len, int32_t final, int32_t at_start, int64_t max_records,
                             char *text_out, int64_t text_cap, int64_t text_used_in, int64_t *offs,
                             stag_record_t *recs, int64_t *consumed, int64_t *text_used, int32_t *lead_trail) {
    if (!buf || len < 0 || max_records < 0 || !text_out || !offs || !recs || !consumed || !text_used ||
        text_used_in < 0 || text_cap - text_used_in < len)
        return -1;
    int64_t n = 0, pos = 0, used = text_used_in;
    offs[0] = used;
    *consumed = 0;
    if (max_records == 0) { *text_used = used; return 0; }   /* no room for a record: nothing is consumed */
    /* the open record */
    int open = 0, fb = 0;
    int64_t r_begin = 0, r_id = 0, r_idlen = -1, r_lines = 0, r_text = used, r_lead = 0;
    /* closes the open record's text: with lead_trail, the run of '-' at its end is dropped and both runs are
     * reported (a flagged record is re-parsed by the caller, who then reports none dropped) */
#define STAG_CLOSE_TRIM()                                                        \
    do {                                                                         \
        if (lead_trail) {                                                        \
            const int64_t t_ = gap_run_back(text_out, r_text, used);             \
            used -= t_;                                                          \
            lead_trail[2 * n] = (int32_t)(r_lead > 0x7fffffff ? 0x7fffffff : r_lead); \
            lead_trail[2 * n + 1] = (int32_t)(t_ > 0x7fffffff ? 0x7fffffff : t_);     \
        }                                                                        \
    } while (0)

#ifdef STAG_HAVE_PLAIN_LINES
    const int plain_lines = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi");
#endif
    while (pos < len) {
#ifdef STAG_HAVE_PLAIN_LINES
        if (plain_lines && open && !fb && buf[pos] != '>' && pos + 64 <= len) {
            copy_plain_lines(buf, len, &pos, text_out, &used, r_text, &r_lines, &r_lead, lead_trail != NULL);
            if (pos >= len) break;
        }
#endif
        const char *nl = (const char *)memchr(buf + pos, '\n', (size_t)(len - pos));
        if (!nl && !final) break;   /* incomplete line: wait for more bytes */
        const int64_t line_end = nl ? (int64_t)(nl - buf) : len;
        const int64_t next = nl ? line_end + 1 : len;
        if (buf[pos] == '>') {
            if (open) {   /* the previous record ends where this header begins */
                if (r_lines > 0 || fb) {   /* a flagged header may hide lines behind a '\r' */
                    STAG_CLOSE_TRIM();
                    recs[n].id_begin = r_id; recs[n].id_len = r_idlen; recs[n].raw_begin = r_begin; recs[n].raw_end = pos;
                    recs[n].n_fields = r_lines; recs[n].flag = fb ? STAG_INGEST_FALLBACK : STAG_INGEST_OK; recs[n].pad = 0;
                    offs[++n] = used;
                } else {
                    used = r_text;
                }
                open = 0;
                *consumed = pos;
                if (n >= max_records) { *text_used = used; return n; }
            }
            fb = 0;
            const int64_t e = rstrip_end(buf, pos, line_end, &fb);
            {   /* a '\r' is a line break for text-mode readers: anywhere but right at the end of the line it
                 * starts another line there (which counts even when it is blank after stripping) */
                int64_t lim = line_end;
                if (lim > pos && buf[lim - 1] == '\r') --lim;
                if (memchr(buf + pos, '\r', (size_t)(lim - pos))) fb = 1;
            }
            open = 1;
            r_begin = pos; r_id = pos + 1; r_idlen = e - (pos + 1); r_lines = 0; r_text = used; r_lead = 0;
        } else {
            if (!open) {
                if (!at_start || *consumed != 0 || n != 0) return -1;   /* every later chunk starts at a header */
                open = 1; fb = 0;
                r_begin = pos; r_id = pos; r_idlen = -1; r_lines = 0; r_text = used; r_lead = 0;
            }
            const int64_t e = rstrip_end(buf, pos, line_end, &fb);
            unsigned char hi = 0;   /* a non-ASCII character is ONE column for the reference, not one per byte */
            int64_t from = pos;
            if (lead_trail && used == r_text) {   /* nothing but '-' so far: this line's run of them is dropped too */
                from += gap_run(buf, pos, e);
                r_lead += from - pos;
            }
            memcpy(text_out + used, buf + from, (size_t)(e - from));
            for (int64_t i = from; i < e; ++i) {   /* vectorised by the compiler */
                const unsigned char c = (unsigned char)buf[i];
                hi |= (unsigned char)(c | (c == '\r' ? 0x80 : 0));
            }
            if (hi & 0x80) fb = 1;   /* also an inner '\r': a line break for text-mode readers */
            used += e - from;
            ++r_lines;
        }
        pos = next;
    }
    if (final && pos >= len && open) {
        if ((r_lines > 0 || fb) && n < max_records) {
            STAG_CLOSE_TRIM();
            recs[n].id_begin = r_id; recs[n].id_len = r_idlen; recs[n].raw_begin = r_begin; recs[n].raw_end = len;
            recs[n].n_fields = r_lines; recs[n].flag = fb ? STAG_INGEST_FALLBACK : STAG_INGEST_OK; recs[n].pad = 0;
            offs[++n] = used;
            *consumed = len;
        } else if (r_lines == 0 && !fb) {
            used = r_text;
            *consumed = len;
        } else {
            used = r_text;   /* no room for it: the caller calls again */
        }
    } else if (open) {
        used = r_text;       /* incomplete record: its lines are re-read next time */
    }
    if (final && !open && pos >= len) *consumed = len;
    *text_used = used;
    return n;
#undef STAG_CLOSE_TRIM
}

int64_t stag_onehot_parse(const char *buf, int64_t len, int32_t final, int32_t n_features, int32_t words_per_row,
                          int64_t max_rows, uint32_t *bits_out, stag_record_t *recs, int64_t *consumed) {
    if (!buf || len < 0 || max_rows < 0 || !bits_out || !recs || !consumed || n_features < 0 ||
        (int64_t)words_per_row * 32 < n_features)
        return -1;
    int64_t n = 0, pos = 0;
    *consumed = 0;
    while (pos < len && n < max_rows) {
        /* universal newlines: the line ends at the first '\n' or '\r' */
        const char *a = (const char *)memchr(buf + pos, '\n', (size_t)(len - pos));
        const int64_t lim = a ? (int64_t)(a - buf) : len;
        const char *b = (const char *)memchr(buf + pos, '\r', (size_t)(lim - pos));
        const int64_t e = b ? (int64_t)(b - buf) : lim;
        int64_t next;
        if (e >= len) {
            if (!final) break;
            next = len;
        } else if (buf[e] == '\r') {
            if (e + 1 < len) next = buf[e + 1] == '\n' ? e + 2 : e + 1;
            else if (final) next = len;
            else break;   /* a '\n' may still follow */
        } else {
            next = e + 1;
        }
        int fb = 0;
        const int64_t le = rstrip_end(buf, pos, e, &fb);
        stag_record_t *r = &recs[n];
        r->raw_begin = pos; r->raw_end = next; r->pad = 0;
        const char *tab = (const char *)memchr(buf + pos, '\t', (size_t)(le - pos));
        const int64_t id_end = tab ? (int64_t)(tab - buf) : le;
        r->id_begin = pos; r->id_len = id_end - pos;
        uint32_t *row = bits_out + (size_t)n * words_per_row;
        memset(row, 0, (size_t)words_per_row * sizeof(uint32_t));
        int64_t nf = 0;
        if (tab) {
            const int64_t body = id_end + 1, blen = le - body;   /* "x\tx\t...x": blen = 2 nf - 1 */
            if (blen <= 0 || (blen & 1) == 0) {
                fb = 1;
            } else {
                nf = (blen + 1) / 2;
                const unsigned char *p = (const unsigned char *)buf + body;
                unsigned bad = 0;
                if (nf == n_features) {
                    for (int64_t w = 0; w * 32 < nf; ++w) {
                        const int64_t f0 = w * 32, f1 = f0 + 32 < nf ? f0 + 32 : nf;
                        uint32_t v = 0;
                        for (int64_t f = f0; f < f1; ++f) {
                            const unsigned c = p[2 * f];
                            bad |= (c & 0xfeu) ^ 0x30u;            /* '0' or '1' */
                            v |= (uint32_t)(c & 1u) << (f - f0);
                        }
                        row[w] = v;
                    }
                } else {
                    for (int64_t f = 0; f < nf; ++f) bad |= (p[2 * f] & 0xfeu) ^ 0x30u;
                }
                for (int64_t f = 0; f + 1 < nf; ++f) bad |= p[2 * f + 1] ^ (unsigned)'\t';
                if (bad) { fb = 1; memset(row, 0, (size_t)words_per_row * sizeof(uint32_t)); }
            }
        }
        r->n_fields = nf;
        r->flag = fb ? STAG_INGEST_FALLBACK : STAG_INGEST_OK;
        ++n;
        pos = next;
        *consumed = pos;
    }
    return n;
}

int64_t stag_onehot_lines(const char *buf, int64_t len, int32_t final, int64_t max_rows, stag_record_t *recs,
                          int64_t *body_begin, int64_t *body_end, int64_t *consumed) {
    if (!buf || len < 0 || max_rows < 0 || !recs || !body_begin || !body_end || !consumed) return -1;
    int64_t n = 0, pos = 0;
    *consumed = 0;
    while (pos < len && n < max_rows) {
        /* the same line ends as stag_onehot_parse: the first '\n' or '\r' */
        const char *a = (const char *)memchr(buf + pos, '\n', (size_t)(len - pos));
        const int64_t lim = a ? (int64_t)(a - buf) : len;
        const char *b = (const char *)memchr(buf + pos, '\r', (size_t)(lim - pos));
        const int64_t e = b ? (int64_t)(b - buf) : lim;
        int64_t next;
        if (e >= len) {
            if (!final) break;
            next = len;
        } else if (buf[e] == '\r') {
            if (e + 1 < len) next = buf[e + 1] == '\n' ? e + 2 : e + 1;
            else if (final) next = len;
            else break;
        } else {
            next = e + 1;
        }
        int fb = 0;
        const int64_t le = rstrip_end(buf, pos, e, &fb);
        const char *tab = (const char *)memchr(buf + pos, '\t', (size_t)(le - pos));
        const int64_t id_end = tab ? (int64_t)(tab - buf) : le;
        stag_record_t *r = &recs[n];
        r->raw_begin = pos; r->raw_end = next; r->pad = 0; r->n_fields = 0;
        r->id_begin = pos; r->id_len = id_end - pos;
        r->flag = fb ? STAG_INGEST_FALLBACK : STAG_INGEST_OK;
        /* a flagged line, or one without fields, gets an empty body: the device parser flags it in turn */
        body_begin[n] = (tab && !fb) ? id_end + 1 : le;
        body_end[n] = le;
        ++n;
        pos = next;
        *consumed = pos;
    }
    return n;
}

int64_t stag_onehot_format(const uint32_t *bits, int64_t n, int32_t n_features, int32_t words_per_row,
                           const char *ids, const int64_t *id_off, char *out, int64_t out_cap) {
    if (!bits || n < 0 || n_features < 0 || !ids || !id_off || (!out && out_cap > 0)) return -1;
    int64_t need = 0;
    for (int64_t i = 0; i < n; ++i) need += (id_off[i + 1] - id_off[i]) + 1 + (n_features ? 2 * (int64_t)n_features - 1 : 0) + 1;
    if (need > out_cap) return -need;
    char lut[256][16];
    for (int v = 0; v < 256; ++v)
        for (int k = 0; k < 8; ++k) { lut[v][2 * k] = (char)('0' + ((v >> k) & 1)); lut[v][2 * k + 1] = '\t'; }
    char *o = out;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t il = id_off[i + 1] - id_off[i];
        memcpy(o, ids + id_off[i], (size_t)il);
        o += il;
        *o++ = '\t';   /* print(id, "", sep="\t") also writes the separator when there are no features */
        const uint32_t *row = bits + (size_t)i * words_per_row;
        /* eight features at a time: byte b of the row -> the 16 characters "x\tx\t...x\t" from a table (the last
         * feature's tab is taken back below); the text is 2 bytes per feature, so this loop is the cost of -S */
        int32_t f = 0;
        for (; f + 8 <= n_features; f += 8) {
            memcpy(o, lut[(row[f >> 5] >> (f & 31)) & 0xffu], 16);
            o += 16;
        }
        for (; f < n_features; ++f) {
            *o++ = (char)('0' + ((row[f >> 5] >> (f & 31)) & 1u));
            *o++ = '\t';
        }
        if (n_features > 0) --o;   /* no tab behind the last feature */
        *o++ = '\n';
    }
    return (int64_t)(o - out);
}

int64_t stag_rows_format_short(const char *id_buf, const int64_t *id_begin, const int64_t *id_len, const int32_t *key,
                               int64_t n, const char *str_blob, const int64_t *str_off, char *out, int64_t out_cap) {
    if (!id_buf || !id_begin || !id_len || !key || n < 0 || !str_blob || !str_off || (!out && out_cap > 0)) return -1;
    int64_t need = 0;
    for (int64_t i = 0; i < n; ++i)
        if (key[i] >= 0) need += id_len[i] + 1 + (str_off[key[i] + 1] - str_off[key[i]]) + 1;
    if (need > out_cap) return -need - 1;
    char *o = out;
    for (int64_t i = 0; i < n; ++i) {
        if (key[i] < 0) continue;
        memcpy(o, id_buf + id_begin[i], (size_t)id_len[i]);
        o += id_len[i];
        *o++ = '\t';
        const int64_t a = str_off[key[i]], b = str_off[key[i] + 1];
        memcpy(o, str_blob + a, (size_t)(b - a));
        o += b - a;
        *o++ = '\n';
    }
    return (int64_t)(o - out);
}
