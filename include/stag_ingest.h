/*
 * stag_ingest.h -- C ABI of libstag_ingest.so: host-side text parsers and writers on either
 * side of the GPU path.  Plain C, no GPU, no allocation: the caller owns
 * every buffer.  These replace, chunk by chunk, the per-line Python loops of zellerlab/stag 0.8.2
 * (paths relative to the reference root):
 *
 *   stag_fasta_scan      <- stag/helpers.py:107-121  read_fasta over the `esl-reformat a2m -` pipe
 *                           (stag/align.py:139-147): records of wrapped lines -> one string each
 *   stag_onehot_parse    <- stag/classify.py:16-20   alignment_reader: "id\t0\t1\t..." lines of a -s file
 *   stag_onehot_format   <- stag/classify.py:160-162 / stag/align.py:207  the -S writer
 *   stag_rows_format_short <- stag/classify.py:182-190  the two-column output lines
 *
 * Streaming protocol (both parsers): the caller hands over whatever bytes it has; the parser
 * consumes whole records only and reports how many bytes it used.  The caller keeps the rest,
 * appends the next read and calls again; `final` != 0 says no more bytes will come.
 *
 * Exactness: anything the fast path cannot reproduce byte for byte (non-ASCII bytes in a FASTA
 * line, a '\r' inside one -- a line break for a text-mode reader --, fields other than "0"/"1")
 * is not guessed at: the record is flagged
 * STAG_INGEST_FALLBACK and the caller re-parses its raw bytes [raw_begin, raw_end) the slow way.
 */
#ifndef STAG_INGEST_H
#define STAG_INGEST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define STAG_INGEST_OK 0
#define STAG_INGEST_FALLBACK 1 /* re-parse this record from its raw bytes */

typedef struct {
    int64_t id_begin, id_len;   /* the id inside the INPUT buffer; id_len < 0: no header line (sid is None) */
    int64_t raw_begin, raw_end; /* the record's bytes in the input buffer */
    int64_t n_fields;           /* onehot: fields after the id; fasta: lines of the record */
    int32_t flag;               /* STAG_INGEST_OK / STAG_INGEST_FALLBACK */
    int32_t pad;
} stag_record_t;

/*
 * helpers.py:107-121.  Lines end at '\n' (binary stream iteration); each line is right-stripped of
 * ASCII whitespace; a line starting with '>' opens a record whose id is the rest of that line
 * (head_start = 1); the other lines are concatenated into text_out.  Record i's string is
 * text_out[offs[i] .. offs[i+1]); offs[0] is written as text_used_in (so batches can be appended).
 * A header with no line after it yields nothing, as in the reference (`if seq:`).
 * `at_start` != 0: the buffer begins the stream (lines before the first header form a record
 * without id); otherwise the buffer must begin with a header line.
 * Returns the number of records written (<= max_records), or -1 on bad arguments.
 * *consumed: bytes of buf used; *text_used: bytes of text_out used (including text_used_in).
 */
int64_t stag_fasta_scan(const char *buf, int64_t len, int32_t final, int32_t at_start, int64_t max_records,
                        char *text_out, int64_t text_cap, int64_t text_used_in, int64_t *offs,
                        stag_record_t *recs, int64_t *consumed, int64_t *text_used);

/*
 * The same scan for the trimmed form that stag_hip.h's stag_classify_a2m_trimmed takes: of each record's string
 * (align.py:37-58's input) the run of '-' at its start and the run at its end are not copied to text_out;
 * lead_trail[2 i] and lead_trail[2 i + 1] say how long they were (room for 2 * max_records values).  A '-' is a
 * gap column whatever surrounds it, so nothing else about the record changes.  A record flagged
 * STAG_INGEST_FALLBACK is re-parsed by the caller, who reports it untrimmed.  lead_trail == NULL: stag_fasta_scan.
 * (On x86-64 with AVX2 -- tested at run time -- plain lines of at most 64 bytes are copied with vector loads and stores;
 * the results are those of the byte-wise code, which takes every other line and is all there is elsewhere.  The
 * text buffer must have room for every byte not yet consumed, as the arguments already require: text_cap -
 * text_used_in >= len.)
 */
int64_t stag_fasta_scan_trim(const char *buf, int64_t len, int32_t final, int32_t at_start, int64_t max_records,
                             char *text_out, int64_t text_cap, int64_t text_used_in, int64_t *offs,
                             stag_record_t *recs, int64_t *consumed, int64_t *text_used, int32_t *lead_trail);

/*
 * classify.py:16-20.  Lines end at '\n', '\r' or "\r\n" (text-mode universal newlines); each is
 * right-stripped, split at '\t'; the first field is the id, the others must be "0" or "1" for the
 * fast path.  Row i gets words_per_row uint32 words in bits_out (bit f -> word f>>5, bit f&31) when
 * it has exactly n_features fields; otherwise its words are zero and n_fields tells the caller.
 * Returns rows written (<= max_rows) or -1 on bad arguments.
 */
int64_t stag_onehot_parse(const char *buf, int64_t len, int32_t final, int32_t n_features, int32_t words_per_row,
                          int64_t max_rows, uint32_t *bits_out, stag_record_t *recs, int64_t *consumed);

/*
 * classify.py:16-19 without the fields: the lines of a -s file for the GPU parser (stag_parse_onehot_dev /
 * stag_onehot_text_parse in stag_hip.h).  Same line ends, right-strip and id as stag_onehot_parse; instead of
 * parsing the fields it reports where they are: body i = buf[body_begin[i] .. body_end[i]) is what follows the
 * id's tab, right-stripped.  A line the fast path must not take (STAG_INGEST_FALLBACK: it ends in a non-ASCII
 * byte) and a line without fields get an empty body, which the GPU parser flags for the caller's slow path.
 * Returns lines written (<= max_rows) or -1 on bad arguments; *consumed: bytes of buf used (whole lines).
 */
int64_t stag_onehot_lines(const char *buf, int64_t len, int32_t final, int64_t max_rows, stag_record_t *recs,
                          int64_t *body_begin, int64_t *body_end, int64_t *consumed);

/*
 * classify.py:160-162: print(id, "\t".join("1"/"0" per feature), sep="\t").  ids are the n strings
 * ids[id_off[i] .. id_off[i+1]).  Returns the bytes written to out, or -(bytes needed) when
 * out_cap is too small.
 */
int64_t stag_onehot_format(const uint32_t *bits, int64_t n, int32_t n_features, int32_t words_per_row,
                           const char *ids, const int64_t *id_off, char *out, int64_t out_cap);

/*
 * classify.py:182-190 for the short output: print(id, assigned lineage, sep="\t") per classified sequence.  Row i
 * is written when key[i] >= 0 (a negative key: the sequence did not pass the filter, no line): its id is
 * id_buf[id_begin[i] .. id_begin[i] + id_len[i]) -- the bytes stag_fasta_scan found, never turned into Python
 * strings -- and its lineage is the string str_blob[str_off[key[i]] .. str_off[key[i] + 1]) (the caller builds one
 * string per distinct (end position, selected level) of the batch).  Returns the bytes written, or
 * -(bytes needed) - 1 when out_cap is too small.
 */
int64_t stag_rows_format_short(const char *id_buf, const int64_t *id_begin, const int64_t *id_len, const int32_t *key,
                               int64_t n, const char *str_blob, const int64_t *str_off, char *out, int64_t out_cap);

#ifdef __cplusplus
}
#endif
#endif /* STAG_INGEST_H */
