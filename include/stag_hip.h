/*
 * stag_hip.h -- C ABI of libstag_hip.so, the MI355X (gfx950) implementation of STAG's
 * post-alignment classification path.
 *
 * The reference (zellerlab/stag 0.8.2, pure Python) has no FFI layer; these entry points are
 * what a binding for this path replaces, function by function (paths relative to the
 * reference root):
 *
 *   stag_encode_a2m*      <- stag/align.py:28-58   convert_alignment  (+ :160 min_perc_state filter)
 *   stag_backtranslate*   <- stag/align.py:63-94   protein2gene_alignment (host: one pair; _dev: a batch)
 *   stag_classify_protein_a2m* <- stag/align.py:155-164 + stag/classify.py:155-158 for protein databases
 *   stag_classify_bits*   <- stag/classify.py:22-124  run_logistic_prediction, find_best_score,
 *                            predict_iter, find_correct_level, find_n_aligned_characters, classify_seq
 *   stag_classify_a2m*    <- the loop body of stag/classify.py:155-158 fed by stag/align.py:149-164
 *   stag_classify_concat  <- stag/classify_genome.py:364-388 concat_alignments + the classify call of :494
 *   stag_db_create        <- the in-memory result of stag/databases.py:36-108 load_db, packed
 *
 * Conventions: plain C types only; every function returns STAG_OK (0) or a negative
 * STAG_E_* code and stag_last_error() then describes it (thread-local); the caller owns
 * all buffers; the library keeps no pointer after a call returns except inside the opaque
 * stag_db_t.  Per-sequence failures (the reference's Python exceptions) are reported in
 * status[] as STAG_SEQ_* and never abort the batch.  Functions ending in _dev take DEVICE
 * pointers and a hipStream_t (passed as void*; NULL = the null stream) and return without
 * synchronising; the others take HOST pointers and are synchronous.
 *
 * One stag_db_t lives on one GPU.  Calls on the same handle must not overlap (they share
 * the handle's workspace); different handles are independent.
 */
#ifndef STAG_HIP_H
#define STAG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STAG_OK 0
#define STAG_E_INVALID (-1)   /* bad argument / inconsistent tables */
#define STAG_E_HIP (-2)       /* a HIP runtime call failed */
#define STAG_E_NOMEM (-3)
#define STAG_E_NOGPU (-4)     /* no usable gfx950 device */

/* Per-sequence status.  1..5 are the exceptions the reference raises for that sequence. */
#define STAG_SEQ_OK 0
#define STAG_SEQ_MISSING_CLASSIFIER 1 /* KeyError:  classifiers[sibling]            classify.py:40 */
#define STAG_SEQ_STRING_CLASSIFIER 2  /* TypeError: "no_negative_examples" evaluated classify.py:24-25 */
#define STAG_SEQ_RANKCUT_SHAPE 3      /* ValueError: coeff * perc does not broadcast classify.py:25 */
#define STAG_SEQ_EMPTY_PATH 4         /* IndexError: perc[0] on an empty path        classify.py:106 */
#define STAG_SEQ_ZERO_COLUMNS 5       /* ZeroDivisionError: n_char == 0              align.py:58 */
#define STAG_SEQ_FILTERED 6           /* below min_perc_state: not yielded           align.py:160 */
#define STAG_SEQ_LENGTH_MISMATCH 7    /* ValueError: 5*n_char != L (coeff * seq)     classify.py:25 */
#define STAG_SEQ_BACKTRANSLATE 8      /* protein2gene_alignment returned None (align.py:66-74); the reference
                                         then fails with TypeError in convert_alignment(None) */

/*
 * Host-side description of one single-gene database (what load_db returns), already
 * numbered: tree positions in breadth-first order, root ("tree_root") = 0, so that the
 * children of node i are the consecutive ids child_first[i] .. child_first[i+1]-1 in the
 * order the database stores them (classify.py:39 iterates them in that order).
 */
typedef struct {
    int32_t n_nodes;
    int32_t n_features;        /* L = 5 * alignment columns */
    const int32_t *child_first; /* [n_nodes+1]; child_first[0] == 1, child_first[n_nodes] == n_nodes */
    const uint8_t *row_kind;   /* [n_nodes] 0 = no classifier, 1 = coefficients, 2 = "no_negative_examples" */
    const double *intercept;   /* [n_nodes] coeff_raw[0] (classify.py:24) */
    const int64_t *row_off;    /* [n_nodes+1] CSR over the non-zero coefficients of each row */
    const int32_t *row_feat;   /* feature index 0..L-1, ascending inside a row */
    const double *row_w;       /* coefficient of that feature */
    int32_t n_keys;            /* rank-cut keys, in sorted(str) order (classify.py:62) */
    int32_t n_tf;              /* coefficients per rank-cut row (excluding the intercept) */
    const double *tf;          /* [n_keys][1+n_tf] */
} stag_tables_t;

/* Per-sequence outputs.  Any pointer may be NULL (that output is skipped), except status. */
typedef struct {
    int32_t *path_node;     /* [n][max_depth] node ids along the greedy walk, -1 beyond path_len */
    double *path_prob;      /* [n][max_depth] after the placeholder fix-up of classify.py:106-110 */
    int32_t *path_len;      /* [n] */
    int32_t *n_lead_single; /* [n] leading levels decided by "exactly one child" (they print as int 1) */
    double *level_prob;     /* [n][n_keys] find_correct_level probabilities, sorted-key order */
    int32_t *selected;      /* [n] index into the sorted keys, -1 = nothing selected */
    int32_t *n_aligned;     /* [n] find_n_aligned_characters */
    int32_t *status;        /* [n] STAG_SEQ_* */
} stag_results_t;

typedef struct stag_db stag_db_t;

const char *stag_last_error(void);
int stag_device_count(int32_t *count);

/* Packs the tables into device memory on GPU `device`. */
int stag_db_create(const stag_tables_t *tables, int32_t device, stag_db_t **out);
void stag_db_destroy(stag_db_t *db);
/* max_depth: longest root-to-leaf path; words_per_row: uint32 words of one packed one-hot row. */
int stag_db_info(const stag_db_t *db, int32_t *max_depth, int32_t *words_per_row, int32_t *n_keys,
                 int64_t *table_bytes);
/* Grows the handle's device workspace for batches of up to n sequences / text_bytes bytes. */
int stag_db_reserve(stag_db_t *db, int64_t n, int64_t text_bytes);

/* uint32 words per packed row for L features (multiple of 4, so rows are 16-byte aligned). */
int32_t stag_words_per_row(int32_t n_features);

/*
 * align.py:37-58 + :160.  text holds n aligned (A2M) strings back to back; string i is
 * text[offs[i] .. offs[i+1]).  Bytes 0x0a..0x0d ('\n', '\v', '\f', '\r') inside a string are
 * skipped (A2M records may be line-wrapped; helpers.py:112 strips them).  One byte is one character:
 * 'a'..'z' are insert states and are dropped (align.py:47), every other byte is a column.  A caller whose
 * strings can hold letters outside ASCII (for which the reference asks str.islower()) passes a stand-in of
 * the same class instead, as stag_amd.tables.ascii_stand_in does.  Output row i (words_per_row words) has bit f set
 * iff one-hot feature f of string i is True: word f>>5, bit f&31.  For the _dev form, d_text is a
 * 16-byte aligned buffer of text_bytes bytes, text_bytes a multiple of 16 and >= offs[n] (pad it).
 * status: OK, ZERO_COLUMNS,
 * FILTERED (pct < min_perc_state) or LENGTH_MISMATCH (5*n_char != n_features).
 */
int stag_encode_a2m(int32_t device, int32_t n_features, const char *text, const int64_t *offs, int64_t n,
                    double min_perc_state, uint32_t *bits, int32_t *n_char, int32_t *n_aligned,
                    int32_t *status);
int stag_encode_a2m_dev(int32_t n_features, const char *d_text, int64_t text_bytes, const int64_t *d_offs,
                        int64_t n, double min_perc_state, uint32_t *d_bits, int32_t *d_n_char,
                        int32_t *d_n_aligned, int32_t *d_status, void *stream);

/* classify.py:94-124 for n packed rows.  status_in (may be NULL) carries encode's status:
 * sequences that are not STAG_SEQ_OK are skipped and keep their status. */
int stag_classify_bits(stag_db_t *db, const uint32_t *bits, int64_t n, const int32_t *status_in,
                       const stag_results_t *out);
int stag_classify_bits_dev(stag_db_t *db, const uint32_t *d_bits, int64_t n, const int32_t *d_status_in,
                           const stag_results_t *d_out, void *stream);

/*
 * classify.py:16-20 (alignment_reader) on the GPU, for the lines of a -s file in the form stag itself writes
 * (-S, classify.py:160-162): after the id, fields that are the single characters '0' / '1' separated by single
 * tabs.  The caller finds the lines (stag_onehot_lines in stag_ingest.h does, without looking at the fields) and
 * hands over, per line, its body text[body_begin[i] .. body_end[i]): what follows the id's tab, right-stripped.
 * Row i (words_per_row words, bit f -> word f>>5, bit f&31) is written when the line has exactly n_features such
 * fields and is zero otherwise; n_fields[i] = the fields of the line (0 when the body is not of that form);
 * flag[i] = 1 when the body is not "x<tab>x<tab>...x" with every x in {0, 1}: the reference's int() per field
 * accepts much more (" 1", "+1", "01", other digits), so such a line is not guessed at -- the caller parses it the
 * reference's way and supplies the row (patch_rows below).
 * _dev form: d_text is a 4-byte aligned buffer of text_bytes bytes, text_bytes a multiple of 4 (pad it).
 */
int stag_parse_onehot_dev(int32_t n_features, const char *d_text, int64_t text_bytes, const int64_t *d_body_begin,
                          const int64_t *d_body_end, int64_t n, uint32_t *d_bits, int32_t *d_n_fields,
                          int32_t *d_flag, void *stream);
/* Host form, in two steps so that the caller can parse the flagged lines in between: _parse copies the text up,
 * parses it and leaves the n rows in the handle; _classify overwrites the rows patch_index[j] (j < n_patch) with
 * patch_rows[j * words_per_row ..], then runs classify.py:94-124 on the n rows as stag_classify_bits does
 * (status_in as there, e.g. STAG_SEQ_LENGTH_MISMATCH for lines with n_fields != n_features). */
int stag_onehot_text_parse(stag_db_t *db, const char *text, int64_t text_bytes, const int64_t *body_begin,
                           const int64_t *body_end, int64_t n, int32_t *n_fields, int32_t *flag);
int stag_onehot_text_classify(stag_db_t *db, int64_t n, const int64_t *patch_index, const uint32_t *patch_rows,
                              int64_t n_patch, const int32_t *status_in, const stag_results_t *out);

/* Encode + classify without the rows leaving the GPU.  bits_out (may be NULL) receives the
 * packed rows (for -S, classify.py:160-162).
 * The host form is a pipeline: the batch is cut into sub-batches of about 64 MB of text (65,536
 * to 262,144 records, the last few smaller) whose text goes up on one stream while the previous sub-batch is
 * computed on a second and the results of the one before come down on a third, straight into the caller's arrays.
 * The copies overlap only if the caller's buffers are page-locked -- stag_host_alloc gives such memory (text, offs,
 * every output array); with pageable buffers the call is still correct, the HIP runtime then stages each copy itself.
 * offs (and lead_trail below) are validated sub-batch by sub-batch, each before its upload is queued: a malformed
 * record stops the call with STAG_E_INVALID, and the output arrays may then hold the results of the sub-batches
 * before it. */
int stag_classify_a2m(stag_db_t *db, const char *text, const int64_t *offs, int64_t n,
                      double min_perc_state, const stag_results_t *out, uint32_t *bits_out,
                      int32_t *n_char_out);
int stag_classify_a2m_dev(stag_db_t *db, const char *d_text, int64_t text_bytes, const int64_t *d_offs,
                          int64_t n, double min_perc_state, const stag_results_t *d_out, void *stream);

/*
 * The same for records whose flanking gap columns the host has dropped (align.py:37-58 again: a '-' column is
 * 10000, which is what a row says wherever nothing else was written, so a run of '-' at either end of an aligned
 * string only moves the column count).  Record i's aligned string = lead_trail[2 i] times '-', then
 * text[offs[i] .. offs[i+1]), then lead_trail[2 i + 1] times '-'; every output is what stag_classify_a2m gives
 * for that string (n_char counts the dropped columns too).  stag_fasta_scan_trim (stag_ingest.h) produces this
 * form while it copies the records out of the aligner's stream: an amplicon read of 253 columns in a
 * 1,500-column alignment travels to the GPU as 0.27 kB instead of 1.5 kB.  lead_trail may be NULL (nothing
 * dropped: stag_classify_a2m).
 */
int stag_classify_a2m_trimmed(stag_db_t *db, const char *text, const int64_t *offs, const int32_t *lead_trail,
                              int64_t n, double min_perc_state, const stag_results_t *out, uint32_t *bits_out,
                              int32_t *n_char_out);
int stag_classify_a2m_trimmed_dev(stag_db_t *db, const char *d_text, int64_t text_bytes, const int64_t *d_offs,
                                  const int32_t *d_lead_trail, int64_t n, double min_perc_state,
                                  const stag_results_t *d_out, void *stream);

/*
 * NO COUNTERPART IN THE REFERENCE.  stag_classify_a2m_dev for callers that cannot allocate device
 * memory: code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  Same
 * arguments, same kernels, same results, but the call allocates nothing (no hipMalloc, hipFree or
 * event creation): its workspace -- packed rows, grouping arrays, level records, and a stand-in for
 * every output pointer left NULL -- is carved out of one static device array of
 * STAG_NOALLOC_SCRATCH_BYTES per GPU (held from library load until the process ends; its address is
 * looked up by stag_db_create).  A batch whose workspace does not fit is refused with STAG_E_INVALID
 * and a message naming the bytes it would need: about n * (4 * words_per_row + 16 * max_depth + 40)
 * + 544 * (widest tree level + 2) + 8 * n_nodes bytes when every output pointer is given (~56,000
 * reads of a 1,500-column, 7-rank, 20,000-node database).  All calls on one GPU share that array,
 * whatever their handle: calls that overlap on one GPU (two streams, two threads) race; it keeps
 * the previous call's contents, and everything that must start at zero is cleared by the call.
 */
#define STAG_NOALLOC_SCRATCH_BYTES (64 * 1024 * 1024)
int stag_classify_a2m_noalloc_dev(stag_db_t *db, const char *d_text, int64_t text_bytes, const int64_t *d_offs,
                                  int64_t n, double min_perc_state, const stag_results_t *d_out, void *stream);

/*
 * align.py:63-94 on the GPU for n (protein alignment, gene) pairs laid out like the A2M text
 * (record i = text[offs[i] .. offs[i+1])).  Record i's codon alignment is written to
 * d_out[d_out_begin[i] .. d_out_end[i]) with d_out_begin[i] = 3 * prot_offs[i]; d_out must hold
 * 3 * prot_bytes bytes.  d_status[i]: STAG_SEQ_OK, or STAG_SEQ_BACKTRANSLATE when check_length is
 * set and the gene is neither 3x nor 3x+3 the number of non-'-' characters (then nothing is written).
 */
int stag_backtranslate_dev(const char *d_prot, int64_t prot_bytes, const int64_t *d_prot_offs, const char *d_gene,
                           const int64_t *d_gene_offs, int64_t n, int32_t check_length, char *d_out, int64_t out_bytes,
                           int64_t *d_out_begin, int64_t *d_out_end, int32_t *d_status, void *stream);

/* Protein databases (align.py:155-160 + :162 + classify.py:155-158): back-translate, encode and
 * classify without the codon alignment leaving the GPU.  Arguments as stag_classify_a2m. */
int stag_classify_protein_a2m(stag_db_t *db, const char *prot, const int64_t *prot_offs, const char *gene,
                              const int64_t *gene_offs, int64_t n, double min_perc_state, const stag_results_t *out,
                              uint32_t *bits_out, int32_t *n_char_out);
int stag_classify_protein_a2m_dev(stag_db_t *db, const char *d_prot, int64_t prot_bytes, const int64_t *d_prot_offs,
                                  const char *d_gene, const int64_t *d_gene_offs, int64_t n, double min_perc_state,
                                  const stag_results_t *d_out, void *stream);

/*
 * classify_genome.py:364-388 + :494 fused: the concatenated row of each genome is assembled on the
 * GPU from the marker genes' packed rows and classified against the concatenated database, without
 * the round trip through a text file of n_features 0/1 fields per genome.  Gene p (in the
 * database's gene order) has gene_len[p] features and n_rows[p] packed rows of words_per_row[p]
 * words at rows[p] (host); src_row[j * n_genes + p] is the row of gene p that belongs to genome j,
 * or -1 when the genome lacks the gene (zero-filled, :375).  The gene lengths must add up to the
 * database's n_features.  bits_out (may be NULL) receives the concatenated rows.
 */
int stag_classify_concat(stag_db_t *db, int32_t n_genes, const int32_t *gene_len, const uint32_t *const *rows,
                         const int32_t *words_per_row, const int64_t *n_rows, const int32_t *src_row, int64_t n_genomes,
                         const stag_results_t *out, uint32_t *bits_out);

/* align.py:63-94 on the host, one pair.
 * out must hold 3*prot_len bytes.  *out_len = -1 reproduces the reference's `return None`. */
int stag_backtranslate(const char *prot, int64_t prot_len, const char *gene, int64_t gene_len,
                       int32_t check_length, char *out, int64_t *out_len);

/* Page-locked host memory for the buffers of the host entry points (hipHostMalloc / hipHostFree): what the
 * ingestion side parses text into and results come back in, so that the pipeline of stag_classify_a2m runs its
 * copies as DMA beside the kernels.  No counterpart in the reference (it has no device).  NULL on failure. */
void *stag_host_alloc(int64_t bytes);
void stag_host_free(void *p);

/*
 * Per-kernel device timing for bench.py: when enabled every launch on this handle is
 * bracketed by HIP events on the launch stream.  stag_profile_read synchronises, sums the
 * elapsed times per kernel class and clears the record.  Classes: 0 encode, 1 walk_level,
 * 2 group (assign/scan/scatter), 3 rankcut, 4 other.
 */
#define STAG_PROF_CLASSES 5
int stag_profile_enable(stag_db_t *db, int32_t on);
int stag_profile_read(stag_db_t *db, double *ms_per_class, int64_t *launches_per_class);

#ifdef __cplusplus
}
#endif
#endif /* STAG_HIP_H */
