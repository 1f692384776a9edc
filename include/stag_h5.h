/*
 * stag_h5.h -- C ABI of libstag_h5.so: a minimal reader for the .stagDB HDF5 layout.
 *
 * Replaces the h5py calls inside stag/databases.py:36-108 (load_db) of zellerlab/stag 0.8.2.
 * The reference's loader only works with h5py 2.10 (it compares vlen strings with str,
 * databases.py:51,101; h5py >= 3 returns bytes); this reader goes to libhdf5 directly and
 * returns bytes + lengths, so the Python host decides how to decode.  Layout being read is the
 * one databases.py:111-147 (save_to_file) writes:
 *   /tool_version, /db_type, /hmm_file      vlen str [1]
 *   /align_protein, /use_cmalign            bool [1] (HDF5 enum over int8)
 *   /taxonomy/<node>                        vlen str [n_children]   (leaves: empty)
 *   /tax_function/<level>                   float64 [1 + n_levels]
 *   /classifiers/<node>                     float64 [1 + L]  or  vlen str ["no_negative_examples"]
 *
 * All functions return 0 on success and a negative code on failure; stag_h5_last_error()
 * describes the failure (thread-local).  Host only; no GPU involved.
 */
#ifndef STAG_H5_H
#define STAG_H5_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct stag_h5 stag_h5_t;

#define STAG_H5_FLOAT 0
#define STAG_H5_STRING 1
#define STAG_H5_INTEGER 2 /* includes the bool enum */

const char *stag_h5_last_error(void);
int stag_h5_open(const char *path, stag_h5_t **out);
void stag_h5_close(stag_h5_t *h);
/* members of a group, in name order (the order h5py iterates them, databases.py:78,87,99) */
int stag_h5_group_size(stag_h5_t *h, const char *group, int64_t *n);
int stag_h5_member_name(stag_h5_t *h, const char *group, int64_t idx, char *buf, int64_t buflen);
/* 1 if the object exists */
int stag_h5_exists(stag_h5_t *h, const char *name);
int stag_h5_dataset_info(stag_h5_t *h, const char *name, int32_t *cls, int64_t *n);
int stag_h5_read_f64(stag_h5_t *h, const char *name, double *out, int64_t n);
int stag_h5_read_u8(stag_h5_t *h, const char *name, uint8_t *out, int64_t n);
/* n strings concatenated into buf (no terminators), lens[i] = bytes of string i.  *need gets
 * the total byte count; when buflen < *need nothing is copied and the call still succeeds. */
int stag_h5_read_strings(stag_h5_t *h, const char *name, char *buf, int64_t buflen, int64_t *lens, int64_t n,
                         int64_t *need);

#ifdef __cplusplus
}
#endif
#endif
