/* libstag_h5.so: reads the .stagDB HDF5 layout through libhdf5 (see include/stag_h5.h). */
#include "stag_h5.h"

#include <hdf5.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct stag_h5 {
    hid_t file;
};

static __thread char g_err[512];

static int fail(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return -1;
}

const char *stag_h5_last_error(void) { return g_err; }

int stag_h5_open(const char *path, stag_h5_t **out) {
    if (!path || !out) return fail("null argument");
    H5open();
    H5Eset_auto2(H5E_DEFAULT, NULL, NULL); /* errors are reported through return codes */
    hid_t f = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
    if (f < 0) return fail("cannot open %s as HDF5", path);
    stag_h5_t *h = (stag_h5_t *)calloc(1, sizeof *h);
    if (!h) { H5Fclose(f); return fail("out of memory"); }
    h->file = f;
    *out = h;
    return 0;
}

void stag_h5_close(stag_h5_t *h) {
    if (!h) return;
    H5Fclose(h->file);
    free(h);
}

int stag_h5_exists(stag_h5_t *h, const char *name) {
    if (!h || !name) return 0;
    return H5Lexists(h->file, name, H5P_DEFAULT) > 0;
}

int stag_h5_group_size(stag_h5_t *h, const char *group, int64_t *n) {
    if (!h || !group || !n) return fail("null argument");
    hid_t g = H5Gopen2(h->file, group, H5P_DEFAULT);
    if (g < 0) return fail("no group %s", group);
    H5G_info_t info;
    herr_t e = H5Gget_info(g, &info);
    H5Gclose(g);
    if (e < 0) return fail("H5Gget_info(%s) failed", group);
    *n = (int64_t)info.nlinks;
    return 0;
}

int stag_h5_member_name(stag_h5_t *h, const char *group, int64_t idx, char *buf, int64_t buflen) {
    if (!h || !group || !buf || buflen < 1) return fail("null argument");
    ssize_t len = H5Lget_name_by_idx(h->file, group, H5_INDEX_NAME, H5_ITER_INC, (hsize_t)idx, buf, (size_t)buflen,
                                     H5P_DEFAULT);
    if (len < 0) return fail("no member %lld in %s", (long long)idx, group);
    if (len >= buflen) return fail("member name of %lld bytes does not fit", (long long)len);
    return 0;
}

int stag_h5_dataset_info(stag_h5_t *h, const char *name, int32_t *cls, int64_t *n) {
    if (!h || !name || !cls || !n) return fail("null argument");
    hid_t d = H5Dopen2(h->file, name, H5P_DEFAULT);
    if (d < 0) return fail("no dataset %s", name);
    hid_t t = H5Dget_type(d), s = H5Dget_space(d);
    H5T_class_t c = H5Tget_class(t);
    *cls = c == H5T_FLOAT ? STAG_H5_FLOAT : c == H5T_STRING ? STAG_H5_STRING : STAG_H5_INTEGER;
    *n = (int64_t)H5Sget_simple_extent_npoints(s);
    H5Sclose(s); H5Tclose(t); H5Dclose(d);
    return 0;
}

int stag_h5_read_f64(stag_h5_t *h, const char *name, double *out, int64_t n) {
    if (!h || !name || (!out && n)) return fail("null argument");
    hid_t d = H5Dopen2(h->file, name, H5P_DEFAULT);
    if (d < 0) return fail("no dataset %s", name);
    hid_t s = H5Dget_space(d);
    int64_t have = (int64_t)H5Sget_simple_extent_npoints(s);
    H5Sclose(s);
    herr_t e = -1;
    if (have == n) e = n ? H5Dread(d, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, out) : 0;
    H5Dclose(d);
    if (have != n) return fail("%s holds %lld values, caller expects %lld", name, (long long)have, (long long)n);
    return e < 0 ? fail("reading %s as float64 failed", name) : 0;
}

int stag_h5_read_u8(stag_h5_t *h, const char *name, uint8_t *out, int64_t n) {
    if (!h || !name || (!out && n)) return fail("null argument");
    hid_t d = H5Dopen2(h->file, name, H5P_DEFAULT);
    if (d < 0) return fail("no dataset %s", name);
    hid_t t = H5Dget_type(d), s = H5Dget_space(d);
    int64_t have = (int64_t)H5Sget_simple_extent_npoints(s);
    herr_t e = -1;
    int bad_size = H5Tget_size(t) != 1;
    if (have == n && !bad_size) {
        hid_t nt = H5Tget_native_type(t, H5T_DIR_ASCEND); /* the bool enum is read as its 1-byte base */
        e = n ? H5Dread(d, nt, H5S_ALL, H5S_ALL, H5P_DEFAULT, out) : 0;
        H5Tclose(nt);
    }
    H5Sclose(s); H5Tclose(t); H5Dclose(d);
    if (bad_size) return fail("%s is not a 1-byte type", name);
    if (have != n) return fail("%s holds %lld values, caller expects %lld", name, (long long)have, (long long)n);
    return e < 0 ? fail("reading %s failed", name) : 0;
}

int stag_h5_read_strings(stag_h5_t *h, const char *name, char *buf, int64_t buflen, int64_t *lens, int64_t n,
                         int64_t *need) {
    if (!h || !name || !need || (n && !lens)) return fail("null argument");
    hid_t d = H5Dopen2(h->file, name, H5P_DEFAULT);
    if (d < 0) return fail("no dataset %s", name);
    hid_t t = H5Dget_type(d), s = H5Dget_space(d);
    int64_t have = (int64_t)H5Sget_simple_extent_npoints(s);
    int rc = 0;
    *need = 0;
    if (H5Tget_class(t) != H5T_STRING) rc = fail("%s is not a string dataset", name);
    else if (have != n) rc = fail("%s holds %lld strings, caller expects %lld", name, (long long)have, (long long)n);
    else if (n == 0) rc = 0;
    else if (H5Tis_variable_str(t) > 0) {
        char **ptrs = (char **)calloc((size_t)n, sizeof(char *));
        hid_t mt = H5Tcopy(H5T_C_S1);
        H5Tset_size(mt, H5T_VARIABLE);
        H5Tset_cset(mt, H5Tget_cset(t));
        if (!ptrs) rc = fail("out of memory");
        else if (H5Dread(d, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, ptrs) < 0) rc = fail("reading %s failed", name);
        else {
            int64_t total = 0;
            for (int64_t i = 0; i < n; i++) { lens[i] = ptrs[i] ? (int64_t)strlen(ptrs[i]) : 0; total += lens[i]; }
            *need = total;
            if (buf && buflen >= total) {
                int64_t o = 0;
                for (int64_t i = 0; i < n; i++) { if (lens[i]) memcpy(buf + o, ptrs[i], (size_t)lens[i]); o += lens[i]; }
            }
            H5Dvlen_reclaim(mt, s, H5P_DEFAULT, ptrs);
        }
        H5Tclose(mt);
        free(ptrs);
    } else {
        size_t sz = H5Tget_size(t);
        char *raw = (char *)malloc(sz * (size_t)n + 1);
        if (!raw) rc = fail("out of memory");
        else if (H5Dread(d, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, raw) < 0) rc = fail("reading %s failed", name);
        else {
            int64_t total = 0;
            for (int64_t i = 0; i < n; i++) { lens[i] = (int64_t)strnlen(raw + i * sz, sz); total += lens[i]; }
            *need = total;
            if (buf && buflen >= total) {
                int64_t o = 0;
                for (int64_t i = 0; i < n; i++) { memcpy(buf + o, raw + i * sz, (size_t)lens[i]); o += lens[i]; }
            }
        }
        free(raw);
    }
    H5Sclose(s); H5Tclose(t); H5Dclose(d);
    return rc;
}
