/* Test tooling (not part of the product): writes a .stagDB file in the HDF5 layout that the reference's
 * stag/databases.py:111-147 (save_to_file) defines -- SURVEY.md section 8 row a13 -- so that tests can build
 * databases of any size on the machine they run on (the reference's writer needs h5py, which the test
 * interpreter does not have).  Built by oracle/Makefile into tests/csrc/libstagdb_writer.so over the same
 * libhdf5 that libstag_h5.so reads with; tests/stagdb_writer.py is its ctypes wrapper.
 *
 *   tool_version, db_type        variable-length UTF-8 string [1]
 *   align_protein, use_cmalign   bool [1] (h5py's enum FALSE=0 / TRUE=1 over int8)
 *   hmm_file                     variable-length string [1], gzip
 *   taxonomy/<node>              variable-length string [n_children], gzip
 *   tax_function/<level>         float64 [1 + D], gzip
 *   classifiers/<node>           float64 [1 + L], gzip 8   or   string [1] "no_negative_examples"
 */
#include <hdf5.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static hid_t str_type(void) {
    hid_t t = H5Tcopy(H5T_C_S1);
    H5Tset_size(t, H5T_VARIABLE);
    H5Tset_cset(t, H5T_CSET_UTF8);
    return t;
}

static hid_t plist(int64_t n, int gzip) {
    hid_t p = H5Pcreate(H5P_DATASET_CREATE);
    if (gzip > 0 && n > 0) {
        hsize_t chunk = (hsize_t)n;
        H5Pset_chunk(p, 1, &chunk);
        H5Pset_deflate(p, (unsigned)gzip);
    }
    return p;
}

int64_t stagdb_create(const char *path) {
    H5open();
    H5Eset_auto2(H5E_DEFAULT, NULL, NULL);
    return (int64_t)H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
}

int stagdb_close(int64_t f) { return H5Fclose((hid_t)f) < 0 ? -1 : 0; }

int stagdb_group(int64_t f, const char *name) {
    hid_t g = H5Gcreate2((hid_t)f, name, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    if (g < 0) return -1;
    H5Gclose(g);
    return 0;
}

int stagdb_strings(int64_t f, const char *name, const char *const *values, int64_t n, int gzip) {
    hsize_t dim = (hsize_t)n;
    hid_t t = str_type(), s = H5Screate_simple(1, &dim, NULL), p = plist(n, gzip);
    hid_t d = H5Dcreate2((hid_t)f, name, t, s, H5P_DEFAULT, p, H5P_DEFAULT);
    int rc = d < 0 ? -1 : 0;
    if (!rc && n > 0 && H5Dwrite(d, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, values) < 0) rc = -1;
    if (d >= 0) H5Dclose(d);
    H5Pclose(p); H5Sclose(s); H5Tclose(t);
    return rc;
}

int stagdb_f64(int64_t f, const char *name, const double *values, int64_t n, int gzip) {
    hsize_t dim = (hsize_t)n;
    hid_t s = H5Screate_simple(1, &dim, NULL), p = plist(n, gzip);
    hid_t d = H5Dcreate2((hid_t)f, name, H5T_IEEE_F64LE, s, H5P_DEFAULT, p, H5P_DEFAULT);
    int rc = d < 0 ? -1 : 0;
    if (!rc && n > 0 && H5Dwrite(d, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, values) < 0) rc = -1;
    if (d >= 0) H5Dclose(d);
    H5Pclose(p); H5Sclose(s);
    return rc;
}

int stagdb_bool(int64_t f, const char *name, int value) {
    hsize_t dim = 1;
    int8_t v0 = 0, v1 = 1, v = value ? 1 : 0;
    hid_t t = H5Tenum_create(H5T_STD_I8LE);
    H5Tenum_insert(t, "FALSE", &v0);
    H5Tenum_insert(t, "TRUE", &v1);
    hid_t s = H5Screate_simple(1, &dim, NULL);
    hid_t d = H5Dcreate2((hid_t)f, name, t, s, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    int rc = d < 0 ? -1 : 0;
    if (!rc && H5Dwrite(d, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, &v) < 0) rc = -1;
    if (d >= 0) H5Dclose(d);
    H5Sclose(s); H5Tclose(t);
    return rc;
}
