/*
 * pf_h5.c -- minimal HDF5 dataset reader/writer used by the Python host (h5py is not available).
 *
 * Covers exactly what the file contract needs: rank-0/1/2 datasets of f64, i64, i8,
 * h5py-style booleans (HDF5 ENUM{FALSE=0,TRUE=1} over int8, read as int8 like the reference does at
 * c_cuda/fdtd_data.h:804-806), gzip-compressed or not; and writing plain datasets
 * (write_outputs, c_cuda/fdtd_data.h:928-980).  Links the system libhdf5 (HDF5 C API).
 */
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
#include "hdf5.h"

enum { PF_H5_F64 = 0, PF_H5_F32 = 1, PF_H5_I64 = 2, PF_H5_I8 = 3, PF_H5_BOOL = 4, PF_H5_U8 = 5, PF_H5_I32 = 6 };

static char g_err[512];
const char *pf_h5_last_error(void) { return g_err; }
static int fail(const char *what, const char *a, const char *b) {
   snprintf(g_err, sizeof g_err, "pf_h5: %s (%s%s%s)", what, a ? a : "", b ? " :: " : "", b ? b : "");
   return -1;
}
static void quiet(void) { H5Eset_auto2(H5E_DEFAULT, NULL, NULL); }

static hid_t memtype(int code) {
   switch (code) {
      case PF_H5_F64: return H5T_NATIVE_DOUBLE;
      case PF_H5_F32: return H5T_NATIVE_FLOAT;
      case PF_H5_I64: return H5T_NATIVE_INT64;
      case PF_H5_I8: return H5T_NATIVE_INT8;
      case PF_H5_BOOL: return H5T_NATIVE_INT8;
      case PF_H5_U8: return H5T_NATIVE_UINT8;
      case PF_H5_I32: return H5T_NATIVE_INT32;
   }
   return -1;
}

/* class: 0 integer, 1 float, 2 enum(bool), -1 other */
int pf_h5_info(const char *path, const char *name, int *ndims, int64_t *dims, int *cls, int *size) {
   quiet();
   hid_t f = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
   if (f < 0) return fail("cannot open file", path, NULL);
   hid_t d = H5Dopen2(f, name, H5P_DEFAULT);
   if (d < 0) { H5Fclose(f); return fail("no such dataset", path, name); }
   hid_t s = H5Dget_space(d);
   int nd = H5Sget_simple_extent_ndims(s);
   hsize_t hd[8] = {0};
   if (nd > 8) nd = 8;
   if (nd > 0) H5Sget_simple_extent_dims(s, hd, NULL);
   for (int i = 0; i < nd; i++) dims[i] = (int64_t)hd[i];
   *ndims = nd;
   hid_t t = H5Dget_type(d);
   H5T_class_t c = H5Tget_class(t);
   *cls = (c == H5T_INTEGER) ? 0 : (c == H5T_FLOAT) ? 1 : (c == H5T_ENUM) ? 2 : -1;
   *size = (int)H5Tget_size(t);
   H5Tclose(t); H5Sclose(s); H5Dclose(d); H5Fclose(f);
   return 0;
}

/* names of the root group's datasets, '\n'-separated, into buf (cap bytes); returns the count or -1 */
struct list_ctx { char *buf; size_t cap, len; int n, overflow; };
static herr_t list_cb(hid_t g, const char *name, const H5L_info_t *info, void *op) {
   (void)info;
   struct list_ctx *c = (struct list_ctx *)op;
   H5O_info_t oi;
   if (H5Oget_info_by_name(g, name, &oi, H5P_DEFAULT) < 0 || oi.type != H5O_TYPE_DATASET) return 0;
   size_t l = strlen(name);
   if (c->len + l + 2 > c->cap) { c->overflow = 1; return 0; }
   memcpy(c->buf + c->len, name, l);
   c->len += l;
   c->buf[c->len++] = '\n';
   c->buf[c->len] = 0;
   c->n++;
   return 0;
}
int pf_h5_list(const char *path, char *buf, int64_t cap) {
   quiet();
   hid_t f = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
   if (f < 0) return fail("cannot open file", path, NULL);
   struct list_ctx c = {buf, (size_t)cap, 0, 0, 0};
   if (cap > 0) buf[0] = 0;
   H5Literate(f, H5_INDEX_NAME, H5_ITER_INC, NULL, list_cb, &c);
   H5Fclose(f);
   if (c.overflow) return fail("dataset list does not fit the buffer", path, NULL);
   return c.n;
}

int pf_h5_exists(const char *path, const char *name) {
   quiet();
   hid_t f = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
   if (f < 0) return 0;
   htri_t r = H5Lexists(f, name, H5P_DEFAULT);
   H5Fclose(f);
   return r > 0;
}

int pf_h5_read(const char *path, const char *name, int code, void *out, int64_t nelem) {
   quiet();
   hid_t mt = memtype(code);
   if (mt < 0) return fail("bad type code", name, NULL);
   hid_t f = H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT);
   if (f < 0) return fail("cannot open file", path, NULL);
   hid_t d = H5Dopen2(f, name, H5P_DEFAULT);
   if (d < 0) { H5Fclose(f); return fail("no such dataset", path, name); }
   hid_t s = H5Dget_space(d);
   hssize_t n = H5Sget_simple_extent_npoints(s);
   H5Sclose(s);
   if ((int64_t)n != nelem) { H5Dclose(d); H5Fclose(f); return fail("element count mismatch", path, name); }
   herr_t st = H5Dread(d, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, out);
   if (st < 0) {
      /* enum datasets whose conversion path is refused: read through the enum's integer base type */
      hid_t t = H5Dget_type(d);
      if (H5Tget_class(t) == H5T_ENUM && H5Tget_size(t) == 1) {
         hid_t nt = H5Tget_native_type(t, H5T_DIR_ASCEND);
         st = H5Dread(d, nt, H5S_ALL, H5S_ALL, H5P_DEFAULT, out);
         H5Tclose(nt);
      }
      H5Tclose(t);
   }
   H5Dclose(d); H5Fclose(f);
   if (st < 0) return fail("H5Dread failed", path, name);
   return 0;
}

/* mode 0: create/truncate the file; 1: open existing (create if missing) and replace/add the dataset */
int pf_h5_write(const char *path, const char *name, int code, int ndims, const int64_t *dims,
                const void *data, int mode, int gzip) {
   quiet();
   hid_t mt = memtype(code);
   if (mt < 0) return fail("bad type code", name, NULL);
   hid_t f = -1;
   if (mode == 1 && access(path, F_OK) == 0) { /* append to an existing file: never fall back to truncating it */
      f = H5Fopen(path, H5F_ACC_RDWR, H5P_DEFAULT);
      if (f < 0) return fail("cannot open existing file for update (locked, unreadable or not HDF5)", path, NULL);
   } else {
      f = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
      if (f < 0) return fail("cannot create file", path, NULL);
   }
   if (H5Lexists(f, name, H5P_DEFAULT) > 0) H5Ldelete(f, name, H5P_DEFAULT);
   hsize_t hd[8];
   int64_t total = 1;
   for (int i = 0; i < ndims; i++) { hd[i] = (hsize_t)dims[i]; total *= dims[i]; }
   hid_t s = (ndims == 0) ? H5Screate(H5S_SCALAR) : H5Screate_simple(ndims, hd, NULL);
   hid_t ft = mt;
   int own_ft = 0;
   if (code == PF_H5_BOOL) { /* h5py's bool: ENUM{FALSE=0,TRUE=1} over int8 */
      ft = H5Tenum_create(H5T_NATIVE_INT8);
      int8_t v = 0; H5Tenum_insert(ft, "FALSE", &v);
      v = 1; H5Tenum_insert(ft, "TRUE", &v);
      own_ft = 1;
   }
   hid_t pl = H5Pcreate(H5P_DATASET_CREATE);
   if (gzip > 0 && ndims > 0 && total > 0) {
      hsize_t ch[8];
      int64_t per = 1;
      for (int i = ndims - 1; i >= 0; i--) { /* chunk: whole trailing dims, first dim cut to ~1M elements */
         ch[i] = hd[i];
         if (i == 0) { int64_t c = (1 << 20) / (per > 0 ? per : 1); if (c < 1) c = 1; if ((hsize_t)c < hd[0]) ch[0] = (hsize_t)c; }
         else per *= dims[i];
      }
      H5Pset_chunk(pl, ndims, ch);
      H5Pset_deflate(pl, (unsigned)gzip);
   }
   hid_t d = H5Dcreate2(f, name, ft, s, H5P_DEFAULT, pl, H5P_DEFAULT);
   herr_t st = -1;
   if (d >= 0) {
      st = (total > 0) ? H5Dwrite(d, own_ft ? ft : mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, data) : 0;
      H5Dclose(d);
   }
   if (own_ft) H5Tclose(ft);
   H5Pclose(pl); H5Sclose(s); H5Fclose(f);
   if (st < 0) return fail("H5Dcreate/H5Dwrite failed", path, name);
   return 0;
}
