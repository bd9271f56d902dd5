/* Writes tests/golden/keras_weights_libhdf5.h5 with the REAL HDF5 library (1.10.x, default "earliest" file format --
 * what h5py writes by default), in the layout of a Keras 2.x `model.save_weights("x.h5")` file
 * (tf.keras.Model.save_weights -> hdf5_format.save_weights_to_hdf5_group; reference call sites AttemptFour/main.py:168-190
 * ModelCheckpoint(save_weights_only=True), eval.py:140 load_weights(by_name=True, skip_mismatch=True)):
 *   /            attrs: layer_names (fixed-length string array), backend, keras_version (fixed-length strings)
 *   /<layer>     attrs: weight_names (fixed-length string array of "<layer>/<weight>:0")
 *   /<layer>/<layer>/<weight>:0   float32 dataset, contiguous
 * Values are a closed form of the index so that the pure-Python reader (masters-thesis_amd/h5lite.py) can be checked
 * element by element: data[i] = 0.5 * i - 3 + layer_index.
 * Build + run: see make_h5_fixture.sh.  This file is test tooling, not product code. */
#include <hdf5.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void str_array_attr(hid_t obj, const char* name, const char** items, int n) {
  size_t w = 1;
  for (int i = 0; i < n; ++i) if (strlen(items[i]) > w) w = strlen(items[i]);
  char* buf = (char*)calloc((size_t)n, w);
  for (int i = 0; i < n; ++i) memcpy(buf + (size_t)i * w, items[i], strlen(items[i]));
  hid_t t = H5Tcopy(H5T_C_S1);
  H5Tset_size(t, w);
  H5Tset_strpad(t, H5T_STR_NULLPAD);          /* numpy 'S' arrays as h5py stores them */
  hsize_t dims[1] = {(hsize_t)n};
  hid_t s = H5Screate_simple(1, dims, NULL);
  hid_t a = H5Acreate2(obj, name, t, s, H5P_DEFAULT, H5P_DEFAULT);
  H5Awrite(a, t, buf);
  H5Aclose(a); H5Sclose(s); H5Tclose(t); free(buf);
}

static void str_scalar_attr(hid_t obj, const char* name, const char* value) {
  hid_t t = H5Tcopy(H5T_C_S1);
  H5Tset_size(t, strlen(value));
  H5Tset_strpad(t, H5T_STR_NULLPAD);
  hid_t s = H5Screate(H5S_SCALAR);
  hid_t a = H5Acreate2(obj, name, t, s, H5P_DEFAULT, H5P_DEFAULT);
  H5Awrite(a, t, value);
  H5Aclose(a); H5Sclose(s); H5Tclose(t);
}

static void dataset(hid_t grp, const char* name, int rank, const hsize_t* dims, int layer_index) {
  size_t n = 1;
  for (int i = 0; i < rank; ++i) n *= dims[i];
  float* d = (float*)malloc(n * sizeof(float));
  for (size_t i = 0; i < n; ++i) d[i] = 0.5f * (float)i - 3.0f + (float)layer_index;
  hid_t s = rank ? H5Screate_simple(rank, dims, NULL) : H5Screate(H5S_SCALAR);
  hid_t ds = H5Dcreate2(grp, name, H5T_IEEE_F32LE, s, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
  H5Dwrite(ds, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, d);
  H5Dclose(ds); H5Sclose(s); free(d);
}

/* h5py writes `weight_names = []` of a layer without weights (Dropout, InputLayer) as np.asarray([]): an EMPTY float64
 * attribute of shape (0,) */
static void empty_f64_attr(hid_t obj, const char* name) {
  hsize_t dims[1] = {0};
  hid_t s = H5Screate_simple(1, dims, NULL);
  hid_t a = H5Acreate2(obj, name, H5T_IEEE_F64LE, s, H5P_DEFAULT, H5P_DEFAULT);
  H5Aclose(a); H5Sclose(s);
}

/* second fixture (argv[2] == "empty"): dense_a, dropout (no weights), input_1 (no weights), dense_b */
static int write_with_weightless_layers(const char* path) {
  hid_t f = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
  const char* lnames[4] = {"dense_a", "dropout", "input_1", "dense_b"};
  str_array_attr(f, "layer_names", lnames, 4);
  str_scalar_attr(f, "backend", "tensorflow");
  str_scalar_attr(f, "keras_version", "2.4.0");
  for (int i = 0; i < 4; ++i) {
    hid_t g = H5Gcreate2(f, lnames[i], H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    if (i == 1 || i == 2) {
      empty_f64_attr(g, "weight_names");
    } else {
      hid_t g2 = H5Gcreate2(g, lnames[i], H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
      hsize_t d1[2] = {(hsize_t)(3 + i), 4}, d2[1] = {4};
      dataset(g2, "kernel:0", 2, d1, i); dataset(g2, "bias:0", 1, d2, i);
      char w[2][64];
      sprintf(w[0], "%s/kernel:0", lnames[i]); sprintf(w[1], "%s/bias:0", lnames[i]);
      const char* wn[2] = {w[0], w[1]};
      str_array_attr(g, "weight_names", wn, 2);
      H5Gclose(g2);
    }
    H5Gclose(g);
  }
  H5Fclose(f);
  printf("wrote %s\n", path);
  return 0;
}

int main(int argc, char** argv) {
  const char* path = argc > 1 ? argv[1] : "keras_weights_libhdf5.h5";
  if (argc > 2 && strcmp(argv[2], "empty") == 0) return write_with_weightless_layers(path);
  hid_t f = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
  /* 24 layers: more than one symbol-table node in the root group (leaf K = 4 -> 8 entries per node) */
  enum { NL = 24 };
  char names[NL][32];
  const char* lnames[NL];
  for (int i = 0; i < NL; ++i) {
    if (i == 0) strcpy(names[i], "emb_text");
    else if (i == 1) strcpy(names[i], "lstm");
    else if (i == 2) strcpy(names[i], "time_distributed_softmax");
    else if (i == 3) strcpy(names[i], "input_bn");
    else sprintf(names[i], "dense_in_%d", i - 4);
    lnames[i] = names[i];
  }
  str_array_attr(f, "layer_names", lnames, NL);
  str_scalar_attr(f, "backend", "tensorflow");
  str_scalar_attr(f, "keras_version", "2.4.0");
  for (int i = 0; i < NL; ++i) {
    hid_t g = H5Gcreate2(f, names[i], H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    hid_t g2 = H5Gcreate2(g, names[i], H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    char w[4][96];
    const char* wn[4];
    int nw = 0;
    if (i == 0) {
      hsize_t d[2] = {11, 6}; dataset(g2, "embeddings:0", 2, d, i);
      sprintf(w[nw], "%s/embeddings:0", names[i]); nw++;
    } else if (i == 1) {
      hsize_t d1[2] = {22, 32}, d2[2] = {8, 32}, d3[1] = {32};
      dataset(g2, "kernel:0", 2, d1, i); dataset(g2, "recurrent_kernel:0", 2, d2, i); dataset(g2, "bias:0", 1, d3, i);
      sprintf(w[0], "%s/kernel:0", names[i]); sprintf(w[1], "%s/recurrent_kernel:0", names[i]); sprintf(w[2], "%s/bias:0", names[i]); nw = 3;
    } else if (i == 3) {
      hsize_t d[1] = {16};
      dataset(g2, "gamma:0", 1, d, i); dataset(g2, "beta:0", 1, d, i); dataset(g2, "moving_mean:0", 1, d, i); dataset(g2, "moving_variance:0", 1, d, i);
      sprintf(w[0], "%s/gamma:0", names[i]); sprintf(w[1], "%s/beta:0", names[i]); sprintf(w[2], "%s/moving_mean:0", names[i]);
      sprintf(w[3], "%s/moving_variance:0", names[i]); nw = 4;
    } else {
      hsize_t d1[2] = {(hsize_t)(5 + i), 16}, d2[1] = {16};
      if (i == 2) { d1[0] = 64; d1[1] = 101; d2[0] = 101; }             /* a larger dataset (26 KB) */
      dataset(g2, "kernel:0", 2, d1, i); dataset(g2, "bias:0", 1, d2, i);
      sprintf(w[0], "%s/kernel:0", names[i]); sprintf(w[1], "%s/bias:0", names[i]); nw = 2;
    }
    for (int k = 0; k < nw; ++k) wn[k] = w[k];
    str_array_attr(g, "weight_names", wn, nw);
    H5Gclose(g2); H5Gclose(g);
  }
  H5Fclose(f);
  printf("wrote %s\n", path);
  return 0;
}
