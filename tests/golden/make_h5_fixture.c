/* Generator of tests/golden/keras_layout_libhdf5.h5: a small file in the layout of a Keras
 * weight file (layer groups, weight_names / layer_names string-array attributes, nested dataset
 * paths, an empty attribute), written by the REAL HDF5 library so that chessrl_amd/h5lite.py's
 * reader is pinned against bytes it did not produce.  Built and run in the development container
 * only (HDF5 1.10.6 headers/library under /opt/conda):
 *   gcc -O1 -I/opt/conda/include tests/golden/make_h5_fixture.c -L/opt/conda/lib -lhdf5 \
 *       -Wl,-rpath,/opt/conda/lib -o /tmp/make_h5_fixture && /tmp/make_h5_fixture tests/golden/keras_layout_libhdf5.h5
 * Dataset element i holds start + 0.25 * i (starts: see main). */
#include <hdf5.h>
#include <string.h>
#include <stdlib.h>
static void str_attr_array(hid_t loc, const char *name, const char **vals, int n) {
    size_t maxlen = 1; for (int i = 0; i < n; i++) if (strlen(vals[i]) > maxlen) maxlen = strlen(vals[i]);
    hid_t t = H5Tcopy(H5T_C_S1); H5Tset_size(t, maxlen); H5Tset_strpad(t, H5T_STR_NULLPAD);
    hsize_t dims[1] = {(hsize_t)n};
    hid_t s = H5Screate_simple(1, dims, NULL);
    char *buf = calloc(n ? n : 1, maxlen);
    for (int i = 0; i < n; i++) memcpy(buf + i * maxlen, vals[i], strlen(vals[i]));
    hid_t a = H5Acreate2(loc, name, t, s, H5P_DEFAULT, H5P_DEFAULT);
    H5Awrite(a, t, buf); H5Aclose(a); H5Sclose(s); H5Tclose(t); free(buf);
}
static void str_attr_scalar(hid_t loc, const char *name, const char *val) {
    hid_t t = H5Tcopy(H5T_C_S1); H5Tset_size(t, strlen(val)); H5Tset_strpad(t, H5T_STR_NULLPAD);
    hid_t s = H5Screate(H5S_SCALAR);
    hid_t a = H5Acreate2(loc, name, t, s, H5P_DEFAULT, H5P_DEFAULT);
    H5Awrite(a, t, val); H5Aclose(a); H5Sclose(s); H5Tclose(t);
}
static void dset(hid_t g, const char *name, int rank, hsize_t *dims, float start) {
    hid_t lcpl = H5Pcreate(H5P_LINK_CREATE); H5Pset_create_intermediate_group(lcpl, 1);
    hid_t s = rank ? H5Screate_simple(rank, dims, NULL) : H5Screate(H5S_SCALAR);
    hid_t d = H5Dcreate2(g, name, H5T_IEEE_F32LE, s, lcpl, H5P_DEFAULT, H5P_DEFAULT);
    size_t n = 1; for (int i = 0; i < rank; i++) n *= dims[i];
    float *buf = malloc(n * sizeof(float)); for (size_t i = 0; i < n; i++) buf[i] = start + 0.25f * (float)i;
    H5Dwrite(d, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf);
    free(buf); H5Dclose(d); H5Sclose(s); H5Pclose(lcpl);
}
int main(int argc, char **argv) {
    hid_t f = H5Fcreate(argv[1], H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    const char *layers[] = {"input_1", "conv2d", "batch_normalization", "activation", "policy_out"};
    str_attr_array(f, "layer_names", layers, 5);
    str_attr_scalar(f, "backend", "tensorflow");
    str_attr_scalar(f, "keras_version", "2.2.4-tf");
    hid_t g; hsize_t d4[4] = {3, 3, 2, 4}, d1[1] = {4}, d2[2] = {8, 5}, d5[1] = {5};
    g = H5Gcreate2(f, "input_1", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT); str_attr_array(g, "weight_names", NULL, 0); H5Gclose(g);
    g = H5Gcreate2(f, "conv2d", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    { const char *w[] = {"conv2d/kernel:0", "conv2d/bias:0"}; str_attr_array(g, "weight_names", w, 2); }
    dset(g, "conv2d/kernel:0", 4, d4, 1.0f); dset(g, "conv2d/bias:0", 1, d1, -2.0f); H5Gclose(g);
    g = H5Gcreate2(f, "batch_normalization", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    { const char *w[] = {"batch_normalization/gamma:0", "batch_normalization/beta:0", "batch_normalization/moving_mean:0", "batch_normalization/moving_variance:0"};
      str_attr_array(g, "weight_names", w, 4); }
    dset(g, "batch_normalization/gamma:0", 1, d1, 10.f); dset(g, "batch_normalization/beta:0", 1, d1, 20.f);
    dset(g, "batch_normalization/moving_mean:0", 1, d1, 30.f); dset(g, "batch_normalization/moving_variance:0", 1, d1, 40.f); H5Gclose(g);
    g = H5Gcreate2(f, "activation", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT); str_attr_array(g, "weight_names", NULL, 0); H5Gclose(g);
    g = H5Gcreate2(f, "policy_out", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    { const char *w[] = {"policy_out/kernel:0", "policy_out/bias:0"}; str_attr_array(g, "weight_names", w, 2); }
    dset(g, "policy_out/kernel:0", 2, d2, 100.f); dset(g, "policy_out/bias:0", 1, d5, 200.f); H5Gclose(g);
    H5Fclose(f); return 0;
}
