/* TEST INFRASTRUCTURE -- writes the small HDF5 / NetCDF-4-style files under tests/golden/hdf5/ that pin parcels_amd/hdf5.py.
 * Built and run in the build container only (tools/make_hdf5_fixtures.sh: gcc against the libhdf5 1.10 of /opt/conda); the files
 * are committed, the library is not needed to read them.  Every value is a formula of its indices, restated in
 * tests/test_hdf5_reader.py:   f32(t,z,y,x) = 1000 t + 100 z + 10 y + x + 0.25     packed i16 = (7 t + 5 z + 3 y + x) % 2000 - 1000
 */
#include <hdf5.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NT 5
#define NZ 3
#define NY 6
#define NX 8

static void attr_f64(hid_t obj, const char* name, double v) {
    hid_t sp = H5Screate(H5S_SCALAR);
    hid_t a = H5Acreate2(obj, name, H5T_IEEE_F64LE, sp, H5P_DEFAULT, H5P_DEFAULT);
    H5Awrite(a, H5T_NATIVE_DOUBLE, &v);
    H5Aclose(a);
    H5Sclose(sp);
}
static void attr_i16(hid_t obj, const char* name, int16_t v) {
    hid_t sp = H5Screate(H5S_SCALAR);
    hid_t a = H5Acreate2(obj, name, H5T_STD_I16LE, sp, H5P_DEFAULT, H5P_DEFAULT);
    H5Awrite(a, H5T_NATIVE_SHORT, &v);
    H5Aclose(a);
    H5Sclose(sp);
}

/* which: 0 = library defaults (superblock 0, v1 object headers, symbol-table groups); 1 = netCDF-4 style (creation order tracked and
 * indexed, many objects -> dense link storage); 2 = libver latest (v2 headers, layout v4 chunk indices) */
static void make(const char* path, int which) {
    hid_t fapl = H5Pcreate(H5P_FILE_ACCESS), fcpl = H5Pcreate(H5P_FILE_CREATE);
    if (which == 2) H5Pset_libver_bounds(fapl, H5F_LIBVER_LATEST, H5F_LIBVER_LATEST);
    if (which == 1) H5Pset_link_creation_order(fcpl, H5P_CRT_ORDER_TRACKED | H5P_CRT_ORDER_INDEXED);
    hid_t f = H5Fcreate(path, H5F_ACC_TRUNC, fcpl, fapl);
    hsize_t dims[4] = {NT, NZ, NY, NX};
    static float data[NT][NZ][NY][NX];
    static int16_t pk[NT][NZ][NY][NX];
    for (int t = 0; t < NT; t++)
        for (int z = 0; z < NZ; z++)
            for (int y = 0; y < NY; y++)
                for (int x = 0; x < NX; x++) {
                    data[t][z][y][x] = 1000.f * t + 100.f * z + 10.f * y + x + 0.25f;
                    pk[t][z][y][x] = (int16_t)((7 * t + 5 * z + 3 * y + x) % 2000 - 1000);
                }
    hid_t sp = H5Screate_simple(4, dims, NULL);
    /* U: float32, one time level per chunk split in y and x (edge chunks partial), shuffle + deflate */
    {
        hid_t dcpl = H5Pcreate(H5P_DATASET_CREATE);
        hsize_t ch[4] = {1, NZ, 4, 5};
        H5Pset_chunk(dcpl, 4, ch);
        H5Pset_shuffle(dcpl);
        H5Pset_deflate(dcpl, 4);
        if (which == 1) H5Pset_attr_creation_order(dcpl, H5P_CRT_ORDER_TRACKED);
        hid_t d = H5Dcreate2(f, "U", H5T_IEEE_F32LE, sp, H5P_DEFAULT, dcpl, H5P_DEFAULT);
        H5Dwrite(d, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, data);
        attr_f64(d, "some_number", 42.5);
        H5Dclose(d);
        H5Pclose(dcpl);
    }
    /* V: packed int16 with CF attributes, chunks of two time levels, deflate + fletcher32; time level 3 never written -> fill value */
    {
        hid_t dcpl = H5Pcreate(H5P_DATASET_CREATE);
        hsize_t ch[4] = {1, 2, NY, NX};
        H5Pset_chunk(dcpl, 4, ch);
        H5Pset_deflate(dcpl, 2);
        H5Pset_fletcher32(dcpl);
        int16_t fill = -32767;
        H5Pset_fill_value(dcpl, H5T_NATIVE_SHORT, &fill);
        hid_t d = H5Dcreate2(f, "V", H5T_STD_I16LE, sp, H5P_DEFAULT, dcpl, H5P_DEFAULT);
        for (int t = 0; t < NT; t++) {
            if (t == 3) continue;
            hsize_t st[4] = {t, 0, 0, 0}, cn[4] = {1, NZ, NY, NX};
            hid_t fs = H5Dget_space(d), ms = H5Screate_simple(4, cn, NULL);
            H5Sselect_hyperslab(fs, H5S_SELECT_SET, st, NULL, cn, NULL);
            H5Dwrite(d, H5T_NATIVE_SHORT, ms, fs, H5P_DEFAULT, pk[t]);
            H5Sclose(ms);
            H5Sclose(fs);
        }
        attr_f64(d, "scale_factor", 0.01);
        attr_f64(d, "add_offset", 1.5);
        attr_i16(d, "_FillValue", fill);
        H5Dclose(d);
        H5Pclose(dcpl);
    }
    /* W: big-endian float64, contiguous */
    {
        static double w[NT][NZ][NY][NX];
        for (int t = 0; t < NT; t++)
            for (int z = 0; z < NZ; z++)
                for (int y = 0; y < NY; y++)
                    for (int x = 0; x < NX; x++) w[t][z][y][x] = -(double)data[t][z][y][x];
        hid_t d = H5Dcreate2(f, "W", H5T_IEEE_F64BE, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        H5Dwrite(d, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, w);
        H5Dclose(d);
    }
    /* T2: a (time, y, x) variable, ONE chunk for everything (libver latest: single-chunk index), no filter */
    {
        hsize_t d3[3] = {NT, NY, NX};
        hid_t s3 = H5Screate_simple(3, d3, NULL);
        hid_t dcpl = H5Pcreate(H5P_DATASET_CREATE);
        H5Pset_chunk(dcpl, 3, d3);
        static float t2[NT][NY][NX];
        for (int t = 0; t < NT; t++)
            for (int y = 0; y < NY; y++)
                for (int x = 0; x < NX; x++) t2[t][y][x] = data[t][0][y][x];
        hid_t d = H5Dcreate2(f, "T2", H5T_IEEE_F32LE, s3, H5P_DEFAULT, dcpl, H5P_DEFAULT);
        H5Dwrite(d, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, t2);
        H5Dclose(d);
        H5Pclose(dcpl);
        H5Sclose(s3);
    }
    /* S: chunked WITHOUT a filter and with fixed dimensions (libver latest: implicit / fixed-array index) */
    {
        hid_t dcpl = H5Pcreate(H5P_DATASET_CREATE);
        hsize_t ch[4] = {1, 1, NY, NX};
        H5Pset_chunk(dcpl, 4, ch);
        if (which == 2) H5Pset_alloc_time(dcpl, H5D_ALLOC_TIME_EARLY); /* -> implicit index */
        hid_t d = H5Dcreate2(f, "S", H5T_IEEE_F32LE, sp, H5P_DEFAULT, dcpl, H5P_DEFAULT);
        H5Dwrite(d, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, data);
        H5Dclose(d);
        H5Pclose(dcpl);
    }
    /* F: chunked + deflate with fixed dimensions (libver latest: fixed-array index of filtered chunks) */
    {
        hid_t dcpl = H5Pcreate(H5P_DATASET_CREATE);
        hsize_t ch[4] = {1, NZ, NY, NX};
        H5Pset_chunk(dcpl, 4, ch);
        H5Pset_deflate(dcpl, 1);
        hid_t d = H5Dcreate2(f, "F", H5T_IEEE_F32LE, sp, H5P_DEFAULT, dcpl, H5P_DEFAULT);
        H5Dwrite(d, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, data);
        H5Dclose(d);
        H5Pclose(dcpl);
    }
    /* coordinate variables + enough extra objects for dense link storage, and a sub-group */
    {
        double tm[NT], lon[NX];
        for (int t = 0; t < NT; t++) tm[t] = 86400.0 * t;
        for (int x = 0; x < NX; x++) lon[x] = 0.5 * x;
        hsize_t n1 = NT;
        hid_t s1 = H5Screate_simple(1, &n1, NULL);
        hid_t d = H5Dcreate2(f, "time_counter", H5T_IEEE_F64LE, s1, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        H5Dwrite(d, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, tm);
        H5Dclose(d);
        H5Sclose(s1);
        n1 = NX;
        s1 = H5Screate_simple(1, &n1, NULL);
        d = H5Dcreate2(f, "nav_lon", H5T_IEEE_F64LE, s1, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        H5Dwrite(d, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, lon);
        H5Dclose(d);
        for (int k = 0; k < 14; k++) {
            char nm[32];
            snprintf(nm, sizeof nm, "extra_variable_number_%02d", k);
            d = H5Dcreate2(f, nm, H5T_IEEE_F64LE, s1, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
            for (int x = 0; x < NX; x++) lon[x] = k + 0.125 * x;
            H5Dwrite(d, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, lon);
            H5Dclose(d);
        }
        hid_t g = H5Gcreate2(f, "grp", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        d = H5Dcreate2(g, "inner", H5T_IEEE_F64LE, s1, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        H5Dwrite(d, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, lon);
        H5Dclose(d);
        H5Gclose(g);
        H5Sclose(s1);
    }
    H5Sclose(sp);
    H5Fclose(f);
    H5Pclose(fapl);
    H5Pclose(fcpl);
}

int main(int argc, char** argv) {
    const char* dir = argc > 1 ? argv[1] : ".";
    char p[1024];
    snprintf(p, sizeof p, "%s/default_v0.h5", dir);
    make(p, 0);
    snprintf(p, sizeof p, "%s/netcdf4_style_dense.nc", dir);
    make(p, 1);
    snprintf(p, sizeof p, "%s/libver_latest.h5", dir);
    make(p, 2);
    return 0;
}
