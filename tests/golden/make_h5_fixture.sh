#!/bin/sh
# Builds make_h5_fixture.c against the HDF5 1.10 C library that ships in this image under /opt/conda and writes the
# fixtures next to this script.  Needed only to REGENERATE tests/golden/keras_weights_libhdf5*.h5 (committed).
set -e
cd "$(dirname "$0")"
gcc -w -O1 -o /tmp/make_h5_fixture make_h5_fixture.c -I/opt/conda/include -L/opt/conda/lib -lhdf5 -Wl,-rpath,/opt/conda/lib
/tmp/make_h5_fixture keras_weights_libhdf5.h5
/tmp/make_h5_fixture keras_weights_libhdf5_weightless.h5 empty
