#!/bin/bash
# Builds the diagnostic (-DSECEDO_STAMPS) library into ab_libs/lib_stamps.so, then restores the normal build.
set -e
cd "$(dirname "$0")/.."
mkdir -p ab_libs
make -C secedo_amd/csrc clean >/dev/null
make -C secedo_amd/csrc -j4 EXTRA=-DSECEDO_STAMPS 2>&1 | grep -E "error|rror:" || true
cp secedo_amd/libsecedo_simmat.so ab_libs/lib_stamps.so
make -C secedo_amd/csrc clean >/dev/null
make -C secedo_amd/csrc -j4 2>&1 | grep -E "error|warning" || true
ls -la ab_libs/lib_stamps.so secedo_amd/libsecedo_simmat.so
