#!/bin/bash
# A/B timing helper: builds the library of a given commit (default HEAD) next to the working tree's as masic_amd/lib/libmasic_hip_base.so;
# select it on the GPU box with MASIC_HIP_LIB=$PWD/masic_amd/lib/libmasic_hip_base.so (masic_amd/_lib.py).
set -e
REV=${1:-HEAD}
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
git -C "$R" archive "$REV" masic_amd/csrc include | tar -x -C "$T"
make -s -C "$T/masic_amd/csrc" -j8 OUT=libbase.so >/dev/null
cp "$T/masic_amd/csrc/libbase.so" "$R/masic_amd/lib/libmasic_hip_base.so"
rm -rf "$T"
echo "built masic_amd/lib/libmasic_hip_base.so from $REV"
