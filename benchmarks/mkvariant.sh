#!/bin/bash
# Build a tuning variant of the HIP library that differs from the product in a few translation units only:
#   benchmarks/mkvariant.sh NAME "ffc_k_conv.hip ffc_k_bwdz.hip" -DFFC_KO=4 ...
# The product's objects are copied into lib/variants/NAME/obj, the listed units are recompiled with the extra flags.
# (knock-out builds: FFC_SKIP_AGPR_CHECK=1 in the environment)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; TUS=$2; shift 2
V=$R/flash-fft-conv_amd/lib/variants/$NAME
mkdir -p $V/obj
cp -p $R/flash-fft-conv_amd/lib/obj/*.o $R/flash-fft-conv_amd/lib/obj/*.agpr_ok $V/obj/ 2>/dev/null || true
for t in $TUS; do rm -f $V/obj/$t.o $V/obj/${t%.*}.agpr_ok; done
rm -f $V/libflashfftconv_hip.so
cd $R/flash-fft-conv_amd && python build.py --variant $NAME "$@"
