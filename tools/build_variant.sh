#!/bin/bash
# Experiment builds of the library: tools/build_variant.sh NAME "-DLFM_MEASURE -DLFM_EXP_..."  ->  tools/variants/NAME/liblfm_hip.so
# (only dit.hip is recompiled; ops.o / vae.o come from the shipped build in lfm_amd/_lib).  The built libraries are git-ignored and travel with gpurun.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
out=${VARDIR:-tools/variants}/$name
mkdir -p $out
python -c "from lfm_amd import _build; _build.build()" >/dev/null
SLPFLAG=-fno-slp-vectorize; [ "$NOSLP" = "0" ] && SLPFLAG=""   # NOSLP=0: with the SLP vectorizer (A/B of the build flag)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-macro-redefined $SLPFLAG $@ -c lfm_amd/csrc/dit.hip -o $out/dit.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/liblfm_hip.so $out/dit.o lfm_amd/_lib/ops.o lfm_amd/_lib/vae.o
rm -f $out/dit.o
echo built $out/liblfm_hip.so
