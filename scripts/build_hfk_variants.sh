#!/bin/bash
# r06: variant builds of csrc/geom.hip ALONE (the victim of scripts/coexec_variants.py), into mmego_amd/lib/variants/ (git-ignored *.so,
# they travel to the GPU box with the snapshot).  Every variant but "pad" is built WITHOUT head_fk_loss's 144-KB LDS request.
set -e
cd "$(dirname "$0")/.."
out=mmego_amd/lib/variants
mkdir -p $out
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=on -shared -Wno-unused-function"
b() { name=$1; shift; /opt/rocm/bin/hipcc $F "$@" mmego_amd/csrc/geom.hip -o $out/libgeom_$name.so & }
b base  
b sel    -DMMEGO_HFK_VARIANT=1
b rcp    -DMMEGO_HFK_VARIANT=2
b selrcp -DMMEGO_HFK_VARIANT=3
wait
b nopk   -fno-slp-vectorize
b selnopk -DMMEGO_HFK_VARIANT=1 -fno-slp-vectorize
# (r05 "pad" variant: the 144-KB LDS request, removed from the product in r06)
wait
ls -la $out
