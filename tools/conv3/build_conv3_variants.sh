#!/bin/bash
# Builds tools/conv3/libvx_conv3.so - round 5's one-pass GroupNorm + SiLU + 3x3 convolution, a tool since round 6 (not in
# the shipped ABI) - and, optionally, compile-time variants of it for tools/conv3/conv3_bench.py (VX_CONV3_LIBRARY=...):
#   build_conv3_variants.sh                                       the plain tool library
#   build_conv3_variants.sh "name:flag,flag name2:flag ..."       e.g. "abl2:-DVX_C3_ABLATE=2 abl4:-DVX_C3_ABLATE=4"
# (VX_C3_ABLATE: 1 no MFMA, 2 no plane normalisation, 4 no plane copies after the prologue, 8 no weight copies)
# Links against v-express_amd/libvexpress_hip.so (vx_set_error, vx_check_launch, the last-kernel slot): build that first.
cd "$(dirname "$0")"
PKG=../../v-express_amd
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm -Wno-unused-function"
LINK="-shared -fPIC -L$PKG -l:libvexpress_hip.so -Wl,-rpath,\$ORIGIN/$PKG"
/opt/rocm/bin/hipcc $F -c vx_conv3.hip -o /tmp/c3_plain.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/c3_plain.o $LINK -o libvx_conv3.so || exit 1
mkdir -p ../c3libs
for v in $1; do
  name=${v%%:*}; flags=${v#*:}
  ( /opt/rocm/bin/hipcc $F ${flags//,/ } -c vx_conv3.hip -o /tmp/c3_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/c3_$name.o $LINK -Wl,-rpath,\$ORIGIN/../conv3/$PKG -o ../c3libs/$name.so ) &
done
wait
ls -la libvx_conv3.so ../c3libs 2>/dev/null
