#!/bin/bash
# Which traffic does the bf16-storage conv loop's time follow?  Builds the library with parts of the LDS traffic of conv_h2_kernel's
# 16x16x32 loop removed (MQ_H2_ABLATE, WRONG results on purpose: fragments replicated instead of read, DMA pieces skipped) and times
# the same layers with each.  The counts the vmcnt waits assume change with the DMA ablations only in the direction "fewer in
# flight", so the loop still terminates.   usage: tools/h2_ablate.sh [layer substrings]
set -e
cd "$(dirname "$0")/.."
C=mqgan_amd/csrc
make -s -j8 -C $C
for a in 0 1 2 3 4 8 12 15 16 32 64 96; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -DMQ_H2_ABLATE=$a -Iinclude -I$C -x hip -c $C/igemm_h2.hip -o /tmp/igemm_h2_abl$a.o
    objs=$(ls $C/*.o | grep -v igemm_h2.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libmqgan_abl$a.so $objs /tmp/igemm_h2_abl$a.o
    echo "== MQ_H2_ABLATE=$a (fprop/dgrad - 1: 1/4 A frags, 2: 1/TN B frags, 4: one B DMA piece, 8: one A DMA piece; wgrad - 16: one fragment read per operand, 32: 1/4 dy DMA, 64: 1/4 x DMA)"
    MQGAN_HIP_LIB=/tmp/libmqgan_abl$a.so python3 tools/h2_micro.py "$@" 2>&1 | grep -v amdgpu.ids
done
