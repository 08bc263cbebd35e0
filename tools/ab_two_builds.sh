#!/bin/bash
# Same-box A/B of two builds of libmqgan_hip.so (kernel rates differ by +-4 % from box to box: only alternation on ONE box decides
# anything smaller - DESIGN 3.1).  Builds the library with FILE as of git revision REV into ab_libs/lib_a.so and with the working
# tree's FILE into ab_libs/lib_b.so (both travel with the gpurun snapshot: *.so is git-ignored, not gpurun-ignored), then prints
# the command that alternates them through MQGAN_HIP_LIB.
# The working tree is never touched (the r5 version wrote REV:FILE over the tracked source and relied on an EXIT trap to
# put it back - a kill during the first hipcc run left the old revision in the tree): revision A is built from a scratch copy of
# mqgan_amd/csrc + include in a temporary directory.
#   usage (in the build container): tools/ab_two_builds.sh REV mqgan_amd/csrc/igemm_h2.hip [bench flags ...]
set -e
REV=$1; FILE=$2; shift 2
FLAGS=${*:---config hifimusic --precision bf16 --steps 4 --warmup 2 --no-cpu-baseline --no-vq --no-alt --no-configs}
cd "$(dirname "$0")/.."
ROOT=$PWD
mkdir -p ab_libs
TMP=$(mktemp -d "${TMPDIR:-/tmp}/ab_build.XXXXXX")
trap 'rm -rf "$TMP"' EXIT
mkdir -p "$TMP/mqgan_amd" "$TMP/tools"
cp -r mqgan_amd/csrc "$TMP/mqgan_amd/csrc"
cp -r include "$TMP/include"
cp -r tools/microbench "$TMP/tools/microbench"
rm -rf "$TMP/mqgan_amd/csrc/obj" "$TMP/mqgan_amd/csrc/obj_variants"
git show "$REV:$FILE" > "$TMP/$FILE"
make -s -j8 -C "$TMP/mqgan_amd/csrc" > /dev/null
cp "$TMP/mqgan_amd/libmqgan_hip.so" ab_libs/lib_a.so
python -c "import __graft_entry__ as g; g.build_native()" > /dev/null
cp mqgan_amd/libmqgan_hip.so ab_libs/lib_b.so
cat <<MSG
built ab_libs/lib_a.so ($FILE @ $REV, from a scratch copy) and ab_libs/lib_b.so (working tree).  On the GPU box, e.g.:
  gpurun --timeout 900 -- 'for i in 1 2 3; do for v in a b; do echo -n "\$v "; MQGAN_HIP_LIB=\$PWD/ab_libs/lib_\$v.so python bench.py $FLAGS 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.readlines()[-1])[\"ms_per_step\"],2))"; done; done'
MSG
