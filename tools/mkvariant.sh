#!/bin/bash
# usage: tools/mkvariant.sh <name> "<extra hipcc flags>" [Q ... | all]   (CPU; default Q list: 5; all: every object, host side included)
# Builds a variant of the product library out of tree (tools/variants/<name>/*.so, git-ignored): the in-tree objects are
# reused, only the fused kernels of the listed quadrature sizes are recompiled with the extra flags (-D tuning hooks of
# kernel_fused_pencil.hpp, -mllvm options).  tools/r3_variants.sh runs variants against the default on one box.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); name=$1; flags=$2; shift 2; qs=${@:-5}
T=/tmp/variants/$name; W=$T/pkg/csrc; rm -rf $T; mkdir -p $W $T/include; cp -r $R/ceedpetscsolid_amd/csrc/. $W/; cp $R/include/*.h $T/include/
cd $W; touch build/*.o build/isa_summary.txt
if [ "$qs" = all ]; then rm -f build/*.o; qs="2 3 4 5 6 7 8"; fi
if [ "$qs" = misc ]; then rm -f build/misc.o; qs=""; fi
for q in $qs; do rm -f build/fused_q${q}p*.o; done
make -s -j8 EXTRA_HIPFLAGS="$flags" libceed_mi355x.so libsolid_harness_mi355x.so
mkdir -p $R/tools/variants/$name; cp $W/*.so $R/tools/variants/$name/
python3 $R/tools/isa_guard.py $(for q in $qs; do ls build/fused_q${q}p*.o; done) --summary /tmp/variants/$name.isa.txt > /dev/null 2>&1 || true
grep -E "P=5,Q=5,HyperFSdF,geo=1|P=5,Q=5,HyperSSdF,geo=1" /tmp/variants/$name.isa.txt || true
