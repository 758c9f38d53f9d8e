#!/bin/bash
# A/B builds of k_count_kmers (k = 51 only): metalign_amd/libmetalign_hip_<tag>.so for every "tag:flags" argument, loaded with
# MG_LIB_PATH (tools/kcount_probe.py on one box, one after the other: the pool's boxes differ by more than most changes are worth).
#   bash tools/kcount_variants.sh "w3:-DMG_KC_WAVES_PER_EU=3" "w4:-DMG_KC_WAVES_PER_EU=4"
set -eu
cd "$(dirname "$0")/../metalign_amd/csrc"
OBJS="mg_core.o mg_sort.o mg_sketch.o mg_sketch_cmash.o mg_sketch_multi.o mg_contain.o mg_refpipe.o mg_pgzip.o mg_inflate.o mg_profile.o mg_ingest.o mg_stream.o"
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc -O3 -std=c++20 -fPIC -Wall -Wno-unused-function --offload-arch=gfx950 -DMG_KC_ONLY_K=51 $flags -c mg_kcount.hip -o /tmp/mg_kcount_$tag.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmetalign_hip_$tag.so $OBJS /tmp/mg_kcount_$tag.o -lz
  echo "built libmetalign_hip_$tag.so ($flags)"
done
