#!/bin/bash
# For a machine that HAS what this build's image lacks: KMC 3 (kmc, kmc_tools, kmc_dump), CMash (MakeStreamingDNADatabase.py,
# MakeStreamingPrefilter.py, StreamingQueryDNADatabase.py on PATH or under $CMASH_SCRIPTS) and, for this package's side, an AMD GPU.
# It runs the reference's pre-filter — the same command lines scripts/select_db.py:44-76 and local_tests/retrain_and_test_metalign.sh:49-66
# issue — and this package's build_db + select_db on ONE small fixture, and compares the two containment tables column by column.
# Until somebody has run it, stage A/B of this package is pinned to its own oracle only ("parity unpinned": DESIGN.md §2).
#
#   bash tools/verify_against_cmash.sh [--dry_run] [WORK_DIR]
#
# --dry_run: no tool is called; the fixture is generated and checked and every command is printed (what tests/test_select_and_cli.py runs).
# Exit status: 0 identical (or dry run fine); 1 the tables differ (tools/verify_kit/compare_csv.py says where and what it would mean);
# 2 a tool is missing.
set -u
DRY=0
if [ "${1:-}" = "--dry_run" ]; then DRY=1; shift; fi
HERE="$(cd "$(dirname "$0")/.." && pwd)"
WORK="${1:-$(mktemp -d /tmp/mg_verify.XXXXXX)}"
PY="${PYTHON:-python3}"
THREADS="${THREADS:-8}"
CM="${CMASH_SCRIPTS:+$CMASH_SCRIPTS/}"
run() { echo "+ $*"; if [ "$DRY" = "0" ]; then "$@" || { echo "FAILED: $*" >&2; exit 2; }; fi; }

echo "== fixture -> $WORK"
$PY "$HERE/tools/verify_kit/make_fixture.py" "$WORK" || exit 2
FILES="$WORK/training_files.txt"
REF="$WORK/reference"; OURS="$WORK/ours"
mkdir -p "$REF/tmp" "$OURS"

if [ "$DRY" = "0" ]; then
  for t in kmc kmc_tools kmc_dump; do command -v $t > /dev/null || { echo "$t is not on PATH" >&2; exit 2; }; done
  for t in MakeStreamingDNADatabase.py MakeStreamingPrefilter.py StreamingQueryDNADatabase.py; do
    [ -e "${CM}$t" ] || command -v $t > /dev/null || { echo "$t not found (PATH or \$CMASH_SCRIPTS)" >&2; exit 2; }
  done
fi

echo "== the reference's side: CMash training (retrain_and_test_metalign.sh:49-66), KMC + streaming query (select_db.py:44-76)"
run $PY ${CM}MakeStreamingDNADatabase.py "$FILES" "$REF/cmash_db_n1000_k60.h5" -n 1000 -k 60
run $PY ${CM}MakeStreamingPrefilter.py "$REF/cmash_db_n1000_k60.h5" "$REF/cmash_db_n1000_k60_30-60-10.bf" 30-60-10
# (dump_kmers.py of the reference's local_tests: every sketched 60-mer as a FASTA record; the reference's own script is used as it is)
run $PY "${METALIGN_REFERENCE:-/path/to/Metalign}/local_tests/dump_kmers.py" "$REF/cmash_db_n1000_k60.h5" "$REF/cmash_db_n1000_k60_dump.fa"
run kmc -v -k60 -fa -ci0 -cs3 -t$THREADS -jlogsample "$REF/cmash_db_n1000_k60_dump.fa" "$REF/cmash_db_n1000_k60_dump" "$REF/tmp"
run kmc -v -k60 -fq -ci2 -cs3 -t$THREADS -jlog_sample "$WORK/reads.fq" "$REF/tmp/reads_60mers" "$REF/tmp"
run kmc_tools simple "$REF/cmash_db_n1000_k60_dump" "$REF/tmp/reads_60mers" intersect "$REF/tmp/60mers_intersection"
run kmc_dump "$REF/tmp/60mers_intersection" "$REF/tmp/60mers_intersection_dump"
if [ "$DRY" = "0" ]; then awk '{print ">seq"; print $1}' "$REF/tmp/60mers_intersection_dump" > "$REF/tmp/60mers_intersection_dump.fa"; else echo "+ awk '{print \">seq\"; print \$1}' …_dump > …_dump.fa"; fi
run $PY ${CM}StreamingQueryDNADatabase.py "$REF/tmp/60mers_intersection_dump.fa" "$REF/cmash_db_n1000_k60.h5" "$REF/cmash_query_results.csv" 30-60-10 -c 0 -r 1000000 -v -f "$REF/cmash_db_n1000_k60_30-60-10.bf" --sensitive

echo "== this package's side: the table of the reference pipeline (hash mode 1 = CMash's k-mer hash as recollected), then select_db's pre-filter"
run $PY -m metalign_amd.build_db "$FILES" "$OURS/sketch_table" -n 1000 -k 30,40,50,60 --reference_pipeline --hash_mode cmash
run $PY -m metalign_amd.select_db "$WORK/reads.fq" "$WORK/" --sketch_table "$OURS/sketch_table" --temp_dir "$OURS/tmp" --keep_temp_files --dbinfo_out "$OURS/subset_db_info.txt" --db "$OURS/subset.fna"
echo "   (three more tables worth a run if the first differs: --hash_mode canonical; either mode with --sketch_hash forward)"

echo "== compare"
if [ "$DRY" = "1" ]; then
  echo "+ $PY $HERE/tools/verify_kit/compare_csv.py $REF/cmash_query_results.csv $OURS/tmp/cmash_query_results.csv"
  $PY - "$WORK" <<'PY' || exit 2
import json, os, sys, gzip
w = sys.argv[1]
e = json.load(open(os.path.join(w, "expected.json")))
files = [l.strip() for l in open(os.path.join(w, "training_files.txt")) if l.strip()]
assert len(files) == e["genomes"] == 20 and all(os.path.exists(f) for f in files)
assert sum(1 for _ in open(os.path.join(w, "reads.fq"))) == 4 * e["reads"]
assert gzip.open(files[0]).readline().startswith(b">NZ_VERIFY0000.1")
assert sum(1 for _ in open(os.path.join(w, "db_info.txt"))) == 22
print("dry run fine: fixture of %d genomes / %d reads, %d present genomes" % (e["genomes"], e["reads"], len(e["present"])))
PY
  exit 0
fi
$PY "$HERE/tools/verify_kit/compare_csv.py" "$REF/cmash_query_results.csv" "$OURS/tmp/cmash_query_results.csv"
