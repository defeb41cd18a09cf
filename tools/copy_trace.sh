#!/bin/bash
# which memory copies does a step issue?  tools/copy_trace.sh tag [batch]  -> gpurun_out/<tag>_copies.txt
TAG=${1:-ct}; B=${2:-64}; ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/pct
timeout 600 rocprofv3 --memory-copy-trace --hip-runtime-trace --output-format csv -d /tmp/pct -o r -- python $ROOT/bench.py --batch $B --steps 4 --warmup 2 --no-cpu-baseline --no-alt > $OUT/${TAG}_ct.json 2> $OUT/${TAG}_ct.err
ls -R /tmp/pct | head -20
python - <<PY > $OUT/${TAG}_copies.txt
import csv, glob, collections
for f in glob.glob("/tmp/pct/**/*memory_copy_trace.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    print(f, len(rows), rows[0].keys() if rows else None)
    c=collections.Counter((r.get("Direction"), r.get("Bytes") or r.get("Size")) for r in rows)
    for k,v in c.most_common(40): print(v, k)
for f in glob.glob("/tmp/pct/**/*hip_api_trace.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    c=collections.Counter(r.get("Function") for r in rows)
    for k,v in c.most_common(25): print(v, k)
PY
cat $OUT/${TAG}_copies.txt
