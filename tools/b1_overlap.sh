#!/bin/bash
# busy / gap / overlap digest of the single-utterance loop:  tools/b1_overlap.sh tag [batch]
TAG=${1:-ov}; B=${2:-1}; ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for w in full vocoder; do for ms in 1 3; do
  rm -rf /tmp/pov
  PARROT_MRF_STREAMS=$ms timeout 600 rocprofv3 --kernel-trace -d /tmp/pov -o r -- python $ROOT/bench.py --batch $B --steps 40 --warmup 5 --no-cpu-baseline --no-alt --workload $w > $OUT/${TAG}_ov.json 2> $OUT/${TAG}_ov.err
  echo "workload=$w mrf_streams=$ms ms_per_step=$(python -c "import json;print(json.load(open('$OUT/${TAG}_ov.json'))['ms_per_step'])")"
  python $ROOT/tools/overlap_report.py $(find /tmp/pov -name '*.db' | head -1)
done; done
