#!/bin/bash
# Poll GPU clock / power (rocm-smi) every 0.2 s while a command runs:  tools/clock_probe.sh out.log -- cmd args...
OUT=$1; shift; shift
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | tr '\n' ' '; echo; sleep 0.2; done ) > "$OUT" &
POLL=$!
"$@"
RC=$?
kill $POLL 2>/dev/null
exit $RC
