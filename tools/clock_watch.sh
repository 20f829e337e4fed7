#!/bin/bash
# sample the shader clock and the package power while a command runs:  bash tools/clock_watch.sh <cmd...>
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.25; done ) > /tmp/clk.log &
W=$!
"$@"
kill $W 2>/dev/null
sort /tmp/clk.log | uniq -c | sort -rn | head -8
