#!/bin/bash
# usage: tools/hb.sh cmd...  -> runs cmd with a heartbeat line on stdout every 60 s (gpurun takes a silent run for hung)
( while true; do sleep 60; echo "[hb] $(date +%T)"; done ) & HB=$!
"$@"; rc=$?
kill $HB 2>/dev/null
exit $rc
