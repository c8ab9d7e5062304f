#!/usr/bin/env bash
# Runs a list of GPU steps on the gpurun box; each step under its own timeout, logs under gpurun_out/<tag>_*.log.
# A step that times out (or is killed) aborts the session: no further GPU work after a suspected hang.
# usage: tools/gpu_session.sh <tag> "<timeout> <name> <command...>" ...
set -u
tag="$1"; shift
mkdir -p gpurun_out
for spec in "$@"; do
    tmo=$(echo "$spec" | cut -d' ' -f1)
    name=$(echo "$spec" | cut -d' ' -f2)
    cmd=$(echo "$spec" | cut -d' ' -f3-)
    log="gpurun_out/${tag}_${name}.log"
    echo "=== [$name] $cmd" | tee "$log"
    start=$(date +%s)
    timeout -k 10 "$tmo" bash -c "$cmd" >> "$log" 2>&1
    rc=$?
    echo "=== [$name] exit $rc after $(( $(date +%s) - start )) s" | tee -a "$log"
    tail -n 6 "$log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
        echo "=== step $name timed out: aborting the session"
        exit 1
    fi
done
exit 0
