#!/bin/bash
# Development aid: rebuild libspalign.so, and only if that succeeded send gpurun_job.sh to a GPU box (retrying while the pod's
# GPU slots are busy: exit code 3 = nothing charged).
#   tools/gpu.sh [timeout_seconds]
cd "$(dirname "$0")/.."
if make -s -j8 -C superpixel-align_amd/csrc 2>&1 | grep -E "error|Error"; then echo "BUILD FAILED"; exit 1; fi
make -s -C oracle liborc.so || exit 1
for attempt in 1 2 3 4 5 6 7 8 9 10; do
    /usr/local/graft/bin/gpurun --timeout "${1:-1500}" -- 'bash gpurun_job.sh'
    rc=$?
    [ $rc -ne 3 ] && exit $rc
    echo "[gpu.sh] no slot (attempt $attempt), retrying in 120 s"
    sleep 120
done
exit 3
