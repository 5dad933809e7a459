#!/bin/bash
# Development aid: rebuild libspalign.so, and only if that succeeded send gpurun_job.sh to a GPU box.
#   tools/gpu.sh [timeout_seconds]
set -e
cd "$(dirname "$0")/.."
make -s -j8 -C superpixel-align_amd/csrc 2>&1 | grep -E "error|Error" && { echo "BUILD FAILED"; exit 1; }
make -s -C oracle liborc.so
exec /usr/local/graft/bin/gpurun --timeout "${1:-1500}" -- 'bash gpurun_job.sh'
