#!/bin/bash
# generic GPU call: tools/gpu_call.sh <tag> <command...>  -- runs from the repo root, logs under gpurun_out/<tag>/
cd $GRAFT_REPO_ROOT
tag=$1; shift
mkdir -p gpurun_out/$tag
eval "$@" > gpurun_out/$tag/out.log 2>&1
echo "rc=$?"
tail -c 3000 gpurun_out/$tag/out.log
