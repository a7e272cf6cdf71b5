#!/bin/bash
# developer: time the cooperative kernel with parts switched off (results are then wrong): where does a sample's time go
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$1
for m in 0 1 2 3 4 32 8 16 63; do echo "ACME_COOP_DBG=$m"; ACME_COOP_DBG=$m timeout 120 python tools/generic_shape_probe.py 8192 1102 "nn 20" 2>&1 | tail -1 | cut -c1-200; done | tee gpurun_out/$1/coop_dbg.txt
