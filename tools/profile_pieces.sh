#!/bin/bash
# In-situ cost of the kernel's pieces: build a -DACME_PROFILE_PIECES library and time the
# bench with each piece repeated R extra times per Newton iteration.
ROOT=${GRAFT_REPO_ROOT:-$PWD}; export PYTHONPATH=$ROOT; cd $ROOT
cp acme_jl_amd/csrc/libacme_hip.so /tmp/libacme_hip.keep
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DACME_PROFILE_PIECES acme_jl_amd/csrc/acme_hip.hip -o acme_jl_amd/csrc/libacme_hip.so || exit 1
run() { timeout 120 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --samples 2205 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['config']['newton_iters_per_sample'])"; }
ACME_PROF0=0 ACME_PROF1=0 ACME_PROF2=0 run base
ACME_PROF0=2 run eval_x2
ACME_PROF1=2 run lu_x2
ACME_PROF2=2 run back_x2
cp /tmp/libacme_hip.keep acme_jl_amd/csrc/libacme_hip.so
