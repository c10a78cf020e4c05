#!/bin/bash
# usage: [RS_TUNE_SHAPE=dups] tools/pmc_pass.sh <tag> "<counters>" <kb> [lib]   (GPU box) -- one rocprofv3 --pmc pass over tools/rs_tune, csv under gpurun_out/pmc_<tag>
tag="$1"; ctrs="$2"; kb="$3"; lib="${4:-radix-sort_amd/lib/librsort_hip.so}"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 90 rocprofv3 --pmc $ctrs --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -o p -- $GRAFT_REPO_ROOT/tools/rs_tune $GRAFT_REPO_ROOT/$lib $kb 30 ${RS_TUNE_SHAPE:-random} 1 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
