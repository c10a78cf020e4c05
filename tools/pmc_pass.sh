#!/bin/bash
# usage: tools/pmc_pass.sh <tag> "<counters>" <kb>    (GPU box) -- one rocprofv3 --pmc pass over tools/rs_tune, csv under gpurun_out/pmc_<tag>
tag="$1"; ctrs="$2"; kb="$3"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 90 rocprofv3 --pmc $ctrs --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -o p -- $GRAFT_REPO_ROOT/tools/rs_tune $GRAFT_REPO_ROOT/radix-sort_amd/lib/librsort_hip.so $kb 30 random 1 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
