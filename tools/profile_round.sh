#!/bin/bash
# usage: tools/profile_round.sh <tag>   (GPU box) -- the round's evidence in one call: bench.py, the same bench (randomized rows) under
# rocprofv3 --kernel-trace --stats, and one --pmc pass per counter group over tools/rs_tune; everything under gpurun_out/<tag>_*
tag="$1"; R=$GRAFT_REPO_ROOT
cd $R && timeout -k 10 600 python bench.py --steps 5 --warmup 2 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof -- python3 $R/bench.py --steps 5 --warmup 2 --shapes randomized --extra-types "" --pairs "" --strings "" --sizes "" --composite-n 0 --no-cpu > $R/gpurun_out/${tag}_bench_rocprof.json 2> $R/gpurun_out/${tag}_bench_rocprof.err || exit 1
cd $R && python tools/rocpd_kernel_stats.py gpurun_out/${tag}_prof "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --shapes randomized (u32, u64; N = 2^30)" > gpurun_out/${tag}_kernel_stats_randomized.csv || exit 1
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof_dup -- python3 $R/bench.py --steps 5 --warmup 2 --shapes duplicates --extra-types "" --pairs "" --strings "" --sizes "" --composite-n 0 --no-cpu > $R/gpurun_out/${tag}_bench_duplicates_rocprof.json 2> $R/gpurun_out/${tag}_bench_duplicates_rocprof.err || exit 1
cd $R && python tools/rocpd_kernel_stats.py gpurun_out/${tag}_prof_dup "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --shapes duplicates (u32, u64; N = 2^30)" > gpurun_out/${tag}_kernel_stats_duplicates.csv || exit 1
for kb in 4 8; do
  tools/pmc_pass.sh ${tag}_f$kb "FETCH_SIZE" $kb || exit 1
  tools/pmc_pass.sh ${tag}_w$kb "WRITE_SIZE" $kb || exit 1
done
tools/pmc_pass.sh ${tag}_a "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" 4 || exit 1
tools/pmc_pass.sh ${tag}_c "GRBM_GUI_ACTIVE TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TD_TD_BUSY_sum" 4 || exit 1
# the duplicates shape (256 values): SQ and TA/TD groups, 4-byte keys
RS_TUNE_SHAPE=dups tools/pmc_pass.sh ${tag}_da "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" 4 || exit 1
RS_TUNE_SHAPE=dups tools/pmc_pass.sh ${tag}_dc "GRBM_GUI_ACTIVE TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TD_TD_BUSY_sum" 4 || exit 1
python tools/pmc_summary.py gpurun_out/pmc_${tag}_f4 gpurun_out/pmc_${tag}_w4 gpurun_out/pmc_${tag}_f8 gpurun_out/pmc_${tag}_w8 gpurun_out/pmc_${tag}_a gpurun_out/pmc_${tag}_c gpurun_out/pmc_${tag}_da gpurun_out/pmc_${tag}_dc > gpurun_out/${tag}_pmc.txt
