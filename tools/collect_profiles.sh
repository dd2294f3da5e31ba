#!/bin/bash
# Developer tool (GPU box): the rocprofv3 summaries committed under profiles/ for a round.
#   tools/collect_profiles.sh r03        -> gpurun_out/<tag>_*  (copy the summaries into profiles/ afterwards)
# kernel-trace/stats and every --pmc pass are SEPARATE runs (never combined with other trace domains).
set -u
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_f16 -- python3 bench.py --steps 1 --warmup 0 --no-graphs --no-cpu-baseline --no-roofline --dual-stream 0 > $out/${tag}_f16.log 2>&1
python tools/kstats.py $out/${tag}_f16 45 > $out/${tag}_kernel_stats_frames16.txt 2>&1
cp $(ls $out/${tag}_f16/*/*kernel_stats.csv | head -1) $out/${tag}_kernel_stats_frames16_eager.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_f1 -- python3 bench.py --frames 1 --steps 1 --warmup 0 --no-graphs --no-cpu-baseline --no-roofline --dual-stream 0 > $out/${tag}_f1.log 2>&1
python tools/kstats.py $out/${tag}_f1 30 > $out/${tag}_kernel_stats_frames1.txt 2>&1
cp $(ls $out/${tag}_f1/*/*kernel_stats.csv | head -1) $out/${tag}_kernel_stats_frames1_eager.csv
# HBM traffic of the dominant shapes (FETCH_SIZE and WRITE_SIZE cannot share a pass)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_pmc_fetch -- python3 tools/pmc_conv.py 32 > $out/${tag}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${tag}_pmc_write -- python3 tools/pmc_conv.py 32 > $out/${tag}_pmc_write.log 2>&1
# MFMA occupancy / waits / LDS conflicts
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/${tag}_pmc_sq1 -- python3 tools/pmc_conv.py 32 > $out/${tag}_pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $out/${tag}_pmc_sq2 -- python3 tools/pmc_conv.py 32 > $out/${tag}_pmc_sq2.log 2>&1
python tools/pmc_summary.py $out/${tag}_pmc_fetch $out/${tag}_pmc_write $out/${tag}_pmc_sq1 $out/${tag}_pmc_sq2 > $out/${tag}_pmc_summary.json
cat $out/${tag}_pmc_fetch.log | grep "^shape" > $out/${tag}_pmc_shapes.txt
head -c 3000 $out/${tag}_pmc_summary.json
