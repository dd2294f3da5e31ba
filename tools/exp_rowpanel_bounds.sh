#!/bin/bash
# Developer tool (GPU box): which resource bounds the row-panel GEMM launches?  Three scratch builds of tools/build_dev.sh on the same
# box — shipped, row stores compiled out (-DDC_EXP_RP_NOSTORE=1), 7 of 8 MFMAs compiled out (-DDC_EXP_RP_NOMFMA=1) — through tools/bench_gemm.py.
# The table of DESIGN.md §5 (round 3) is this script's output.
set -e
cd "$(dirname "$0")/.."
for v in base nostore nomfma; do
  rm -f tools/ab/obj/gemm_rowpanel.o
  case $v in base) F="";; nostore) F="-DDC_EXP_RP_NOSTORE=1";; nomfma) F="-DDC_EXP_RP_NOMFMA=1";; esac
  tools/build_dev.sh $F > /dev/null 2>&1
  echo "== $v"
  DC_LIB_PATH=tools/ab/libdc_dev.so DC_BENCH_K=320 python tools/bench_gemm.py 2>&1 | grep "M="
done
rm -f tools/ab/obj/gemm_rowpanel.o
