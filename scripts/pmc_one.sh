#!/bin/bash
# usage: pmc_one.sh <outname> "<counters>" script.py args...   (one rocprofv3 --pmc pass, csv)
set -e
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/$1; shift
C="$1"; shift
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT -o pmc -- python3 "$@" > $OUT.log 2>&1
echo done
