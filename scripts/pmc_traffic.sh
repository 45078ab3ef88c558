#!/bin/bash
# usage: pmc_traffic.sh <name> script.py args...   -> two separate --pmc passes (FETCH_SIZE, WRITE_SIZE), csv
set -e
N=$1; shift
bash /root/repo/scripts/pmc_one.sh pmc_${N}_fetch "FETCH_SIZE" "$@"
bash /root/repo/scripts/pmc_one.sh pmc_${N}_write "WRITE_SIZE" "$@"
