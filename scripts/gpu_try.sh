#!/bin/bash
set -u
bash scripts/collect_profiles.sh r04 topkall topkall10 topk100 topk18k pgrid ptk ingest recs
