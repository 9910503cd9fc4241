#!/bin/bash
# SQ counters (three passes) of kernels whose name contains <filter>, for one bench.py side config:
#   bash tools/pmc_bench.sh <tag> <filter> <bench.py args...>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; filt=$2; shift; shift
exec bash tools/pmc_py.sh "$tag" "$filt" bench.py "$@"
