#!/bin/bash
# round 4, call v: kernel stats + PMC passes of the group-by shapes, the sort and the sparse-key hash join
cd "$GRAFT_REPO_ROOT"
bash profiles/collect_r04.sh ${1:-v} gb sort hj
