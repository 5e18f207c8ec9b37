#!/bin/bash
# Dev tool (GPU box): one bench leg under several builds / environments, alternating processes.
#   [ROUNDS=2] tools/ab_leg.sh <leg> "<ENV=.. lib>" ...   e.g.  tools/ab_leg.sh cfg2 "KR_MSW_OVERLAP=0 knode-cosserat_amd/lib/libknode_rod.so" "KR_MSW_OVERLAP=1 ..."
leg=$1; shift
for r in $(seq 1 ${ROUNDS:-2}); do
  for spec in "$@"; do
    envs=${spec% *}; lib=${spec##* }
    [ "$envs" = "$lib" ] && envs=""
    line=$(env $envs KR_LIB_PATH=$PWD/$lib timeout -k 10 200 python3 tools/leg_only.py $leg 2>/dev/null | grep "^leg ")
    echo "round $r [$spec] $line"
  done
done
