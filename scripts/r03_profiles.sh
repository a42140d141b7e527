#!/bin/bash
# the round's profile set (run on the GPU box): bash scripts/r03_profiles.sh <tag> c2 [c3 c5]
TAG=$1; shift
for cfg in "$@"; do
  echo "== $cfg =="; date +%T
  bash scripts/profile_round.sh $TAG $cfg 5 || { echo "profile_round $cfg failed"; tail -5 gpurun_out/${TAG}_${cfg}_profile.log; exit 1; }
done
date +%T
