#!/bin/bash
# first-count cost and steady step time for several probe sizes: tools/place_ab2.sh
for gb in 1 2 4; do
  for i in 1 2 3; do
    DSKGPU_PLACE=8 DSKGPU_PLACE_GB=$gb python3 tools/first_count.py 2>&1 | grep count | awk -v g=$gb '{printf "probe %s GB: %s %s %s | ", g, $3, $4, $5} END {print ""}'
  done
done
