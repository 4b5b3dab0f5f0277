#!/bin/bash
# usage (on the GPU box, from the repo root): tools/experiments/run_v2_probe.sh [abl]
set -e
B=tools/experiments/bin/c8v2
if [ "$1" = "abl" ]; then
  for shape in "24 24 256" "144 24 256" "144 48 128"; do timeout -k 5 120 $B $shape 32 abl; done
else
  for shape in "24 24 256" "72 24 256" "144 24 256" "48 48 128" "144 48 128" "96 96 64" "288 96 64"; do timeout -k 5 120 $B $shape; done
fi
