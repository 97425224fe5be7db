#!/bin/bash
# build the kbench variants of one experiment round in parallel: tools/kbench/build_variants.sh  tag1 "flags1"  tag2 "flags2" ...
cd "$(dirname "$0")/../.."
mkdir -p tools/kbench/bin
while [ $# -gt 1 ]; do
  tag=$1; flags=$2; shift 2
  ( tools/kbench/build.sh tools/kbench/bin/kbench_$tag $flags > tools/kbench/bin/build_$tag.log 2>&1 || echo "BUILD FAILED $tag" ) &
done
wait
ls -la tools/kbench/bin | grep kbench_
