cd tools/kbench
for impl in 7 12 13 14 15 16; do
  echo "== impl $impl"; timeout 60 ./kbench_tl 128 11 3 16 65536 3 $impl | grep -E "^time|c1 main|c2 main|total"
done
