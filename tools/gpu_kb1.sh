cd tools/kbench
for cfg in "128 11 3 16 65536" "128 3 1 16 65536" "64 11 5 16 131072" "32 11 1 16 262144" "32 3 1 16 262144"; do
  timeout 120 ./kbench $cfg | grep -E "time|check"
  timeout 120 ./kbench_tl $cfg
done
