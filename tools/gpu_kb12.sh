cd tools/kbench
for cfg in "128 3 1 16 65536" "128 7 5 16 65536" "128 11 3 16 65536" "128 11 1 3 5000" "64 3 1 16 131072" "64 7 3 16 131072" "64 11 5 16 131072" "32 3 3 16 262144" "32 7 1 16 262144" "32 11 5 16 262144"; do
 for impl in 0 30; do
    echo "== cfg $cfg impl $impl: $(timeout 60 ./kbench $cfg 7 $impl | grep -E '^time|check' | tr '\n' ' ')"
 done
done
timeout 60 ./kbench_tl 128 11 3 16 65536 3 30 | grep -A7 "^timeline"
timeout 60 ./kbench_tl 128 3 1 16 65536 3 30 | grep -A7 "^timeline"
timeout 60 ./kbench_tl 32 11 5 16 262144 3 30 | grep -A7 "^timeline"
