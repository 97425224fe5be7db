cd tools/kbench
for cfg in "128 3 1 16 65536" "128 7 5 16 65536" "128 11 3 16 65536" "128 11 1 3 5000"; do
 for impl in 0 7 9 10; do
    echo "== cfg $cfg impl $impl: $(timeout 60 ./kbench $cfg 7 $impl | grep -E '^time|check' | tr '\n' ' ')"
 done
done
timeout 60 ./kbench_tl 128 11 3 16 65536 3 9 | grep -A12 "^timeline"
