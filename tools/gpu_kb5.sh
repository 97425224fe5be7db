cd tools/kbench
for cfg in "128 3 1 16 65536" "128 7 5 16 65536" "128 11 3 16 65536"; do
 for impl in 0 3 5 6 7 8; do
    echo "== cfg $cfg impl $impl: $(timeout 60 ./kbench $cfg 7 $impl | grep -E '^time|check' | tr '\n' ' ')"
 done
done
for impl in 6 8; do timeout 60 ./kbench_tl 128 11 3 16 65536 3 $impl | grep -A12 "^timeline"; done
timeout 60 ./kbench_tl 128 3 1 16 65536 3 8 | grep -A12 "^timeline"
