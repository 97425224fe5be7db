cd tools/kbench
for cfg in "128 3 1 16 65536" "128 11 3 16 65536"; do
 for impl in 0 1 3 4 2 5 6; do
  for st in 0 8000 16000 30000; do
    if [ $st != 0 ] && { [ $impl = 0 ] || [ $impl = 2 ] || [ $impl = 5 ] || [ $impl = 6 ]; }; then continue; fi
    echo "== cfg $cfg impl $impl stagger $st: $(timeout 60 ./kbench $cfg 5 $impl 0 $st | grep -E '^time|check' | tr '\n' ' ')"
  done
 done
 for impl in 1 3; do for st in 0 16000; do timeout 60 ./kbench_tl $cfg 3 $impl 0 $st | grep -A14 "^timeline"; done; done
done
