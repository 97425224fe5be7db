cd tools/kbench
timeout 60 ./kbench_tl 128 11 3 16 65536 3 20 | grep -A8 "^timeline"
