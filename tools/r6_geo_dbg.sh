for g in 8,28,8 4,28,8 4,28,4 8,28,4 6,28,8 5,28,8 7,28,8 2,28,4 8,14,8 4,56,8 8,56,8 3,28,4 6,28,7; do
  printf "%s: " $g; python tools/debug_split.py 2,56,56,24,144,24,1 $g 2>&1 | grep -A1 "^split 1" | tr '\n' ' ' | sed 's/.*band per block//' | cut -c1-160; echo
done
