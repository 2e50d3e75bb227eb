for n in 2000 4000 6000 7000 8000 10000 12288; do
  for V in "QPDO_DENSE_LOOKAHEAD=1" "QPDO_DENSE_LOOKAHEAD=0"; do
    echo -n "n=$n [$V] "; env $V timeout -k 10 120 python tools/dense_lab.py $n 5 2>&1 | tail -1 | sed 's/.*its/its/'
  done
done
