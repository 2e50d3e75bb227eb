for n in 200 1000 3001 5000; do
  for V in "" "QPDO_DENSE_OUTER=2" "QPDO_DENSE_OUTER=8" "QPDO_SYRK_KC=32" "QPDO_SYRK_SWZ=0" "QPDO_DENSE_FPANEL=1" "QPDO_DENSE_FPANEL=1 QPDO_DENSE_OUTER=2" "QPDO_DENSE_LOOKAHEAD=0"; do
    echo -n "n=$n [$V] "; env $V timeout -k 10 120 python tools/dense_lab.py $n 3 2>&1 | tail -1 | sed 's/.*its/its/'
  done
done
