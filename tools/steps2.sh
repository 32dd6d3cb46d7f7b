for abl in 0 1 2 4 8 3 5 9 12 15; do
  a=$(PN_CONV_ABL=$((18*4096 + abl + 64)) PN_CONV_TILE=${TILE:-1} python tools/conv_kscale.py 2>/dev/null | head -1 | awk '{print $(NF-3)}')
  b=$(PN_CONV_ABL=$((36*4096 + abl + 64)) PN_CONV_TILE=${TILE:-1} python tools/conv_kscale.py 2>/dev/null | head -1 | awk '{print $(NF-3)}')
  echo "abl=$abl  T18=$a T36=$b  per-step=$(python -c "print(round(($b-$a)/18,3))") us"
done
