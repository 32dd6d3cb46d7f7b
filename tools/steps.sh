for st in 1 2 3 5 9 10 18 27 36; do echo -n "steps=$st  "; PN_CONV_ABL=$((st*4096 + ${EXTRA:-0})) PN_CONV_TILE=1 python tools/conv_kscale.py 2>/dev/null | head -1; done
