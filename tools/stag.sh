for sm in 0 1 2 3 4 5; do echo "== stagger mode $sm"; PN_CONV_ABL=$((sm*256)) PN_CONV_TILE=1 python tools/conv_kscale.py 2>/dev/null | head -1; done
