for abl in ${ABLS:-15 0}; do for tile in ${TILES:-1}; do echo "== ABL=$abl TILE=$tile HW=${HW:-256}"; PN_CONV_ABL=$abl PN_CONV_TILE=$tile python tools/conv_kscale.py 2>/dev/null; done; done
