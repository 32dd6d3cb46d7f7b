for t in 0 16 32; do echo "PN_SPARSE_TILE=$t"; PN_SPARSE_TILE=$t python tools/sparse_conv_isolated.py 2 2>&1 | grep -E "\-> 32|sum"; done
