for t in 0 1 2 4 3; do echo "PN_SPARSE_TILE=$t"; PN_SPARSE_TILE=$t python tools/c4_leg.py 2 2>&1 | grep -E "sparse_encoder" | head -1; done
