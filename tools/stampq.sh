cd tools/micro
for e in 0; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DPN_WINO4_EXP=$e -I../../include wino4_stamps.hip -o /tmp/wino4_stamps_$e 2>/dev/null && echo "EXP $e ring 0" && PN_WINO4_RING=0 /tmp/wino4_stamps_$e | head -4
done
