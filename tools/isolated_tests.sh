#!/bin/bash
# every GPU test FILE in its own process (order-dependent state -- lazily initialised statics, plans left by earlier tests -- shows up
# only when a file runs first in its process):  bash tools/isolated_tests.sh
for f in tests/test_hip_*.py tests/test_graft_entry.py; do
  [ -f "$f" ] || continue
  printf "%-40s " "$f"; python -m pytest "$f" -q -m gpu 2>&1 | tail -1
done
