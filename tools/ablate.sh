#!/bin/bash
for k in "$@"; do
  SLSLAM_EXTRA_FLAGS="-DSLS_ABL=$k" python -c "from slslam_amd import build; build.build_lib(force=True)" || exit 1
  echo "ABL=$k"; python tools/sweep.py 1024 2000 chunks=0 steps=2 | tail -1
done
