#!/bin/bash
for c in 128 256 512 1024; do
  PSL_DW_CHUNK=$c python bench.py --steps 5 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python tools/one_line.py chunk=$c
done
python -m pytest tests -q -m gpu -k "backward or map_native" 2>&1 | tail -1
