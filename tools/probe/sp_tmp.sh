#!/bin/bash
python -m pytest tests/test_tile_rows_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
for r in 1 2 3; do echo "fused    $(python tools/kbench.py preprocess_bwd 2>/dev/null | tail -1)"; echo "separate $(MOM_ACT_SEPARATE=1 python tools/kbench.py preprocess_bwd 2>/dev/null | tail -1)"; done
