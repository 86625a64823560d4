#!/bin/bash
V=iclr2025_3d-mom_amd/lib/var
MOM4D_LIB=$PWD/$V/f1.so python -m pytest tests/test_deform_field_gpu.py tests/test_ops_gpu.py tests/test_fused_step_gpu.py tests/test_golden_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
