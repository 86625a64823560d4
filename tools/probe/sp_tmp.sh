#!/bin/bash
V=iclr2025_3d-mom_amd/lib/var
for r in 1 2; do for v in h0 h1 h2; do KBENCH_FREEZE=1 MOM4D_LIB=$PWD/$V/$v.so python tools/kbench.py hexplane_bwd 2>/dev/null | tail -1; done; done
