#!/bin/bash
# The whole GPU suite with the three-wave / the pair form forced for EVERY launch where it is built (any batch size, any number
# of sub-steps) - goldens, oracle parity, fuzz, resets, device surface all then run through it.  Usage (GPU box): tools/forced_forms.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/forced; mkdir -p $O; cd $R
BSKGPU_TRI=1 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_gpu_pair.py --deselect tests/test_gpu_tri.py > $O/tri.log 2>&1; tail -1 $O/tri.log
BSKGPU_TRI=0 BSKGPU_PAIR=1 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_gpu_pair.py --deselect tests/test_gpu_tri.py > $O/pair.log 2>&1; tail -1 $O/pair.log
