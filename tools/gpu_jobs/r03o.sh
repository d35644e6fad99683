#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03o; mkdir -p $O
python tools/gemm_shapes_in_step.py 2>&1 | grep -v amdgpu.ids > $O/gemm_shapes_in_step.txt
python bench.py --batch 2 --seconds 2 --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tiny batch (2 x 2 s): ms_per_step %.2f (host-bound floor), classes %s launches %s' % (d['ms_per_step'], d['roofline']['per_class_ms'], d['roofline']['launches']))" > $O/host_floor.txt 2>&1
cat $O/host_floor.txt; head -50 $O/gemm_shapes_in_step.txt
timeout 600 python -m pytest tests/test_model_gpu.py -q -x -k "s2t_w2v2_golden or padding_free" 2>&1 | tail -3
