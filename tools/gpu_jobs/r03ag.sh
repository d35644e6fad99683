#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "conv or golden or cnn" 2>&1 | tail -3
python tools/find_h2d.py 2>&1 | grep "host-to-device"
python bench.py --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); pc=d['roofline']['per_class_ms']
print('train %.1f utt/s %.2f ms' % (d['value'], d['ms_per_step']), pc, 'sum %.2f' % sum(pc.values()))"
