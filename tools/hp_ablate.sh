#!/bin/bash
# GPU box: bench per-class GEMM times for the product library and the -DHP_ABLATE=n diagnostic builds
# (build them first: see tools/README.md; mri-super-resolution_amd/libinrhip_abl<n>.so)
ROOT=$(pwd)
for a in "" 1 2 4 8 3; do
  if [ -z "$a" ]; then unset INR_LIB; tag=full; else export INR_LIB=$ROOT/mri-super-resolution_amd/libinrhip_abl$a.so; tag=abl$a; fi
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['roofline']['per_class']
        print('$tag', 'ms/step %.2f' % d['ms_per_step'], ' '.join('%s %.3f' % (k[5:], v['avg_ms']) for k,v in c.items()))
"
done
