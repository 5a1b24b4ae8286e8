#!/bin/bash
# GPU box: cache-policy A/B of the persistent GEMMs (diagnostic builds -DHP_A_AUX=2 / -DHP_MUL_AUX=2: non-temporal LDS-DMA of the A
# operand / non-temporal loads of the deferred epilogue's element-wise factor), interleaved rounds of the short bench
ROOT=$(pwd)
for round in 1 2; do
for v in "" nth; do
  if [ -z "$v" ]; then unset INR_LIB; tag="product"; else export INR_LIB=$ROOT/mri-super-resolution_amd/libinrhip_$v.so; tag=$v; fi
  [ -n "$v" ] && [ ! -f "$INR_LIB" ] && continue
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['roofline']['all_gemm_launches']['per_class']
        print('round $round $tag:', 'ms/step %.3f' % d['ms_per_step'], 'other %.4f' % d['roofline']['all_gemm_launches']['other_kernels_ms_per_step'], ' '.join('%s %.3f' % (k[5:], v['avg_ms']) for k,v in c.items()))
"
done; done
