python -m pytest tests/test_gpu_attention.py -x -q -m gpu 2>&1 | tail -3
python tools/bench_kernels.py attn 2>&1 | grep -v amdgpu.ids
for i in 1 2; do python bench.py --steps 10 --warmup 3 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('step', d['value'], d['ms_per_step'], 'gemm', d['roofline']['frac'], 'vit', d['vit_frac_of_mfma_peak'], 'parity', d['parity']['rel_err'])"; done
