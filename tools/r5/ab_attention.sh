for i in 1 2; do
  python bench.py 2>/dev/null | tail -1 > gpurun_out/ab_new_$i.json
  SE3_LIB=tools/r5/ab/lib_attn_unscaled.so python bench.py 2>/dev/null | tail -1 > gpurun_out/ab_old_$i.json
done
python - <<'P'
import json
for n in ('new_1','old_1','new_2','old_2'):
    d=json.loads(open('gpurun_out/ab_%s.json'%n).read())
    r=d['roofline']
    print(n, d['value'], 'attn frac', r.get('frac'), 'quiet', d.get('roofline_quiet',{}).get('frac') if isinstance(d.get('roofline_quiet'),dict) else None, 'us', r.get('avg_launch_us', r.get('launch_us')))
P
