for v in ${VARIANTS:-0 621 651 611}; do for m in batched loop; do
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --variant $v --mode $m 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('variant $v $m', d['value'], d['ms_per_step'], 'single', d['single_computation']['mray_s'], d['dose_crc32'])"
done; done
