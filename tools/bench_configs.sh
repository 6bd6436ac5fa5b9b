# The other BASELINE.json configurations at bench.py's defaults (B = 256 utterances per GPU in 3 trimmed row ranges); GPU box, repo root.
for m in EfficientConformerCTCMedium EfficientConformerCTCLarge ConformerCTCLarge EfficientConformerTransducerMedium; do
  python bench.py --steps 5 --warmup 2 --model $m --no-cpu-baseline --no-roofline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', round(d['value']/1e6,3), 'M frames/s', round(d['ms_per_step'],2), 'ms')"
done
