line='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],2), "us per call", round(d["ms_per_step"]*1e3/(66*32),2))'
for r in 1 2; do
echo -n "c3 steps 3 warmup 1: "; python bench.py --steps 3 --warmup 1 --model llama3.1-8b --no-cpu-baseline --decode-tokens 2 2>/dev/null | python -c "$line"
echo -n "c3 steps 6 warmup 3: "; python bench.py --steps 6 --warmup 3 --model llama3.1-8b --no-cpu-baseline --decode-tokens 2 2>/dev/null | python -c "$line"
done
