for rep in 1 2; do
for st in "0,0" "1000,2000" "2000,4000" "1000,4000" "2000,3000" "3000,5000"; do
  echo "n128 stagger $st: $(SIPP_INSTANCE_STAGGER_US=$st python bench.py --steps 10 --warmup 2 --no-cpu-baseline --inflight 1 2>/dev/null | python -c 'import json,sys; print(json.loads(sys.stdin.read())["ms_per_step"])')"
done
done
for st in "0,0" "10000,20000" "20000,40000" "5000,10000"; do
  echo "n1024 stagger $st: $(SIPP_INSTANCE_STAGGER_US=$st python bench.py --n 1024 --steps 3 --warmup 1 --no-cpu-baseline --inflight 1 2>/dev/null | python -c 'import json,sys; print(json.loads(sys.stdin.read())["ms_per_step"])')"
done
