for o in 0 1 0 1; do
  echo -n "overlap=$o "
  KGW_OVERLAP_SAMPLING=$o python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done
