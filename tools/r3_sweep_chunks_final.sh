mkdir -p gpurun_out/r3y
for c in "--chunks 1" "--chunks 2" "--chunks 3" "--chunks 4" "--chunks 2 --taper 0.5" "--chunks 3 --taper 0.5" "--chunks 2 --taper 0.25" "--chunks 4 --taper 0.6"; do
  echo "== $c"
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-ingest --no-dropin $c 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['stage_ms'])"
done
