for t in "--colmod 3000 --tune cs_pace=0" "--colmod 3000 --kernel rows" "--colmod 30000 --tune cs_pace=0"; do
  echo "== $t"
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-backward --no-epoch $t 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('fwd_ms=%.3f Gedges/s=%.2f nnz=%d'%(r['ms_per_launch'], r['edges_per_s_fwd']/1e9, j['config']['nnz']))"
done
