for B in 32768 65536 131072; do
  echo "== plain B=$B"; python tools/sweep_bench.py --cells "20,8,10" --batch $B --reps 2 2>&1 | grep "^| 20"
  echo "== hetero B=$B"; python tools/sweep_bench.py --cells "20,8,10" --batch $B --hetero /tmp/h.md 2>&1 | grep "^| 20"
done
