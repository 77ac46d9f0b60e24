for d in 0 4; do
  echo "== MPCQP_DBG=$d"
  MPCQP_DBG=$d timeout -k 10 120 python tools/exp_split.py 65536 1.0 0 2>&1 | grep -E "us per solve|diff|equal|status" | tail -7
done
