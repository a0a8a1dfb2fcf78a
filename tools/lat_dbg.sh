for a in "fwd 8 64 64 512 512 1 bf16" "fwd 8 256 256 128 128 1 bf16" "fwd 8 512 512 64 64 1 bf16"; do
  for dbg in 0 1 2 3 8 11; do
    echo -n "dbg=$dbg  "; UNETDC_LAT_DBG=$dbg python3 tools/op_bench.py $a 30 2>/dev/null
  done
done
