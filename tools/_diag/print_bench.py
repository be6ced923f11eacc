"""stdin: one bench.py JSON line -> the few numbers an A/B run compares.  usage: python bench.py ... | python tools/_diag/print_bench.py [label]"""
import json
import sys
r = json.loads(sys.stdin.read())
print(sys.argv[1] if len(sys.argv) > 1 else "", r["value"], r["ms_per_step"], "bf16", r.get("value_bf16_storage"), "raw", r.get("value_windows_from_raw_series"),
      "per-batch", r.get("value_per_batch_launches"), "train", (r.get("train_step") or {}).get("ms_per_step"))
