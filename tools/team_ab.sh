#!/bin/bash
# A/B of the TD3 wave-chain kernel's team size at BASELINE configs[4]'s shard (24 chains): bench.py --team-size 1, 2, 3, 6
for G in 1 2 3 6; do
  timeout 300 python bench.py --only-config 4 --team-size $G 2>/dev/null | tail -1 > /tmp/team_$G.json
  python - $G <<'PY'
import json, sys
G = sys.argv[1]
d = json.loads(open('/tmp/team_%s.json' % G).read())
c = d[0] if isinstance(d, list) else d.get("configs", [d])[0]
print("G", G, "ms_per_step", c.get("ms_per_step"), "kernel_ms", c.get("kernel_ms"), "us/learn", c.get("us_per_learn_step_per_chain"))
PY
done
