#!/bin/bash
# A/B runs of bench.py on one GPU box: every line of the config file (or of stdin) is "label | ENV=VAL ... | bench.py arguments";
# one summary line per run goes to gpurun_out/ab_<tag>.txt, the full JSON lines to gpurun_out/ab_<tag>.jsonl.
#   gpurun --timeout 1500 -- 'bash scripts/ab.sh r04_scan < scripts/ab_scan.cfg'
tag=${1:-ab}
out=gpurun_out/ab_$tag
: > $out.txt; : > $out.jsonl
while IFS='|' read -r label envs args; do
  [ -z "$label" ] && continue
  case "$label" in \#*) continue;; esac
  line=$(env $envs python bench.py $args 2> gpurun_out/ab_last.err | tail -1)
  echo "$line" >> $out.jsonl
  python - "$label" "$line" >> $out.txt <<'PY'
import json, sys
label, line = sys.argv[1].strip(), sys.argv[2]
try:
    d = json.loads(line)
    r = d.get("roofline", {})
    o = d.get("one_batch_at_a_time", {})
    pl = r.get("per_launch") or []
    extra = " ".join(f"{p['kernel'].split('<')[0][-14:]}:{p['ms']:.3f}ms/{p['frac']:.2f}" for p in pl if p.get("ms"))
    print(f"{label:40s} value {d['value']/1e6:6.3f} M q/s  {d['ms_per_step']:.3f} ms/step  scan avg {r.get('avg_launch_ms', 0):.3f} ms  "
          f"other {r.get('other_kernels_ms_per_step')}  one-batch {o.get('ms_per_step')}  {extra}")
except Exception as e:  # noqa: BLE001
    print(f"{label:40s} FAILED: {e}: {line[:200]}")
    print(open("gpurun_out/ab_last.err").read()[-1500:])
PY
done
cat $out.txt
